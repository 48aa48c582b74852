"""Phase timeline of k_mlp_fused from the ANIHIP_FUSED_TRACE stamps (development aid).

    (the stamps are compiled out of the shipped library: build a development copy first)
    VARIANT_TU=all tools/build_variants.sh ftrace "-DANIHIP_DEV_TRACE"
    TORCHANI_AMD_LIB=$PWD/build_alt/libanihip_ftrace.so ANIHIP_FUSED_TRACE=/tmp/ft.bin \
          python tools/kbench.py --side 40 --stages mlp --mask on --reps 1
    python tools/fused_trace.py /tmp/ft.bin
"""
import sys

import numpy as np

tw = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8, 32).astype(np.int64)   # [item][wave][stamp]
END = 15
keep = (tw[:, 0, 0] > 0) & (tw[:, 0, END] > 0)
item_of = np.nonzero(keep)[0]
tw = tw[keep]
l0b = bool((tw[:, 0, 13] > 0).all())   # phase 5 (layer-0 backward inside the kernel) stamps 13 and 8
if l0b:
    STAMPS = [0, 1, 2, 3, 4, 5, 6, 7, 9, 10, 11, 12, 16, 17, 13, 19, 20, 21, 8, END]
    NAMES = ["start->mask", "mask->L0 first group staged", "L0 k-loop", "L0 epilogue", "P1 gemm", "P1 epilogue",
             "P2 gemm", "P2 epilogue+head+seed", "P3 gemm", "P3 epilogue", "P4 gemm", "P5 ring request", "d act0 -> LDS planes",
             "barrier", "P5 row pointer + prev loads", "P5 gemm, half K (last pass)", "next item's prefetch + hand-over + barrier",
             "P5 read-add-write", "end barrier"]
else:
    STAMPS = [0, 1, 2, 3, 4, 5, 6, 7, 9, 10, 11, 12, END]
    NAMES = ["start->mask", "mask->L0 first group staged", "L0 k-loop", "L0 epilogue", "P1 gemm", "P1 epilogue",
             "P2 gemm", "P2 epilogue+head+seed", "P3 gemm", "P3 epilogue", "P4 gemm", "P4 store"]
t = tw[:, 0, :]
d = np.diff(t[:, STAMPS], axis=1)
tot = t[:, END] - t[:, 0]
print(f"{len(t)} items, total {tot.mean():.0f} ticks (median {np.median(tot):.0f})  [shader clock ticks, wave 0]"
      f"{'  (layer-0 backward inside)' if l0b else ''}")
for i, n in enumerate(NAMES):
    print(f"  {n:32s} mean {d[:, i].mean():9.1f}  median {np.median(d[:, i]):9.1f}  ({d[:, i].mean() / tot.mean():6.1%})")

# per wave: when does each wave pass each stamp, relative to wave 0's start of the item (mean over items)
print("per-wave arrival (mean ticks after wave 0 started the item); stamps:", STAMPS)
for w in range(8):
    ok = tw[:, w, END] > 0
    rel = tw[ok][:, w, :][:, STAMPS] - tw[ok][:, 0, 0][:, None]
    print(f"  wave {w}: " + " ".join(f"{v:7.0f}" for v in rel.mean(axis=0)))

# owner order (layer-0 backward inside): workgroup b owns tiles b, b + grid, ...; how far apart do the workgroups finish?
if l0b:
    grid = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    wg = (item_of // 8) % grid
    first = np.full(grid, np.iinfo(np.int64).max)
    last = np.zeros(grid, dtype=np.int64)
    np.minimum.at(first, wg, tw[:, 0, 0])
    np.maximum.at(last, wg, tw[:, 0, END])
    n_items = np.bincount(wg, minlength=grid)
    dur = (last - first).astype(np.float64)
    ok = n_items > 0
    print(f"workgroups {ok.sum()}: items per workgroup {n_items[ok].min()}..{n_items[ok].max()}; busy span (first start -> last end, own clock) "
          f"min {dur[ok].min() / dur[ok].mean():.4f}  max {dur[ok].max() / dur[ok].mean():.4f} of the mean {dur[ok].mean():.0f} ticks, std {dur[ok].std() / dur[ok].mean():.4f}")
    # per XCD (slot 14 low word: HW_ID; high word: XCC_ID)
    xcc = (tw[:, 0, 14] >> 32) & 0xF
    for x in np.unique(xcc):
        m = xcc == x
        tt = (tw[m][:, 0, END] - tw[m][:, 0, 0])
        print(f"  XCC {int(x)}: {m.sum()} items, mean item {tt.mean():.0f} ticks")

# sub-stamps inside the epilogues (round 5): for every wave, the time between consecutive stamps; mean over waves and items,
# and the spread of the waves' arrival at the stamp (max - min over the 8 waves of an item)
SUB = [("L0 epilogue", [3, 22, 23, 24, 25, 4], ["ring request + biases", "celu pairs", "tile max (atomic + barrier)", "split + LDS planes", "barrier"]),
       ("P1 epilogue", [5, 26, 27, 6], ["ring request + celu pairs", "split + LDS planes", "barrier"]),
       ("P2 epilogue", [7, 28, 29, 9], ["ring request + celu pairs + head", "split + LDS planes", "barrier + energies"]),
       ("P3 epilogue", [10, 30, 31, 11], ["ring request + celu' scale", "split + LDS planes", "barrier"])]
if (tw[:, :, 22] > 0).all():
    for title, st, names in SUB:
        ok = (tw[:, :, st] > 0).all(axis=(1, 2))
        x = tw[ok][:, :, st]                      # [item][wave][stamp]
        dd = np.diff(x, axis=2)
        print(f"{title}: per-wave segment times (mean over the 8 waves; [min .. max] of the per-wave means), arrival spread at the segment's end")
        for i, n in enumerate(names):
            per_wave = dd[:, :, i].mean(axis=0)
            spread = (x[:, :, i + 1].max(axis=1) - x[:, :, i + 1].min(axis=1)).mean()
            print(f"    {n:34s} {per_wave.mean():8.1f}  [{per_wave.min():8.1f} .. {per_wave.max():8.1f}]   spread {spread:7.1f}")
