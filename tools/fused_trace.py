"""Phase timeline of k_mlp_fused from the ANIHIP_FUSED_TRACE stamps (development aid).

    (the stamps are compiled out of the shipped library: build a development copy first)
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC -shared -DANIHIP_DEV_TRACE \
          -o tools/_ab/libanihip_trace.so torchani_amd/csrc/*.hip
    TORCHANI_AMD_LIB=$PWD/tools/_ab/libanihip_trace.so ANIHIP_FUSED_TRACE=/tmp/ft.bin \
          python tools/kbench.py --side 40 --stages mlp --mask on --reps 1
    python tools/fused_trace.py /tmp/ft.bin
"""
import sys

import numpy as np

STAMPS = [0, 1, 2, 3, 4, 5, 6, 7, 9, 10, 11, 12, 13]
NAMES = ["start->mask", "mask->L0 first group staged", "L0 k-loop", "L0 epilogue", "P1 gemm", "P1 epilogue",
         "P2 gemm", "P2 epilogue+head+seed", "P3 gemm", "P3 epilogue", "P4 gemm", "P4 store"]

tw = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8, 16).astype(np.int64)   # [item][wave][stamp]
tw = tw[(tw[:, 0, 0] > 0) & (tw[:, 0, 13] > 0)]
t = tw[:, 0, :]
d = np.diff(t[:, STAMPS], axis=1)
tot = t[:, 13] - t[:, 0]
print(f"{len(t)} workgroups, total {tot.mean():.0f} ticks (median {np.median(tot):.0f})  [shader clock ticks, wave 0]")
if t[:, 8].min() > 0 and t[:, 14].min() > 0:
    for a, b, n in ((1, 8, "  mask -> group 0 converted"), (8, 2, "  first barrier"), (2, 14, "  group-0 steps"),
                    (14, 15, "  stage group 1 + barrier"), (15, 3, "  rest of the k-loop")):
        x = t[:, b] - t[:, a]
        print(f"{n:32s} mean {x.mean():9.1f}  median {np.median(x):9.1f}")
for i, n in enumerate(NAMES):
    print(f"  {n:30s} mean {d[:, i].mean():9.1f}  median {np.median(d[:, i]):9.1f}  ({d[:, i].mean() / tot.mean():6.1%})")

# per wave: when does each wave pass each stamp, relative to wave 0's start of the item (mean over items)
print("per-wave arrival (mean ticks after wave 0 started the item); stamps:", STAMPS)
for w in range(8):
    ok = tw[:, w, 13] > 0
    rel = tw[ok][:, w, :][:, STAMPS] - tw[ok][:, 0, 0][:, None]
    print(f"  wave {w}: " + " ".join(f"{v:7.0f}" for v in rel.mean(axis=0)))
