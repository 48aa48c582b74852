"""Co-residency analysis of k_mlp_fused from the ANIHIP_FUSED_TRACE stamps (development aid): groups the work
items by CU and reports how much of a workgroup's lifetime and of its GEMM phases overlaps other workgroups on the
same CU.     python tools/fused_overlap.py /tmp/ft.bin"""
import sys
import collections
import numpy as np

t = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 16)
t = t[(t[:, 0] > 0) & (t[:, 13] > 0)]
hw = (t[:, 14] - 1).astype(np.uint64)
hwid, xcc = hw & np.uint64(0xFFFFFFFF), hw >> np.uint64(32)
cu = (hwid >> np.uint64(8)) & np.uint64(0xF)
sh = (hwid >> np.uint64(12)) & np.uint64(1)
se = (hwid >> np.uint64(13)) & np.uint64(7)
key = (xcc.astype(np.int64) * 64 + se.astype(np.int64) * 32 + sh.astype(np.int64) * 16 + cu.astype(np.int64))
print("distinct CU keys:", len(np.unique(key)), "items:", len(t))
tt = t.astype(np.int64)
GEMM = [(2, 3), (4, 5), (6, 7), (9, 10), (11, 12)]   # stamp pairs bracketing the MFMA loops
groups = collections.defaultdict(list)
for i, k in enumerate(key):
    groups[int(k)].append(i)
life_ov, gemm_ov, n = 0.0, 0.0, 0
conc = []
for k, idx in groups.items():
    idx = sorted(idx, key=lambda i: tt[i, 0])
    for a in idx:
        s0, s1 = tt[a, 0], tt[a, 13]
        others = [b for b in idx if b != a and tt[b, 0] < s1 and tt[b, 13] > s0]
        ov = sum(min(s1, tt[b, 13]) - max(s0, tt[b, 0]) for b in others)
        life_ov += ov / max(1, s1 - s0)
        conc.append(len(others))
        # overlap of my GEMM intervals with the others' GEMM intervals
        mine = [(tt[a, x], tt[a, y]) for x, y in GEMM]
        tot = sum(e - b for b, e in mine)
        g = 0
        for b_ in others:
            for x, y in GEMM:
                ob, oe = tt[b_, x], tt[b_, y]
                for mb, me in mine:
                    g += max(0, min(me, oe) - max(mb, ob))
        gemm_ov += g / max(1, tot)
        n += 1
print(f"mean fraction of a workgroup's lifetime overlapped by co-resident workgroups: {life_ov / n:.2f}")
print(f"mean number of overlapping workgroups: {np.mean(conc):.2f}")
print(f"mean fraction of a workgroup's GEMM time that coincides with a co-resident GEMM phase: {gemm_ov / n:.2f}")
