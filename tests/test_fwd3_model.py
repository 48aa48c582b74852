"""Lane-level model of k_aev_fwd3's bookkeeping (torchani_amd/csrc/aev.hip): slot dealing over the species-pair blocks of
an atom, the per-slot pair iterator, and the segmented reduction through LDS rows + DPP row scans.  The model restates the
kernel's integer logic statement by statement with 64-entry numpy arrays for the lanes; it is checked against brute force
(every pair of every block exactly once; block sums equal to the plain sums).  CPU only -- the GPU parity tests compare the
kernel's AEVs with the reference's."""
import itertools

import numpy as np
import pytest

WAVE = 64


def row_shr(v, n):
    out = np.zeros_like(v)
    for lane in range(WAVE):
        if lane % 16 >= n:
            out[lane] = v[lane - n]
    return out


def row_shl(v, n):
    out = np.zeros_like(v)
    for lane in range(WAVE):
        if lane % 16 + n < 16:
            out[lane] = v[lane + n]
    return out


def block_slots(np_b, I):
    inv = np.float32(1.0) / np.float32(I)
    c = (np.float32(np_b) + np.float32(0.5)) * inv
    c = c.astype(np.int32)
    c = c + (c * I < np_b)
    return (c + 3) & ~3


def wave_isum(v):
    v = v.copy()
    for n in (1, 2, 4, 8):
        v = v + row_shr(v, n)
    return int(v[15] + v[31] + v[47] + v[63])


def run_atom(counts, rng):
    """counts[t] = angular-range neighbors of species t.  Returns (pairs visited per block, reduced sums, expected sums)."""
    S = len(counts)
    lane = np.arange(WAVE)
    # block lanes 7 + P
    nd_tj = np.full(WAVE, 7)
    nd_tk = np.full(WAVE, 7)
    P = 0
    for tj in range(S):
        for tk in range(tj, S):
            nd_tj[7 + P], nd_tk[7 + P] = tj, tk
            P += 1
    cnt = np.array(list(counts) + [0] * (8 - S))
    pr = np.concatenate([[0], np.cumsum(cnt)[:-1]])
    cj, ck = cnt[nd_tj], cnt[nd_tk]
    same_b = nd_tj == nd_tk
    nd = np.where(same_b, cj >= 2, (cj >= 1) & (ck >= 1)) & (lane >= 7) & (lane < 7 + P)
    need = [int(b) for b in np.nonzero(nd)[0]]
    nA = int(cnt.sum())
    if not need:
        return {}, {}, {}
    np_b = np.where(nd, np.where(same_b, (cj * (cj - 1)) >> 1, cj * ck), 0)
    word_oj, word_ok = pr[nd_tj], pr[nd_tk]
    T = (nA * (nA - 1)) >> 1
    I = (T + 63) >> 6
    slots_b = block_slots(np_b, I)
    if wave_isum(slots_b) > 64:
        s1 = block_slots(np_b, I + 1)
        if wave_isum(s1) <= 64:
            I += 1
            slots_b = s1
    remaining = [b - 7 for b in need]
    visited = {b: [] for b in need}
    sums = {}
    expected = {b: np.zeros(32) for b in need}
    n_batches = 0
    while remaining:
        n_batches += 1
        s_run = 0
        myblk = np.zeros(WAVE, dtype=np.int64)
        mys0 = np.zeros(WAVE, dtype=np.int64)
        while remaining:
            Pb = remaining[0]
            ns = int(slots_b[7 + Pb])
            if s_run + ns > 64:
                break
            mine = lane >= s_run
            myblk = np.where(mine, 7 + Pb, myblk)
            mys0 = np.where(mine, s_run, mys0)
            s_run += ns
            remaining = remaining[1:]
        assert s_run > 0
        myblk = np.where(lane < s_run, myblk, 0)
        # per-lane block parameters (ds_bpermute from the block lane)
        same = same_b[myblk] & (myblk > 0)
        oj, ok, nj, nk = word_oj[myblk], word_ok[myblk], cj[myblk], ck[myblk]
        npl = np.where(myblk > 0, np.where(same, (nj * (nj - 1)) >> 1, nj * nk), 0)
        div = np.where(same, (nj - 1) >> 1, nk)
        rect = np.where(same, nj * div, 0x7FFFFFFF)
        half = nj >> 1
        t = (lane - mys0) * I
        inv_div = np.where(div > 0, np.float32(1.0) / np.maximum(div, 1).astype(np.float32), np.float32(0.0)).astype(np.float32)
        qd = ((t.astype(np.float32) + np.float32(0.5)) * inv_div).astype(np.int64)
        rem = t - qd * div
        acc = np.zeros((WAVE, 32))
        for _ in range(I):
            k2 = qd + 1 + rem
            k2 = np.where(k2 >= nj, k2 - nj, k2)
            diam = t >= rect
            jr = np.where(same & diam, t - rect, qd)
            kr = np.where(same, np.where(diam, t - rect + half, k2), rem)
            v = t < npl
            for ln in range(WAVE):
                if v[ln]:
                    ej, ek = int(oj[ln] + jr[ln]), int(ok[ln] + kr[ln])
                    b = int(myblk[ln])
                    # inside the groups of the block's two species
                    assert oj[ln] <= ej < oj[ln] + nj[ln] and ok[ln] <= ek < ok[ln] + (nj[ln] if same[ln] else nk[ln])
                    visited[b].append((ej, ek))
                    val = rng.standard_normal(32)
                    acc[ln] += val
                    expected[b] += val
            t = t + 1
            rem = rem + 1
            carry = rem >= div
            rem = np.where(carry, 0, rem)
            qd = qd + carry
        # ---- segmented reduction ----
        b_m4, b_m8, b_p4 = row_shr(myblk, 4), row_shr(myblk, 8), row_shl(myblk, 4)
        m1 = ((myblk > 0) & (b_m4 == myblk)).astype(float)
        m2 = ((myblk > 0) & (b_m8 == myblk)).astype(float)
        rowi = lane >> 4
        e_b = [int(myblk[15]), int(myblk[31]), int(myblk[47])]
        n_b = [int(myblk[16]), int(myblk[32]), int(myblk[48])]
        c = [((myblk > 0) & (rowi >= r + 1) & (e_b[r] == myblk)).astype(float) for r in range(3)]
        row_last = (lane & 15) >= 12
        nxt = np.where(row_last, np.where(rowi == 0, n_b[0], np.where(rowi == 1, n_b[1], np.where(rowi == 2, n_b[2], 0))), b_p4)
        blk_last = (myblk > 0) & (nxt != myblk)
        w4 = lane & 3
        for rr in range(2):
            red = acc[:, 16 * rr:16 * rr + 16]   # row = slot (lane), 16 values
            p = np.zeros((WAVE, 4))
            for ln in range(WAVE):
                g = ln & ~3
                p[ln] = sum(red[g + r, 4 * w4[ln]:4 * w4[ln] + 4] for r in range(4))
            for comp in range(4):
                p[:, comp] += m1 * row_shr(p[:, comp], 4)
            for comp in range(4):
                p[:, comp] += m2 * row_shr(p[:, comp], 8)
            X = np.zeros((3, 16))
            for ln in range(WAVE):
                if row_last[ln] and rowi[ln] < 3:
                    X[rowi[ln], 4 * w4[ln]:4 * w4[ln] + 4] = p[ln]
            for ln in range(WAVE):
                for r in range(3):
                    p[ln] += c[r][ln] * X[r, 4 * w4[ln]:4 * w4[ln] + 4]
            for ln in range(WAVE):
                if blk_last[ln]:
                    b = int(myblk[ln])
                    sums.setdefault(b, np.full(32, np.nan))
                    sums[b][16 * rr + 4 * w4[ln]:16 * rr + 4 * w4[ln] + 4] = p[ln]
    return visited, sums, expected, (cnt, pr, nd_tj, nd_tk, n_batches, I)


def check(counts, rng):
    res = run_atom(counts, rng)
    if not res[0]:
        return 0, 0
    visited, sums, expected, (cnt, pr, nd_tj, nd_tk, n_batches, I) = res
    for b, pairs in visited.items():
        tj, tk = nd_tj[b], nd_tk[b]
        if tj == tk:
            want = {frozenset((pr[tj] + x, pr[tj] + y)) for x, y in itertools.combinations(range(cnt[tj]), 2)}
            got = [frozenset(p) for p in pairs]
        else:
            want = {(pr[tj] + x, pr[tk] + y) for x in range(cnt[tj]) for y in range(cnt[tk])}
            got = pairs
        assert len(got) == len(want) and set(got) == want, (counts, b)
        assert np.allclose(sums[b], expected[b], atol=1e-9), (counts, b)
    return n_batches, I


def test_water_like_atoms_take_two_iterations():
    rng = np.random.default_rng(0)
    # H-centred atom of the water box: ~11 H and ~5 O inside 3.5 A
    assert check([11, 0, 0, 5], rng) == (1, 2)
    assert check([10, 0, 0, 6], rng) == (1, 2)


def test_every_pair_once_and_block_sums():
    rng = np.random.default_rng(1)
    cases = [[2], [3], [1, 1], [2, 1], [128], [64, 64], [127, 1], [5, 4, 3, 2, 1, 1, 1], [2, 2, 2, 2, 2, 2, 2],
             [1, 1, 1, 1, 1, 1, 1], [20, 16], [36, 0, 0, 0, 0, 0, 1], [9, 8, 7, 6, 5, 4, 3], [40, 30, 20, 10, 5, 5, 5]]
    for c in cases:
        check(c, rng)
    for _ in range(150):
        S = int(rng.integers(1, 8))
        tot = int(rng.integers(2, 60))
        w = rng.random(S) ** 2
        c = np.floor(w / w.sum() * tot).astype(int)
        if c.sum() < 2:
            c[0] += 2
        check([int(x) for x in c], rng)


def test_many_small_blocks_run_in_batches():
    rng = np.random.default_rng(2)
    n_batches, _ = check([2, 2, 2, 2, 2, 2, 2], rng)   # 28 blocks of 1-4 pairs: 28 x 4 padded slots > 64
    assert n_batches >= 2


def rad_positions(cA, cF):
    """The kernel's rad_pos (aev.hip): entry e of a row sorted {angular, far} x species goes to e + the shift of its segment;
    the radial list is then grouped by species, every group padded to a multiple of 8 entries."""
    S = len(cA)
    nA, nR = int(sum(cA)), int(sum(cA) + sum(cF))
    prA = np.concatenate([[0], np.cumsum(cA)[:-1]]).astype(int)
    prF = np.concatenate([[0], np.cumsum(cF)[:-1]]).astype(int)
    need = [t for t in range(S) if cA[t] + cF[t] > 0]
    e = np.arange(((nR + 63) // 64) * 64)
    shA, shF, base = np.zeros_like(e), np.zeros_like(e), 0
    groups = {}
    for t in need:                      # (scalar loop over the present species, ascending)
        oA, oF = prA[t], nA + prF[t]
        shA = np.where(e >= oA, base - oA, shA)
        shF = np.where(e >= oF, base + cA[t] - oF, shF)
        groups[t] = (base, cA[t] + cF[t])
        base += (cA[t] + cF[t] + 7) & ~7
    pos = e + np.where(e < nA, shA, shF)
    return pos[:nR], groups, base


@pytest.mark.parametrize("seed", range(40))
def test_radial_list_grouped_by_species(seed):
    rng = np.random.RandomState(seed)
    S = int(rng.randint(1, 8))
    dens = rng.choice([0.0, 0.3, 1.0], size=S)
    cA = (rng.poisson(6, size=S) * (rng.rand(S) < dens)).astype(int)
    cF = (rng.poisson(14, size=S) * (rng.rand(S) < dens)).astype(int)
    if cA.sum() + cF.sum() == 0:
        cF[0] = 1
    if cA.sum() > 128 or (cA + cF).sum() > 256:
        pytest.skip("beyond the row limits")
    pos, groups, total = rad_positions(cA, cF)
    nA = cA.sum()
    # species of every row entry: angular part sorted by species, then the far part sorted by species
    sp_of = np.concatenate([np.repeat(np.arange(S), cA), np.repeat(np.arange(S), cF)])
    assert len(np.unique(pos)) == len(pos) and pos.min() >= 0 and pos.max() < total   # a placement without collisions
    for t, (base, n) in groups.items():
        mine = np.sort(pos[sp_of == t])
        assert (mine == base + np.arange(n)).all()                # the species' entries fill the head of its group ...
        assert base % 8 == 0                                       # ... groups start on multiples of 8 (the 8-slot steps)
    assert total <= len(pos) + 7 * len(groups) and total % 8 == 0 and total <= 256 + 49
    # inside a group the angular entries come first, in row order (the order of summation of the radial sums is fixed)
    for t, (base, n) in groups.items():
        a = pos[:nA][sp_of[:nA] == t]
        f = pos[nA:][sp_of[nA:] == t]
        assert (np.diff(a) == 1).all() and (np.diff(f) == 1).all()
        if len(a) and len(f):
            assert a.max() + 1 == f.min()
