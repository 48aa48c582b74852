"""The atom lists of the workgroup-level queues (torchani_amd/csrc/aev.hip AtomQueue, nbr.hip k_nbr_cell2) restated in
numpy: workgroup b of B owns the 16-atom groups b, b + B, ...; position p of its list is atom
first + 16 b + (p // 16) * 16 B + p % 16.  Whatever order the waves of a workgroup draw positions in, every atom of
lo..hi is visited exactly once, a workgroup's positions are monotonic in the atom index (so a wave may stop at the first
position past the end), and the XCD-aware block order keeps a workgroup's groups where the fixed-share loop had them."""
import numpy as np
import pytest

QW = 16


def xcd_block(b, nb):
    return b if nb & 7 else (b & 7) * (nb >> 3) + (b >> 3)


def atom(first, block, blocks, p):
    return first + block * QW + (p // QW) * blocks * QW + p % QW


@pytest.mark.parametrize("n,blocks", [(1, 1), (15, 1), (16, 1), (17, 2), (1000, 63), (4096, 256), (9125 * 256 + 64, 256),
                                      (5248, 256), (100000, 256)])
@pytest.mark.parametrize("lo", [0, 37])
def test_every_atom_exactly_once(n, blocks, lo):
    hi = lo + n
    seen = np.zeros(n, dtype=np.int32)
    for b in range(blocks):
        xb = xcd_block(b, blocks)
        p = np.arange(0, (n // (blocks * QW) + 2) * QW)
        a = atom(lo, xb, blocks, p)
        assert (np.diff(a) > 0).all()                      # monotonic: the first position past the end ends the wave
        a = a[a < hi]
        np.add.at(seen, a - lo, 1)
    assert (seen == 1).all()


def test_first_three_positions_are_the_waves_own():
    # a wave starts with positions wib, 16 + wib, 32 + wib (its header / entry prefetch pipeline is three atoms deep) and the
    # counter starts at 48: the queue hands out 48, 49, ...
    blocks = 256
    own = {atom(0, 5, blocks, k * QW + w) for w in range(QW) for k in range(3)}
    nxt = atom(0, 5, blocks, 3 * QW)
    assert len(own) == 48 and nxt not in own and nxt == 5 * QW + 3 * blocks * QW
