"""Golden fixtures with the reference's smooth cutoff (CutoffSmooth order 2, eps 1e-10; cutoffs.py:84-101):
same recipe as gen_golden.py with ``set_global_cutoff_fn("smooth")``.

    python tests/golden/gen_golden_smooth.py      (needs /root/reference; the outputs are committed)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as gg  # noqa: E402


def main():
    gg.CUTOFF_FN = "smooth"
    gg.torch.set_num_threads(8)
    z, x, _, _ = gg.xyz(f"{gg.RES}/small.xyz")
    gg.run_case("small_smooth_ani2x", "ani2x", 26, z, x, aev_rows=np.arange(0, 264, 6))
    z, x, cell, pbc = gg.xyz(f"{gg.RES}/water-0.8nm.xyz")
    gg.run_case("water_pbc_smooth_ani2x", "ani2x", 24, z, x, cell, pbc)
    rs = np.random.RandomState(31)
    Z, X = gg.random_molecules(rs, 4, 12, 7, 4.0, 0.75, pad_prob=0.3)
    gg.run_case("rand_batch_smooth_ani2x", "ani2x", 25, Z, X)


if __name__ == "__main__":
    main()
