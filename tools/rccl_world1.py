"""Development: execute the RCCL calls of the multi-GPU step on ONE GPU (process group of world size 1, collectives forced
by TORCHANI_AMD_FORCE_GROUP=1): spatial-shard exchange (fp32 words and the int64 fixed-point variant), the gather of owned
rows, and the index-range all-reduce, each compared with the plain single-GPU result.

    TORCHANI_AMD_FORCE_GROUP=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 \\
        --master-port 29512 tools/rccl_world1.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import water_box  # noqa: E402


def main():
    from torchani_amd.models import ANI2x
    from torchani_amd.parallel import FORCE_COLLECTIVES, init_from_env

    assert FORCE_COLLECTIVES, "run with TORCHANI_AMD_FORCE_GROUP=1"
    rank, world, local, group = init_from_env()
    assert group is not None and torch.distributed.get_backend(group) == "nccl"
    dev = torch.device("cuda", local)
    sp_np, x_np, cell_np = water_box(16)
    sp, x, cell = torch.from_numpy(sp_np).to(dev), torch.from_numpy(x_np).to(dev), torch.from_numpy(cell_np).to(dev)
    pbc = (True, True, True)
    model = ANI2x(seed=0, device=dev, periodic_table_index=False, neighborlist="cell", row_capacity=128)
    ref = model.energies_and_forces(sp, x, cell, pbc, stress=True)
    for part in ("spatial", "index"):
        for fixed in (False, True):
            model.partition = part
            model.deterministic_forces = fixed
            out = model.energies_and_forces(sp, x, cell, pbc, group=group, reduce_forces=True, stress=True)
            torch.cuda.synchronize()
            dE = abs(float(out.energies - ref.energies))
            dF = float((out.forces - ref.forces).abs().max())
            dW = float((out.virial - ref.virial).abs().max())
            lc = model.last_collective
            print(f"partition {part:7s} fixed_point {fixed!s:5s}: |dE| {dE:.2e}  max|dF| {dF:.2e}  max|dW| {dW:.2e}  "
                  f"collectives {lc['collectives_per_step']} bytes {lc['bytes']}")
            assert dE < 1e-6 * sp.numel() and dF < 2e-5 and dW < 1e-4, (dE, dF, dW)
    objs = [None]
    torch.distributed.all_gather_object(objs, {"rank": rank}, group=group)
    torch.distributed.barrier(group)
    from torchani_amd.parallel import exchange_transport

    tr = exchange_transport()
    print("RCCL", ".".join(str(v) for v in torch.cuda.nccl.version()), "world-1 collectives ok", objs,
          "transport", tr["transport"], "fell back:", tr["fell_back"])
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
