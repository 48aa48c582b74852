"""Host-side cost of one graph-replayed config-2 step (development): where the wall time of ANI.graphed(...)(coords) goes."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.bench_configs import GOLD  # noqa: E402
from torchani_amd.models import ANI2x  # noqa: E402

dev = torch.device("cuda:0")
with np.load(os.path.join(GOLD, "cfg2_xyz13_28_ani2x.npz")) as z:
    sp, x = z["species"].astype(np.int64), z["coords"]
model = ANI2x(seed=0, device=dev, periodic_table_index=False, neighborlist="batch")
spd, xd = torch.from_numpy(sp).to(dev), torch.from_numpy(x).to(dev)
model.auto_graph_atoms = 0
g = model.graphed(spd, xd)


def loop(fn, reps=200, sync_each=False):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
        if sync_each:
            torch.cuda.synchronize()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / reps * 1e6, (t2 - t0) / reps * 1e6


print("full call        host %.1f us / step, wall %.1f us / step" % loop(lambda: g(xd)))
print("full call, sync each step: %.1f us" % loop(lambda: g(xd), sync_each=True)[1])
print("graph.replay()   host %.1f us, wall %.1f us" % loop(lambda: g.graph.replay()))
print("coords.copy_     host %.1f us, wall %.1f us" % loop(lambda: g.coords.copy_(xd)))
print("_pack check      host %.1f us" % loop(lambda: model.neural_networks._pack(dev))[0])
print("_current_sae     host %.1f us" % loop(lambda: g._current_sae())[0])
print("graph nodes: see tools/gpu_cfg2_timeline.sh (AB_GRAPH=1)")
