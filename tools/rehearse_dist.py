#!/usr/bin/env python3
"""Rehearsal of the multi-GPU library paths with several ranks sharing ONE GPU (gloo host staging):
    CODECAD_AMD_DIST_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 \
        --master-port 29512 tools/rehearse_dist.py
Each rank runs dist.mass_properties on its slices; rank 0 compares with the single-GPU driver."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import codecad_amd as cc  # noqa: E402
from codecad_amd import dist  # noqa: E402

rank, world = dist.init()
shape = cc.examples.sponge(3)
got = dist.mass_properties(shape, 1.0 / 243, grid_size=9)
dist.barrier()
if rank == 0:
    want = cc.mass_properties(shape, 1.0 / 243, grid_size=9)
    print("ranks %d: volume %.15f (single GPU %.15f, exact %.15f)" % (world, got.volume, want.volume, (20 / 27) ** 3))
    assert abs(got.volume - want.volume) <= 1e-13 * want.volume
    assert np.allclose(got.inertia_tensor, want.inertia_tensor, rtol=1e-11, atol=1e-16)
    assert abs(got.centroid - want.centroid) < 1e-14
leaves, info = dist.subdivision(shape, 1.0 / 243, grid_size=9)
if rank == 0:
    single = cc.subdivision.subdivision_device(shape, 1.0 / 243, grid_size=9)
    a = sorted(map(tuple, leaves.cpu().numpy()[:, :3].tolist()))
    b = sorted(map(tuple, single.int_corners().tolist()))
    print("leaf blocks: %d sharded, %d single GPU, levels %s" % (len(a), len(b), info["level_counts"]))
    assert a == b and list(info["level_counts"]) == list(single.level_counts)
    print("rehearsal ok")
dist.barrier()
if dist.exchanging():
    import torch.distributed
    torch.distributed.destroy_process_group()
