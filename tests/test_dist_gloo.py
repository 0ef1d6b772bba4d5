"""world_size-2 `gloo` run of the sharding logic used on N GPUs (codecad_amd/dist.py):
balanced slices, the variable-length all-gather, and a full level-synchronous traversal in
which the classification is done by the CPU oracle -- the 2-rank result must equal the
1-rank result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _traverse(tape, levels, resolution, origin, n_objects, hints=None):
    import math
    import oracle
    from codecad_amd import dist

    def classify(level, parents):
        int_step, dims = levels[level]
        out = []
        for ix, iy, iz, obj in parents.tolist():
            half = int_step / 2
            corner = [(ix + half) * resolution + origin[0], (iy + half) * resolution + origin[1],
                      (iz + half) * resolution + origin[2]]
            step = int_step * resolution
            n, cells = oracle.subdivision_step(tape, np.array(corner).astype(np.float32), np.float32(step),
                                               np.float32(step * math.sqrt(3) / 2), dims)
            for i, j, k, _ in cells.tolist():
                out.append([ix + i * int_step, iy + j * int_step, iz + k * int_step, obj])
        return torch.tensor(out, dtype=torch.int32).reshape(-1, 4)

    top = torch.zeros((n_objects, 4), dtype=torch.int32)
    top[:, 3] = torch.arange(n_objects, dtype=torch.int32)
    return dist.run_levels(top, len(levels) - 1, classify, hints=hints)


def _integrate(tape, box_a, levels):
    """Sharded mass-properties integration with the oracle's kernel as the per-level worker."""
    import math
    import oracle
    from codecad_amd import dist
    from codecad_amd.mass_properties import integrals_host, _KEYS

    def level_fn(level, parents):
        s, dims = levels[level]
        leaf = level + 1 == len(levels)
        thr = 0.0 if leaf else s * math.sqrt(3) / 2
        children, corners, sums = [], [], []
        for cx, cy, cz, tag in parents.tolist():
            shifted = np.array([cx + s / 2, cy + s / 2, cz + s / 2])
            su, n, cells = oracle.mass_properties(tape, shifted.astype(np.float32), np.float32(s), np.float32(thr), dims)
            corners.append([cx, cy, cz])
            sums.append(su)
            if not leaf:
                for i, j, k, _ in cells.tolist():
                    children.append([i * s + cx, j * s + cy, k * s + cz, tag])
        part = torch.zeros(10, dtype=torch.float64)
        if corners:
            d = integrals_host(np.array(sums, dtype=np.uint32), np.array(corners, dtype=np.float64), s)
            part = torch.tensor([d[k] for k in _KEYS], dtype=torch.float64)
        return torch.tensor(children, dtype=torch.float64).reshape(-1, 4), part

    top = torch.tensor([[box_a[0], box_a[1], box_a[2], 0.0]], dtype=torch.float64)
    return dist.integrate_levels(top, len(levels), level_fn)


def _setup_mass():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from codecad_amd import examples, nodes, subdivision
    shape = examples.sponge(2)
    res = 1.0 / 27
    box = shape.bounding_box()
    levels = [(res * c, tuple(int(v) for v in d)) for c, d in subdivision.calculate_block_sizes(box, 3, res, 3, False)]
    return nodes.make_program(shape), tuple(box.a), levels


def _setup():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from codecad_amd import examples, nodes, subdivision
    shape = examples.sponge(2)
    res = 1.0 / 54
    box = shape.bounding_box().expanded_additive(res / 2)
    levels = [(c, tuple(int(v) for v in d)) for c, d in subdivision.calculate_block_sizes(box, 3, res, 4, True)]
    return nodes.make_program(shape), levels, res, tuple(box.a)


def _worker(rank, world, port, queue):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    tape, levels, res, origin = _setup()
    from codecad_amd import dist
    r, w = dist.init(backend="gloo")
    assert (r, w) == (rank, world)
    # balanced slices tile the range
    covered = []
    for k in range(world):
        b, e = dist.balanced_slice(11, k, world)
        covered += list(range(b, e))
    assert covered == list(range(11))
    # variable-length all-gather, including an empty contribution
    mine = torch.arange(rank * 3 * 2, dtype=torch.int32).reshape(-1, 2) + 100 * rank
    allrows = dist.allgather_rows(mine)
    want = torch.cat([torch.arange(k * 3 * 2, dtype=torch.int32).reshape(-1, 2) + 100 * k for k in range(world)])
    assert torch.equal(allrows, want)
    # the single-collective form: exact hint, generous hint, and a hint that is too small (falls back)
    for hint in (6, 50, 2):
        assert torch.equal(dist.allgather_rows(mine, hint=hint), want)
    leaves, counts = _traverse(tape, levels, res, origin, n_objects=world)
    # repeating the traversal with remembered hints gives the same result through one collective per level
    hints = []
    first = _traverse(tape, levels, res, origin, n_objects=world, hints=hints)
    again = _traverse(tape, levels, res, origin, n_objects=world, hints=hints)
    assert len(hints) == len(counts) and all(h is not None for h in hints)
    assert torch.equal(first[0], leaves) and torch.equal(again[0], leaves) and again[1] == counts
    total = dist.allreduce_sum(torch.tensor([leaves.shape[0]], dtype=torch.int64))
    assert int(total.item()) == world * leaves.shape[0]
    mtape, box_a, mlevels = _setup_mass()
    integrals = _integrate(mtape, box_a, mlevels)
    if rank == 0:
        queue.put((leaves.numpy().tolist(), counts, integrals.tolist()))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_traversal_equals_single_rank():
    world = 2
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, queue)) for r in range(world)]
    for p in procs:
        p.start()
    got_leaves, got_counts, got_integrals = queue.get(timeout=180)
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0

    tape, levels, res, origin = _setup()
    for k in ("RANK", "WORLD_SIZE"):
        os.environ.pop(k, None)
    leaves, counts = _traverse(tape, levels, res, origin, n_objects=world)
    assert counts == got_counts
    assert sorted(map(tuple, leaves.numpy().tolist())) == sorted(map(tuple, got_leaves))
    # both objects have identical hierarchies
    per_obj = [sorted(t[:3] for t in got_leaves if t[3] == k) for k in range(world)]
    assert per_obj[0] == per_obj[1] and len(per_obj[0]) > 0
    # the sharded integration (slices + all-gather of ambiguous cells + one all-reduce) equals the 1-rank one
    mtape, box_a, mlevels = _setup_mass()
    single = _integrate(mtape, box_a, mlevels).tolist()
    assert np.allclose(got_integrals, single, rtol=1e-13, atol=1e-15)   # first moments are 0 up to rounding noise
    assert got_integrals[0] == pytest.approx((20 / 27) ** 2, rel=1e-12)   # the sponge's exact volume at this resolution


def test_single_process_helpers():
    from codecad_amd import dist
    assert dist.rank_world() == (0, 1)
    assert dist.balanced_slice(10, 0, 1) == (0, 10)
    assert [dist.balanced_slice(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert [dist.x_slab(512, r, 8) for r in (0, 7)] == [(0, 64), (448, 512)]
    t = torch.arange(6).reshape(3, 2)
    assert dist.allgather_rows(t) is t
