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


def _pipeline(tape, levels, resolution, origin, n_objects, capacities, replicate=0):
    """The traversal through dist.LevelPipeline (lists with their lengths in header rows, one fixed-size
    all-gather per level, the balanced share taken without the host looking at any count), classification by
    the CPU oracle.  -> (this rank's share of the leaves, global survivors per level)"""
    import math
    import oracle
    from codecad_amd import dist

    def classify(level, parents, n_parents, max_parents, out, own=None):
        int_step, dims = levels[level]
        k = min(int(n_parents.item()), max_parents)   # an overflowed list holds `capacity` rows: the launch covers no more
        rows = []
        for ix, iy, iz, obj in parents[:k].tolist():
            half = int_step / 2
            corner = [(ix + half) * resolution + origin[0], (iy + half) * resolution + origin[1],
                      (iz + half) * resolution + origin[2]]
            step = int_step * resolution
            n, cells = oracle.subdivision_step(tape, np.array(corner).astype(np.float32), np.float32(step),
                                               np.float32(step * math.sqrt(3) / 2), dims)
            for i, j, k_, _ in cells.tolist():
                # a replicated level (hu_subdivision_level_owned): of every parent's cells a rank lists those it owns
                if own is not None and dist.owner_of([ix, iy, iz, obj], k_ + dims[2] * (j + dims[1] * i), own[0]) != own[1]:
                    continue
                rows.append([ix + i * int_step, iy + j * int_step, iz + k_ * int_step, obj])
        capacity = out.shape[0] - 1
        out[0, 0] = len(rows)                       # like the kernel: the counter counts everything ...
        rows = rows[:capacity]                      # ... and only what fits is stored
        if rows:
            out[1:1 + len(rows)] = torch.tensor(rows, dtype=torch.int32)

    top = torch.zeros((n_objects, 4), dtype=torch.int32)
    top[:, 3] = torch.arange(n_objects, dtype=torch.int32)
    pipe = dist.LevelPipeline(top, capacities, classify, replicate=replicate)
    mine = pipe.enqueue()
    totals = pipe.check()
    return mine[1:1 + int(mine[0, 0])], totals


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


def _integrate_pipeline(tape, box_a, levels, capacities, replicate=0):
    """The same integration through dist.LevelPipeline as dist.MassPipeline drives it on the GPU: 32-byte rows of four
    doubles seen as eight int32 words, the count in the header's first word, every level (the leaf level too) a
    classify call, nothing looked at by the host until check().  -> the ten integrals, all-reduced."""
    import math
    import oracle
    from codecad_amd import dist
    from codecad_amd.mass_properties import integrals_host, _KEYS

    partial = torch.zeros(10, dtype=torch.float64)

    def owned_kernel(row_words, shifted, s, thr, dims, own):
        """hu_mass_properties_level_owned restated over the oracle's distances: the kernel of mass_properties.cl:7-56 with
        every cell that another rank owns left out of the sums and of the list"""
        w = oracle.grid_eval(tape, shifted.astype(np.float32), np.float32(s), dims)[..., 3]
        idx = np.indices(dims).reshape(3, -1).T
        lin = idx[:, 2] + dims[2] * (idx[:, 1] + dims[1] * idx[:, 0])
        mine = np.array([dist.owner_of(row_words, c, own[0]) == own[1] for c in lin.tolist()])
        wf = w.reshape(-1)
        inside = mine & (wf <= -np.float32(thr))
        amb = mine & ~(wf <= -np.float32(thr)) & (wf < np.float32(thr))
        x, y, z = (idx[inside, c].astype(np.uint64) for c in range(3))
        su = np.array([np.sum(x * x), np.sum(x * y), np.sum(x * z), np.sum(x), np.sum(y * y), np.sum(y * z), np.sum(y), np.sum(z * z),
                       np.sum(z), inside.sum()], dtype=np.uint64).astype(np.uint32)
        return su, int(amb.sum()), np.concatenate([idx[amb], np.zeros((int(amb.sum()), 1), dtype=idx.dtype)], axis=1)

    def classify(level, parents, n_parents, max_parents, out, own=None):
        s, dims = levels[level]
        leaf = level + 1 == len(levels)
        thr = 0.0 if leaf else s * math.sqrt(3) / 2
        k = min(int(n_parents.item()), max_parents)
        words = parents[:k].contiguous()
        rows = words.view(torch.float64).reshape(-1, 4)
        children, corners, sums = [], [], []
        for (cx, cy, cz, tag), rw in zip(rows.tolist(), words.view(torch.int32).reshape(-1, 8).numpy().view(np.uint32)):
            shifted = np.array([cx + s / 2, cy + s / 2, cz + s / 2])
            if own is None:
                su, n, cells = oracle.mass_properties(tape, shifted.astype(np.float32), np.float32(s), np.float32(thr), dims)
            else:   # (kernels.hpp k_classify: lo ^ hi of each of the row's four doubles)
                su, n, cells = owned_kernel([rw[0] ^ rw[1], rw[2] ^ rw[3], rw[4] ^ rw[5], rw[6] ^ rw[7]], shifted, s, thr, tuple(int(v) for v in dims), own)
            corners.append([cx, cy, cz])
            sums.append(su)
            if not leaf:
                children += [[i * s + cx, j * s + cy, k_ * s + cz, tag] for i, j, k_, _ in cells.tolist()]
        if corners:
            d = integrals_host(np.array(sums, dtype=np.uint32), np.array(corners, dtype=np.float64), s)
            partial.add_(torch.tensor([d[key] for key in _KEYS], dtype=torch.float64))
        capacity = out.shape[0] - 1
        out[0, 0] = len(children)
        children = children[:capacity]
        if children:
            out[1:1 + len(children)] = torch.tensor(children, dtype=torch.float64).view(torch.int32).reshape(-1, 8)

    top = torch.tensor([[box_a[0], box_a[1], box_a[2], 0.0]], dtype=torch.float64).view(torch.int32).reshape(1, 8)
    pipe = dist.LevelPipeline(top, list(capacities) + [0], classify, replicate=replicate)
    pipe.enqueue()
    totals = pipe.check()
    return dist.allreduce_sum(partial), totals[:-1]


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
    # ONE object over both ranks (strong scaling, what bench.py --gpus N runs) through the device-side protocol:
    # the shares tile the single-rank leaf list, the per-level totals are the single-rank counts
    one, one_counts = _traverse(tape, levels, res, origin, n_objects=1)
    share, totals = _pipeline(tape, levels, res, origin, 1, [max(one_counts) + 3] * len(one_counts))
    assert totals == one_counts
    b, e = dist.balanced_slice(one.shape[0], rank, world)
    assert e - b == share.shape[0]
    shares = dist.allgather_rows(share)
    assert sorted(map(tuple, shares.tolist())) == sorted(map(tuple, one.tolist()))
    # the same with REPLICATED levels: every rank classifies the first k levels in full -- no exchange -- and in the last of
    # them lists only the cells it owns: same totals, the shares still partition the leaf list; with every level replicated
    # the traversal needs no collective at all (the bench's hierarchy), and the owners' shares are balanced
    for k in range(1, len(one_counts) + 1):
        share_r, totals_r = _pipeline(tape, levels, res, origin, 1, [max(one_counts) + 3] * len(one_counts), replicate=k)
        assert totals_r == one_counts, (k, totals_r, one_counts)
        shares_r = dist.allgather_rows(share_r)
        assert sorted(map(tuple, shares_r.tolist())) == sorted(map(tuple, one.tolist())), k
        if k == len(one_counts):
            assert abs(share_r.shape[0] - one.shape[0] / world) <= 0.2 * one.shape[0] / world + 2
    # a list that outgrows its capacity is reported with the sizes that would have sufficed -- on every rank
    try:
        _pipeline(tape, levels, res, origin, 1, [2] * len(one_counts))
        raise AssertionError("overflow went unnoticed")
    except dist.Overflow as err:
        assert err.needed[0] > 2
    mtape, box_a, mlevels = _setup_mass()
    integrals = _integrate(mtape, box_a, mlevels)
    # the device-counted form (what dist.mass_properties runs): same integrals, no count seen by the host on the way
    piped, ambiguous = _integrate_pipeline(mtape, box_a, mlevels, [800] * (len(mlevels) - 1))
    assert np.allclose(piped.tolist(), integrals.tolist(), rtol=1e-13, atol=1e-15) and all(a > 0 for a in ambiguous)
    # ... and with its first levels replicated (the last of them summing and listing owned cells only): no all-gather, same integrals
    for k in range(1, len(mlevels)):
        piped_r, ambiguous_r = _integrate_pipeline(mtape, box_a, mlevels, [800] * (len(mlevels) - 1), replicate=k)
        assert np.allclose(piped_r.tolist(), integrals.tolist(), rtol=1e-13, atol=1e-15) and ambiguous_r == ambiguous, k
    try:
        _integrate_pipeline(mtape, box_a, mlevels, [1] * (len(mlevels) - 1))
        raise AssertionError("overflow went unnoticed")
    except dist.Overflow:
        pass
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


def test_pipeline_single_rank_and_slice_rule():
    """World 1: the pipeline is the plain traversal.  And the share rule of slice_rows_reference (= hu_slice_rows)
    is dist.balanced_slice on the concatenation, with truncated pieces and shares flagged."""
    from codecad_amd import dist
    tape, levels, res, origin = _setup()
    leaves, counts = _traverse(tape, levels, res, origin, n_objects=1)
    share, totals = _pipeline(tape, levels, res, origin, 1, [c + 1 for c in counts])
    assert totals == counts and sorted(map(tuple, share.tolist())) == sorted(map(tuple, leaves.tolist()))
    with pytest.raises(dist.Overflow) as info:
        _pipeline(tape, levels, res, origin, 1, [counts[0] - 1] + [c + 1 for c in counts[1:]])
    assert info.value.needed[0] == counts[0]

    rng = np.random.default_rng(3)
    for world in (1, 2, 3, 8):
        piece_rows = 7
        counts = rng.integers(0, piece_rows, world).tolist()
        gathered = torch.zeros((world, piece_rows, 4), dtype=torch.int32)
        rows = []
        for r, c in enumerate(counts):
            gathered[r, 0, 0] = c
            gathered[r, 1:1 + c] = torch.arange(c * 4, dtype=torch.int32).reshape(c, 4) + 1000 * r
            gathered[r, 1 + c:] = -1     # stale rows behind the count must not leak
            rows += gathered[r, 1:1 + c].tolist()
        got = []
        for rank in range(world):
            out = torch.full((piece_rows, 4), -7, dtype=torch.int32)
            stats = torch.zeros(2, dtype=torch.int32)
            dist.slice_rows_reference(gathered, rank, out, stats)
            b, e = dist.balanced_slice(len(rows), rank, world)
            assert int(out[0, 0]) == e - b and stats.tolist() == [len(rows), 0]
            got += out[1:1 + e - b].tolist()
        assert got == rows
    # a piece whose header claims more rows than it holds, and a share larger than the output
    gathered = torch.zeros((2, 3, 4), dtype=torch.int32)
    gathered[:, 0, 0] = torch.tensor([5, 2])
    out, stats = torch.zeros((3, 4), dtype=torch.int32), torch.zeros(2, dtype=torch.int32)
    dist.slice_rows_reference(gathered, 0, out, stats)
    assert stats.tolist() == [4, 1] and int(out[0, 0]) == 2
    gathered[:, 0, 0] = torch.tensor([2, 2])
    out = torch.zeros((2, 4), dtype=torch.int32)
    dist.slice_rows_reference(gathered, 1, out, stats)
    assert stats.tolist() == [4, 1] and int(out[0, 0]) == 1


def test_mass_pipeline_protocol_single_rank():
    """World 1: the device-counted integration (32-byte rows, leaf level included) equals the level-by-level one."""
    mtape, box_a, mlevels = _setup_mass()
    single = _integrate(mtape, box_a, mlevels).tolist()
    piped, ambiguous = _integrate_pipeline(mtape, box_a, mlevels, [800] * (len(mlevels) - 1))
    assert np.allclose(piped.tolist(), single, rtol=1e-13, atol=1e-15)
    assert piped[0].item() == pytest.approx((20 / 27) ** 2, rel=1e-12)


def test_single_process_helpers():
    from codecad_amd import dist
    assert dist.rank_world() == (0, 1)
    assert dist.balanced_slice(10, 0, 1) == (0, 10)
    assert [dist.balanced_slice(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert [dist.x_slab(512, r, 8) for r in (0, 7)] == [(0, 64), (448, 512)]
    t = torch.arange(6).reshape(3, 2)
    assert dist.allgather_rows(t) is t
