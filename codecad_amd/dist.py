"""Multi-GPU sharding of the hot path: one process per GPU, torch.distributed over RCCL.

The reference has no multi-device code at all (one context, one queue:
reference cl_util/opencl_manager.py:89-98).  What shards, and how:

  * dense grid_eval: voxels are independent -> contiguous x-slabs (x is the slowest index,
    reference cl_util/indexing.h:4), each rank writes its own slab, NO collective;
  * subdivision / mass_properties: cells of one level are independent, levels are not ->
    the global parent list of a level is cut into `world` balanced contiguous slices; each
    rank classifies its slice and compacts its survivors locally (wavefront ballot scan);
    between levels the survivor lists are exchanged with ONE variable-length all-gather
    (counts first, then rows padded to the longest), so every rank holds the whole next
    parent list and load stays balanced wherever the surface lies.  Lists are a few MB at
    most: the exchange is latency-bound on xGMI, far below the per-link bandwidth.

Everything here works on torch tensors of any device, so the same code runs on `gloo`/CPU
(tests/test_dist_gloo.py, world_size 2) and on `nccl` (= RCCL) with one GPU per rank.
"""
import os

import torch
import torch.distributed as dist


def init(backend=None):
    """Join the process group described by RANK/WORLD_SIZE/MASTER_* (torchrun).  Returns
    (rank, world).  Single process when WORLD_SIZE is unset or 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1
    if not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("CODECAD_AMD_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_device())
        dist.init_process_group(backend=backend)
    return dist.get_rank(), dist.get_world_size()


def local_device():
    """GPU ordinal of this rank: LOCAL_RANK, wrapped onto the visible devices (so that a
    multi-rank rehearsal can share one GPU with the gloo backend)."""
    n = max(1, torch.cuda.device_count())
    return int(os.environ.get("LOCAL_RANK", "0")) % n


def _host_staged():
    """gloo cannot gather device tensors: stage through the host (rehearsal path only)."""
    return dist.get_backend() == "gloo"


def rank_world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def balanced_slice(total, rank, world):
    """[begin, end) of the rank's share of `total` items; sizes differ by at most one."""
    base, extra = divmod(total, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def x_slab(sx, rank, world):
    """The x range a rank owns of a dense grid with sx planes (same rule)."""
    return balanced_slice(sx, rank, world)


def allgather_rows(rows, group=None):
    """Variable-length all-gather: every rank passes a (n_i, k) tensor and gets the
    concatenation over ranks in rank order.  Two collectives: counts, then padded rows."""
    rank, world = rank_world()
    if world == 1:
        return rows
    device = rows.device
    if device.type != "cpu" and _host_staged():
        return allgather_rows(rows.cpu(), group).to(device)
    n = torch.tensor([rows.shape[0]], dtype=torch.int64, device=rows.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    longest = max(counts)
    if longest == 0:
        return rows[:0]
    padded = torch.zeros((longest,) + tuple(rows.shape[1:]), dtype=rows.dtype, device=rows.device)
    padded[:rows.shape[0]] = rows
    gathered = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(gathered, padded, group=group)
    out = torch.cat([g[:c] for g, c in zip(gathered, counts)], dim=0)
    if out.is_cuda:
        # the list is handed to kernels launched through the C ABI by data_ptr(): make sure the
        # collective and the concatenation have finished, whatever stream they ran on
        torch.cuda.current_stream(out.device).synchronize()
    return out


def allreduce_sum(t, group=None):
    _, world = rank_world()
    if world > 1:
        if t.device.type != "cpu" and _host_staged():
            c = t.cpu()
            dist.all_reduce(c, op=dist.ReduceOp.SUM, group=group)
            t.copy_(c)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def allreduce_max(t, group=None):
    _, world = rank_world()
    if world > 1:
        if t.device.type != "cpu" and _host_staged():
            c = t.cpu()
            dist.all_reduce(c, op=dist.ReduceOp.MAX, group=group)
            t.copy_(c)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return t


def barrier():
    _, world = rank_world()
    if world > 1:
        dist.barrier()


def run_levels(top_parents, n_levels, classify_level):
    """Level-synchronous traversal with a balanced slice per rank and one exchange per level.

    top_parents: (n, k) tensor, identical on every rank (the global top-level parent list).
    classify_level(level, my_parents) -> (m, k) tensor of that slice's surviving children.
    Returns (global leaf list, [global survivor count per level]).
    """
    rank, world = rank_world()
    parents, counts = top_parents, []
    for level in range(n_levels):
        begin, end = balanced_slice(parents.shape[0], rank, world)
        mine = classify_level(level, parents[begin:end])
        parents = allgather_rows(mine)
        counts.append(int(parents.shape[0]))
        if parents.shape[0] == 0:
            break
    return parents, counts
