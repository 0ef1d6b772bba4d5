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


def force_collectives():
    """CODECAD_AMD_FORCE_COLLECTIVES=1: a single rank still joins a process group and walks the multi-rank path --
    fixed-size all-gather of its one piece, hu_slice_rows, indirect launches, the final all-reduce -- so that every
    line of the N > 1 code runs on the one GPU a developer has (same results as the plain path; RCCL is the backend)."""
    return os.environ.get("CODECAD_AMD_FORCE_COLLECTIVES", "0") == "1"


def exchanging():
    """Do the level exchanges go through collectives?  (more than one rank, or forced with a process group up)"""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or force_collectives()


def init(backend=None):
    """Join the process group described by RANK/WORLD_SIZE/MASTER_* (torchrun).  Returns
    (rank, world).  Single process when WORLD_SIZE is unset or 1 (unless collectives are forced: then a
    one-rank group is created, on 127.0.0.1 with a free port when no launcher set the rendezvous up)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 and not force_collectives():
        return 0, 1
    if world <= 1 and "MASTER_ADDR" not in os.environ:
        import socket
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK=os.environ.get("LOCAL_RANK", "0"))
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


def allgather_rows(rows, group=None, hint=None, return_counts=False):
    """Variable-length all-gather: every rank passes a (n_i, k) tensor and gets the concatenation over
    ranks in rank order.

    Two collectives (counts, then rows padded to the longest) -- or ONE when the caller passes `hint`, an
    upper bound on every rank's row count it expects to hold (e.g. the counts of the previous, identical
    step): rows are padded to `hint` and the true count rides in an extra header row; if some rank has more
    rows than the hint, every rank sees that in the gathered headers and all fall back to the two-step
    form together.  With return_counts the per-rank row counts come back too: (rows, [n_0, ..])."""
    rank, world = rank_world()
    if world == 1:
        return (rows, [int(rows.shape[0])]) if return_counts else rows
    device = rows.device
    if device.type != "cpu" and _host_staged():
        out, counts = allgather_rows(rows.cpu(), group, hint, True)
        return (out.to(device), counts) if return_counts else out.to(device)
    k = tuple(rows.shape[1:])
    if hint is not None and len(k) == 1 and not rows.dtype.is_floating_point:
        hint = int(hint)
        packed = torch.zeros((hint + 1,) + k, dtype=rows.dtype, device=device)
        packed[0, 0] = rows.shape[0]
        fits = rows.shape[0] <= hint
        if fits:
            packed[1:1 + rows.shape[0]] = rows
        gathered = [torch.empty_like(packed) for _ in range(world)]
        dist.all_gather(gathered, packed, group=group)
        counts = torch.stack([g[0, 0] for g in gathered]).tolist()   # one synchronisation
        if max(counts) <= hint:
            out = torch.cat([g[1:1 + c] for g, c in zip(gathered, counts)], dim=0)
            if out.is_cuda:
                torch.cuda.current_stream(out.device).synchronize()
            return (out, counts) if return_counts else out
        # someone overflowed the hint (every rank knows): the general path below
    n = torch.tensor([rows.shape[0]], dtype=torch.int64, device=rows.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = torch.cat(counts).tolist()   # one synchronisation, not one per rank
    longest = max(counts)
    if longest == 0:
        return (rows[:0], counts) if return_counts else rows[:0]
    padded = torch.zeros((longest,) + k, dtype=rows.dtype, device=rows.device)
    padded[:rows.shape[0]] = rows
    gathered = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(gathered, padded, group=group)
    out = torch.cat([g[:c] for g, c in zip(gathered, counts)], dim=0)
    if out.is_cuda:
        # the list is handed to kernels launched through the C ABI by data_ptr(): make sure the
        # collective and the concatenation have finished, whatever stream they ran on
        torch.cuda.current_stream(out.device).synchronize()
    return (out, counts) if return_counts else out


def allreduce_sum(t, group=None):
    if exchanging():
        if t.device.type != "cpu" and _host_staged():
            c = t.cpu()
            dist.all_reduce(c, op=dist.ReduceOp.SUM, group=group)
            t.copy_(c)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def allreduce_max(t, group=None):
    if exchanging():
        if t.device.type != "cpu" and _host_staged():
            c = t.cpu()
            dist.all_reduce(c, op=dist.ReduceOp.MAX, group=group)
            t.copy_(c)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return t


def barrier():
    if exchanging():
        dist.barrier()


def run_levels(top_parents, n_levels, classify_level, hints=None):
    """Level-synchronous traversal with a balanced slice per rank and one exchange per level.

    top_parents: (n, k) tensor, identical on every rank (the global top-level parent list).
    classify_level(level, my_parents) -> (m, k) tensor of that slice's surviving children.
    hints: optional list, one entry per level, updated in place: the largest per-rank survivor count seen
    at that level (+ slack, raised whenever a level overflows it); a caller that repeats the same traversal passes the same list again and
    each level's exchange becomes a single collective (allgather_rows).
    Returns (global leaf list, [global survivor count per level]).
    """
    rank, world = rank_world()
    parents, counts = top_parents, []
    for level in range(n_levels):
        begin, end = balanced_slice(parents.shape[0], rank, world)
        mine = classify_level(level, parents[begin:end])
        hint = hints[level] if hints is not None and level < len(hints) else None
        parents, per_rank = allgather_rows(mine, hint=hint, return_counts=True)
        if hints is not None:   # the same list on every rank: per_rank is global knowledge
            while len(hints) <= level:
                hints.append(None)
            biggest = max(per_rank)
            if hints[level] is None or biggest > hints[level]:
                hints[level] = biggest + biggest // 8 + 16
        counts.append(int(parents.shape[0]))
        if parents.shape[0] == 0:
            break
    return parents, counts


# REPLICATED LEVELS.  The first levels of a hierarchy are tiny (the bench's: 27 cells, then 27 parents of 4096 cells): cutting
# their parent lists into slices, all-gathering the survivors and slicing again costs far more than the work.  Every rank
# classifies such a level IN FULL instead -- no exchange --, and in the last replicated level keeps only the cells it OWNS
# (hu_*_level_owned: owner = mix(hash of the parent's row, the cell's linear index) mod world): the ranks' lists then partition
# that level's survivors evenly -- a cell's owner is as good as random -- whatever order each rank's atomics left its own
# lists in.  A level is replicated while (parents' capacity x cells) stays below this many samples.
REPLICATE_MAX_SAMPLES = int(os.environ.get("CODECAD_AMD_REPLICATE_SAMPLES", 4 << 20))


def replicated_levels(n_top, capacities, cells, limit=None):
    """How many leading levels of a hierarchy are replicated: `capacities[i]` = list capacity of level i's survivors,
    `cells[i]` = cells of a level-i parent.  The same on every rank (capacities are)."""
    limit = REPLICATE_MAX_SAMPLES if limit is None else limit
    k, parents = 0, int(n_top)
    for capacity, c in zip(capacities, cells):
        if parents * int(c) > limit:
            break
        k += 1
        parents = int(capacity)
    return k


def row_hash(words):
    """kernels.hpp row_hash over four uint32 words (a subdivision row; for a mass row: lo ^ hi of each double), in numpy."""
    import numpy
    m = 0xffffffff
    a, b, c, d = (int(numpy.uint32(w)) for w in words)
    h = (a * 0x9e3779b1) & m
    h = ((h ^ (h >> 15)) + b * 0x85ebca77) & m
    h = ((h ^ (h >> 13)) + c * 0xc2b2ae3d) & m
    h = ((h ^ (h >> 16)) + d * 0x27d4eb2f) & m
    return h ^ (h >> 15)


def owner_of(row_words, cell, world):
    """The rank that owns cell `cell` (linear index z + sz * (y + sy * x)) of the parent with these row words
    (kernels.hpp owned(): the parent's hash and the cell mixed, so that neighbouring cells scatter over the ranks)."""
    m = 0xffffffff
    v = (row_hash(row_words) + int(cell) * 0x9e3779b1) & m
    v = ((v ^ (v >> 15)) * 0x85ebca77) & m
    v ^= v >> 13
    return v % int(world)


class Overflow(RuntimeError):
    """A list of a LevelPipeline traversal was longer than its capacity (the result is incomplete)."""

    def __init__(self, needed):
        RuntimeError.__init__(self, "level capacities too small, needed per level: %s" % (needed,))
        self.needed = needed    # per level: the largest per-rank survivor count (and share) seen


def slice_rows_reference(gathered, rank, out, stats, sharers=None):
    """What `hu_slice_rows` computes on the device, in torch ops (CPU tensors: the gloo tests and nothing else).
    gathered: (world, piece_rows, k) integer tensor, row 0 of a piece = header (element 0 = rows that follow);
    out: (capacity + 1, k) <- [header | this rank's balanced share of the concatenated rows]; stats: (2,).
    sharers: among how many ranks the concatenation is shared out (default: as many as there are pieces;
    hu_slice_rows_of: ONE piece, all ranks)."""
    piece_rows = int(gathered.shape[1])
    world = int(sharers) if sharers is not None else int(gathered.shape[0])
    counts = [int(c) for c in gathered[:, 0, 0].tolist()]
    over = any(c > piece_rows - 1 for c in counts)
    counts = [min(c, piece_rows - 1) for c in counts]
    rows = torch.cat([gathered[r, 1:1 + c] for r, c in enumerate(counts)], dim=0)
    begin, end = balanced_slice(int(rows.shape[0]), rank, world)   # (world = the sharers)
    capacity = int(out.shape[0]) - 1
    if end - begin > capacity:
        over, end = True, begin + capacity
    out[0] = 0
    out[0, 0] = end - begin
    out[1:1 + end - begin] = rows[begin:end]
    stats[0] = int(rows.shape[0])
    stats[1] = 1 if over else 0


class LevelPipeline:
    """Level-synchronous traversal in which NOTHING waits for the host between levels.

    The reference reads a counter and a list back to the host after every block (subdivision.py:75-94); the
    drivers of this package read one counter per level; on eight GPUs even that -- plus the host-side slicing of
    the gathered lists -- is what a step consists of (a 512^3 level is ~10 us of kernel per rank).  Here every
    list is a fixed-capacity device buffer [header row | rows...] whose header holds its length:

      classify(level)  counts into the header of `send[level]`, appends to its rows   (hu_subdivision_level_indirect,
                                                                                        hu_mass_properties_level_indirect)
      all-gather       of the whole fixed-size piece, one collective, no sizes on the host     (when `exchanging()`)
      slice            the rank's balanced share of the concatenation -> `mine[level]`, again
                       [header | rows], on the device                                             (hu_slice_rows)
      next level       reads its parent count from that header; its launch is sized for the capacity.

    The host enqueues the whole traversal (and whatever consumes the leaf list) and looks at the headers once,
    afterwards (`check`): a list that outgrew its capacity raises Overflow with the sizes needed.  Capacities
    are per level: survivors PER RANK, hence also an upper bound of a rank's share.

    Rows are int32 words (k = 4: the 16-byte rows of subdivision; k = 8: the 32-byte rows of mass properties, four
    doubles seen as words), so the header's count is the uint32 the kernels read, whatever the rows hold.

    `classify(level, parents, n_parents, max_parents, out)`: parents = (max_parents, k) tensor of rows,
    n_parents = the 1-element view of the header holding their count, out = the (capacity + 1, k) buffer to
    count into / append to (header already zeroed).  For the HIP path see `subdivision_pipeline`, `mass_pipeline`."""

    def __init__(self, top_rows, capacities, classify, slice_rows=None, device=None, stream=None, replicate=0):
        """replicate = k: the first k levels are REPLICATED (see REPLICATE_MAX_SAMPLES): every rank classifies all of their
        parents itself, without an exchange; in the last of them the classification is called with own=(world, rank) and
        lists only the cells this rank owns, so that from there on every rank holds a share.  `classify` must then take
        that keyword (the HIP ones do: hu_*_level_owned).  Levels after the replicated ones are exchanged as described above
        -- there the all-gather re-balances the shares."""
        assert top_rows.dtype == torch.int32, "rows are int32 words (view 32-byte double rows as (n, 8) int32)"
        self.rank, self.world = rank_world()
        self.exchange = exchanging()
        self.replicate = min(max(int(replicate), 0), len(capacities)) if self.exchange else 0
        self.classify = classify
        self.slice_rows = slice_rows or slice_rows_reference
        self.capacities = [int(c) for c in capacities]
        device = device if device is not None else top_rows.device
        self.device, self.stream = device, stream
        k = int(top_rows.shape[1])
        begin, end = balanced_slice(int(top_rows.shape[0]), self.rank, self.world)   # the top list is host knowledge
        if self.replicate:
            begin, end = 0, int(top_rows.shape[0])
        self.top = torch.zeros((max(end - begin, 1) + 1, k), dtype=torch.int32, device=device)
        self.top[0, 0] = end - begin
        self.top[1:1 + end - begin] = top_rows[begin:end]
        self.top_max = max(end - begin, 1)
        self.send = [torch.zeros((c + 1, k), dtype=torch.int32, device=device) for c in self.capacities]
        self.gathered = self.mine = self.stats = None
        self._plan = self._result = None
        if self.exchange:
            self.gathered = [torch.zeros((self.world, c + 1, k), dtype=torch.int32, device=device) for c in self.capacities]
            self.mine = [torch.zeros((c + 1, k), dtype=torch.int32, device=device) for c in self.capacities]
            self.stats = torch.zeros((len(self.capacities), 2), dtype=torch.int32, device=device)

    def enqueue(self):
        """Enqueue the whole traversal on the current stream.  Returns the buffer [header | rows] holding this
        rank's share of the last level's survivors (no exchange: all of them)."""
        if self.exchange and self.stream is not None and torch.device(self.device).type == "cuda":
            # the kernels go to a raw hipStream_t, the collectives to torch's current stream: they must be the same one
            assert self.stream == torch.cuda.current_stream(self.device).cuda_stream, \
                "LevelPipeline.enqueue: make the pipeline's stream torch's current stream (torch.cuda.stream(...))"
        if self._plan is None:
            self._plan, self._result = self._make_plan()
        for call in self._plan:
            call()
        return self._result

    def _make_plan(self):
        """The traversal as a list of calls without arguments.  Every buffer of a pipeline is fixed, so what a call needs --
        views, pointers, converted scalars -- is worked out ONCE here (`classify.bind` / `slice_rows.bind` where the callee
        offers it: the HIP ones do); enqueue() then costs the host one C call or one torch op per item.  On eight GPUs a
        step of the bench is ~0.1 ms of device time: the ~10 us per level this saves are a tenth of it."""
        import functools

        def bound(fn, *args, **kwargs):
            bind = getattr(fn, "bind", None)
            return bind(*args, **kwargs) if bind is not None else functools.partial(fn, *args, **kwargs)

        plan = []
        self.lists = []          # per level: the buffer [header | rows] that holds this rank's survivors of that level
        parents, max_parents = self.top, self.top_max
        for level, capacity in enumerate(self.capacities):
            out = self.send[level]
            plan.append(out[:1].zero_)
            if level < self.replicate:
                # a replicated level: all of its parents, no exchange; the last one lists what this rank owns
                own = (self.world, self.rank) if level + 1 == self.replicate else None
                plan.append(bound(self.classify, level, parents[1:], parents[0, :1], max_parents, out, own=own))
                parents = out
                self.lists.append(parents)
                max_parents = capacity
                continue
            plan.append(bound(self.classify, level, parents[1:], parents[0, :1], max_parents, out))
            if not self.exchange or (capacity == 0 and level + 1 == len(self.capacities)):
                # (a last level that can list nothing -- the leaf level of mass properties -- has nothing to exchange)
                parents = out
            else:
                g = self.gathered[level]
                if g.device.type != "cpu" and _host_staged():     # several ranks on one GPU through gloo: rehearsal only
                    def staged(g=g, out=out):
                        host = [torch.empty(out.shape, dtype=out.dtype) for _ in range(self.world)]
                        dist.all_gather(host, out.cpu())
                        g.copy_(torch.stack(host))
                    plan.append(staged)
                elif g.device.type == "cpu":
                    plan.append(functools.partial(dist.all_gather, list(g.unbind(0)), out))
                else:
                    plan.append(functools.partial(dist.all_gather_into_tensor, g, out))
                plan.append(bound(self.slice_rows, g, self.rank, self.mine[level], self.stats[level]))
                parents = self.mine[level]
            self.lists.append(parents)
            max_parents = capacity
        return plan, parents

    def check(self):
        """Wait for the traversal and validate it.  Returns the global survivor count of every level; raises
        Overflow (with the capacities that would have sufficed) if a list was truncated."""
        if not self.send:
            self.needed = []
            return []
        own = [int(b[0, 0].item()) for b in self.send]       # synchronises
        if not self.exchange:
            totals, needed = own, own
        else:
            st = self.stats.cpu().tolist()
            totals = [int(row[0]) for row in st]
            if self.capacities[-1] == 0:        # (not exchanged, _make_plan: the count is this rank's own, and zero unless it overflowed)
                totals[-1] = own[-1]
            for level in range(self.replicate):     # replicated: every rank counted the whole level -- or, in the last one, its own cells
                totals[level] = own[level]
            if self.replicate:
                k = self.replicate - 1
                totals[k] = int(allreduce_sum(torch.tensor([own[k]], dtype=torch.int64, device=self.send[0].device)).item())
            # hu_slice_rows' truncation flag: set exactly when a gathered piece or a share was cut, i.e. when the counts below
            # say Overflow -- a flag without an overflow would be a protocol error
            truncated = [int(row[1]) != 0 for row in st]
            biggest = allreduce_max(torch.tensor(own, dtype=torch.int64, device=self.send[0].device)).tolist()
            needed = [max(int(b), -(-t // self.world)) for b, t in zip(biggest, totals)]
            for level in range(self.replicate):     # (a replicated level's buffer holds what the rank itself listed)
                needed[level] = int(biggest[level])
        self.needed = needed      # per level: the largest list any rank held (what the capacities must cover)
        if any(n > c for n, c in zip(needed, self.capacities)):
            raise Overflow(needed)
        if self.exchange:
            assert not any(truncated), "hu_slice_rows reported a truncated list although no level overflowed: %s" % (st,)
        return totals


def _hip_slice_rows(lib, check, stream):
    def bind(gathered, rank, out, stats, sharers=None):
        """-> the call without arguments (LevelPipeline._make_plan): pointers and sizes worked out once"""
        row_bytes = int(gathered.shape[2]) * gathered.element_size()
        if sharers is None:
            fn, name = lib.hu_slice_rows, "hu_slice_rows"
            args = (gathered.data_ptr(), int(gathered.shape[0]), int(gathered.shape[1]), row_bytes, rank, out.data_ptr(),
                    int(out.shape[0]) - 1, stats.data_ptr(), stream)
        else:
            assert int(gathered.shape[0]) == 1
            fn, name = lib.hu_slice_rows_of, "hu_slice_rows_of"
            args = (gathered.data_ptr(), int(gathered.shape[1]), row_bytes, rank, int(sharers), out.data_ptr(),
                    int(out.shape[0]) - 1, stats.data_ptr(), stream)
        keep = (gathered, out, stats)      # (the pointers stay valid as long as the call does)

        def run():
            check(fn(*args), name)
            return keep
        return run

    def slice_rows(gathered, rank, out, stats, sharers=None):
        bind(gathered, rank, out, stats, sharers)()
    slice_rows.bind = bind
    return slice_rows


def subdivision_pipeline(tape, levels, resolution, origin, dimension, capacities, device, stream, top_rows=None):
    """A LevelPipeline over the HIP kernels: `levels` as calculate_block_sizes returns them (the leaf level,
    the last one, is left to the consumer like in subdivision_device), `origin` the box corner.  The launches
    go to `stream` (a raw hipStream_t; it must be the current torch stream when the levels are exchanged, the
    collectives follow that: LevelPipeline.enqueue asserts it)."""
    import ctypes
    import math
    import numpy
    from .hip_util import manager as hip_manager, check

    lib = hip_manager.lib
    o = (ctypes.c_double * 3)(*[float(v) for v in origin])

    def bind(level, parents, n_parents, max_parents, out, own=None):
        """-> the launch without arguments (LevelPipeline._make_plan): pointers, sizes and converted scalars worked out once;
        own = (world, rank): a replicated level of which this rank lists the cells it owns (hu_subdivision_level_owned)"""
        int_step, dims = levels[level]
        d = (ctypes.c_uint32 * 3)(int(dims[0]), int(dims[1]), int(dims[2]))
        box_step = int_step * resolution
        thr = box_step * math.sqrt(dimension) / 2
        args = (tape.device_ptr, parents.data_ptr(), n_parents.data_ptr(), int(max_parents), int(int_step), d, dimension,
                float(resolution), o, numpy.float32(box_step), numpy.float32(thr), out.data_ptr(), out[1:].data_ptr(),
                int(out.shape[0]) - 1)
        keep = (parents, n_parents, out, d)
        if own is None:
            fn, name, args = lib.hu_subdivision_level_indirect, "hu_subdivision_level_indirect", args + (stream,)
        else:
            fn, name, args = lib.hu_subdivision_level_owned, "hu_subdivision_level_owned", args + (int(own[0]), int(own[1]), stream)

        def run():
            check(fn(*args), name)
            return keep
        return run

    def classify(level, parents, n_parents, max_parents, out, own=None):
        bind(level, parents, n_parents, max_parents, out, own)()
    classify.bind = bind

    if top_rows is None:
        top_rows = torch.zeros((1, 4), dtype=torch.int32, device=device)
    # the small leading levels are classified by every rank in full, the last of them with ownership (REPLICATE_MAX_SAMPLES)
    cells = [int(d[0]) * int(d[1]) * int(d[2]) for _, d in levels]
    return LevelPipeline(top_rows, capacities, classify, _hip_slice_rows(lib, check, stream), device, stream,
                         replicate=replicated_levels(int(top_rows.shape[0]), capacities, cells))


class MassPipeline:
    """mass_properties over the ranks without a host round trip inside the traversal: a LevelPipeline whose rows are
    the 32-byte `double[4]` rows of hu_mass_properties_level_indirect (seen as eight int32 words), every level also
    reducing its inside cells' ten integrals on the device (hu_mass_integrals_indirect) into its own small buffer.
    `enqueue()` puts the whole integration on the stream; `finish()` waits once, validates the lists (Overflow ->
    rebuild with larger ones, every rank alike) and returns this rank's ten partial integrals, a (10,) float64 tensor
    on the device, for ONE all-reduce.  `levels` = [(cell size, dims)], the last one being the leaf level."""

    def __init__(self, tape, levels, box_a, capacities, device, stream):
        import ctypes
        import math
        import numpy
        from .hip_util import manager as hip_manager, check
        from .mass_properties import integral_rows

        lib = hip_manager.lib
        self.levels, self.device = levels, device
        self.timing = None
        capacities = list(capacities) + [0]     # the leaf level lists nothing
        top = torch.zeros((1, 4), dtype=torch.float64, device=device)
        top[0, :3] = torch.tensor([float(v) for v in box_a], dtype=torch.float64)
        # what a level's launches read and write besides the lists: index sums per parent, the integrals' rows
        self.sums, self.pieces = [], []
        max_parents = 1
        for capacity in capacities:
            self.sums.append(torch.zeros((max_parents, 10), dtype=torch.int32, device=device))
            self.pieces.append(torch.zeros((integral_rows(max_parents), 10), dtype=torch.float64, device=device))
            max_parents = max(capacity, 1)

        def classify(level, parents, n_parents, max_parents, out, own=None):
            s, dims = levels[level]
            leaf = level + 1 == len(levels)
            thr = 0.0 if leaf else s * math.sqrt(3) / 2
            d = (ctypes.c_uint32 * 3)(int(dims[0]), int(dims[1]), int(dims[2]))
            sums, pieces = self.sums[level], self.pieces[level]
            assert int(sums.shape[0]) >= int(max_parents)
            sums.zero_()
            events = (self.timing or {}).get(level)       # bench.py: HIP events around one level's classification launch
            if events:
                check(lib.hu_event_record(events[0], stream), "hu_event_record")
            if own is None:
                check(lib.hu_mass_properties_level_indirect(tape.device_ptr, parents.data_ptr(), n_parents.data_ptr(), int(max_parents),
                                                            float(s), d, numpy.float32(s), numpy.float32(thr), sums.data_ptr(),
                                                            out.data_ptr(), out[1:].data_ptr(), int(out.shape[0]) - 1, stream),
                      "hu_mass_properties_level_indirect")
            else:   # a replicated level: this rank's sums and list hold the cells it owns
                check(lib.hu_mass_properties_level_owned(tape.device_ptr, parents.data_ptr(), n_parents.data_ptr(), int(max_parents),
                                                         float(s), d, numpy.float32(s), numpy.float32(thr), sums.data_ptr(),
                                                         out.data_ptr(), out[1:].data_ptr(), int(out.shape[0]) - 1, int(own[0]), int(own[1]), stream),
                      "hu_mass_properties_level_owned")
            if events:
                check(lib.hu_event_record(events[1], stream), "hu_event_record")
            check(lib.hu_mass_integrals_indirect(parents.data_ptr(), sums.data_ptr(), n_parents.data_ptr(), int(max_parents), float(s),
                                                 pieces.data_ptr(), int(pieces.shape[0]), stream), "hu_mass_integrals_indirect")

        # (the leaf level lists nothing: it is never worth replicating -- its parents are the work -- and needs no exchange)
        cells = [int(d[0]) * int(d[1]) * int(d[2]) for _, d in levels]
        self.pipe = LevelPipeline(top.view(torch.int32).reshape(1, 8), capacities, classify, _hip_slice_rows(lib, check, stream),
                                  device, stream, replicate=min(replicated_levels(1, capacities, cells), len(levels) - 1))

    def enqueue(self):
        self.pipe.enqueue()

    def finish(self):
        """-> (this rank's (10,) partial integrals, global ambiguous-cell count per level)"""
        try:
            totals = self.pipe.check()
        except Overflow as e:
            raise Overflow(e.needed[:-1])       # (the leaf level's "list" has no capacity to speak of)
        partial = torch.zeros(10, dtype=torch.float64, device=self.device)
        for pieces in self.pieces:
            partial = partial + pieces.sum(dim=0)     # a fixed reduction over <= 64 rows per level
        return partial, totals[:-1]


def integrate_levels(top_parents, n_levels, level_fn):
    """Level-synchronous hierarchical integration (mass properties) over all ranks.

    top_parents: (n, k) tensor, identical on every rank.  level_fn(level, my_parents) ->
    (children (m, k) tensor, partial (10,) float64 tensor): the rank's slice classified, its ambiguous
    cells listed, and the ten integrals of its inside cells.  Children are exchanged with one
    variable-length all-gather per level; the integrals are summed locally over the levels and
    all-reduced ONCE at the end (80 bytes).  Returns the (10,) totals, identical on every rank.
    """
    rank, world = rank_world()
    parents = top_parents
    totals = None
    for level in range(n_levels):
        begin, end = balanced_slice(parents.shape[0], rank, world)
        children, partial = level_fn(level, parents[begin:end])
        totals = partial.clone() if totals is None else totals + partial
        if level + 1 == n_levels:
            break
        parents = allgather_rows(children)
        if parents.shape[0] == 0:
            break
    return allreduce_sum(totals)


def mass_properties(shape, resolution, grid_size=None):
    """`codecad_amd.mass_properties` sharded over the ranks of the process group (one GPU each): every level's parent
    list is cut into balanced slices, `hu_mass_properties_level_indirect` + `hu_mass_integrals_indirect` run on the
    slice, the ambiguous cells go through ONE fixed-size all-gather per level and `hu_slice_rows`, the ten integrals are
    all-reduced once -- the whole integration enqueued without a host round trip (MassPipeline).  Same volume, centroid
    and inertia as the single-GPU driver up to the order of the fp64 sums (~1e-15 relative)."""
    from . import nodes, subdivision
    from .mass_properties import finish, _KEYS   # (the package attribute of that name is the function)
    from .hip_util import manager as hip_manager

    if grid_size is None:
        grid_size = 64
    assert shape.dimension() == 3 and resolution > 0 and grid_size > 1 and grid_size ** 5 <= 2 ** 32
    device = torch.device("cuda", local_device())
    hip_manager.use_device(device.index)
    stream = torch.cuda.current_stream(device).cuda_stream
    tape = nodes.make_program_buffer(shape)
    box = shape.bounding_box()
    levels = [(resolution * cell, tuple(int(v) for v in dims)) for cell, dims in
              subdivision.calculate_block_sizes(box, 3, resolution, grid_size, overlap=False)]
    cells = [d[0] * d[1] * d[2] for _, d in levels]
    capacities = subdivision.first_capacities(cells[:-1], row_bytes=32 + 40)
    while True:
        pipe = MassPipeline(tape, levels, (box.a.x, box.a.y, box.a.z), capacities, device, stream)
        pipe.enqueue()
        try:
            partial, _ = pipe.finish()
            break
        except Overflow as e:     # every rank sees the same sizes: all rebuild together
            capacities = [subdivision.checked_capacity(max(int(v * 1.125) + 16, c)) for v, c in zip(e.needed, capacities)]
    totals = allreduce_sum(partial)
    return finish(dict(zip(_KEYS, totals.tolist())))


def subdivision(shape, resolution, overlap_edge_samples=True, grid_size=None):
    """`codecad_amd.subdivision.subdivision_device` sharded over the ranks of the process group: every
    level's parent list is cut into balanced slices, `hu_subdivision_level_indirect` classifies and compacts the
    slice, the survivors go through one fixed-size all-gather and `hu_slice_rows` (LevelPipeline).  Returns (leaves, info): `leaves` an (n, 4) int32 device tensor of
    integer leaf corners, identical (as a set; ordered by rank slice) on every rank, and `info` with
    `dims`, `int_step`, `step`, `resolution`, `origin`, `level_counts` like LeafBlocks, plus `share`: this rank's
    balanced share of the leaves (what a sharded consumer evaluates, e.g. hu_grid_eval_blocks: bench.py step C)."""
    from . import nodes
    from . import subdivision as sub
    from .hip_util import manager as hip_manager

    if grid_size is None:
        grid_size = 128
    assert resolution > 0 and 1 < grid_size <= 256
    device = torch.device("cuda", local_device())
    hip_manager.use_device(device.index)
    stream = torch.cuda.current_stream(device).cuda_stream
    tape = nodes.make_program_buffer(shape)
    dimension = shape.dimension()
    box = shape.bounding_box().expanded_additive(resolution / 2)
    if dimension == 2:
        box = box.flattened()
    levels = sub.calculate_block_sizes(box, dimension, resolution, grid_size, overlap_edge_samples)
    # the whole traversal is enqueued without a host round trip (LevelPipeline); the list capacities start at the
    # single-GPU driver's first guess and grow to what an overflowing traversal reports
    cells = [int(d[0]) * int(d[1]) * int(d[2]) for _, d in levels]
    capacities = sub.first_capacities(cells[:-1])
    while True:
        pipe = subdivision_pipeline(tape, levels, resolution, (box.a.x, box.a.y, box.a.z), dimension, capacities, device, stream)
        mine = pipe.enqueue()
        try:
            counts = pipe.check()
            break
        except Overflow as e:
            capacities = [sub.checked_capacity(max(int(v * 1.125) + 16, c)) for v, c in zip(e.needed, capacities)]
    if len(levels) > 1:
        # (a copy: the pipeline's buffers are sized for their capacities -- a first guess may be tens of MB -- and go with it)
        share = mine[1:1 + int(mine[0, 0])].clone()
        leaves = allgather_rows(share)
    else:
        # the whole shape is one leaf block: it is rank 0's, the other ranks' shares are empty
        leaves = torch.zeros((1, 4), dtype=torch.int32, device=device)
        share = leaves if pipe.rank == 0 else leaves[:0]
    leaf_int_step, leaf_dims = levels[-1]
    info = {"tape": tape, "dims": leaf_dims, "int_step": leaf_int_step, "step": leaf_int_step * resolution,
            "resolution": resolution, "origin": box.a, "level_counts": counts, "share": share}
    return leaves, info
