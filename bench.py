#!/usr/bin/env python3
"""bench.py -- SDF Mvoxels/s (grid_eval + subdivision) on the 512^3 Menger sponge, 1..N GPUs.

One "step" is one pass of the hot path over ONE object (no RNG: the only input is the CSG tree):
  A. dense grid_eval (float4) of sponge(4) on the 512^3 cell-centred grid          [k_grid_eval<0>]
  B. adaptive subdivision of sponge(4) at resolution 1/512, grid 16, overlapping
     leaf samples: levels [(240,3^3),(15,16^3),(1,16^3)], survivors compacted by the
     wavefront ballot scan                                                      [k_classify<0,1>]
  C. grid_eval of ALL surviving 16^3 leaf blocks in one launch (float, PyMCubes layout,
     what the reference's mesh pipeline does block by block)                [k_grid_eval_blocks<1>]
`value` counts SDF samples actually evaluated (A + B + C) per second over all ranks; nothing is cached
between steps and the output buffers are rewritten every step.  B is latency (tiny kernels, all-gathers),
so it runs on a second HIP stream concurrently with A; C waits for both.  Nothing in a step waits for
the host: list lengths stay on the device (codecad_amd.dist.LevelPipeline) and are validated after the
timed region.

N > 1 (torchrun, one rank per GPU, RCCL): STRONG scaling -- the same one object.  The 512^3 grid is cut
into x-slabs (rank r evaluates dist.x_slab(512, r, N) through hu_grid_eval_slab, no collective); the ONE
hierarchy is walked level by level, every rank classifying its balanced slice of the level's parents, with
one fixed-size RCCL all-gather of the survivors per level; the leaf blocks are balanced over the ranks.
`--weak` keeps round 1's mode (N objects, one per GPU) as a secondary measurement.
`--config c5` is BASELINE config 5: sponge(5) at 1/2048 (2048^3 effective), grid 16: subdivision + grid_eval
of its 1.34 M leaf blocks (5.5 G samples) -- the multi-GPU workload that is not launch-latency-sized.

The JSON line carries `roofline` for the dominant kernel (timed with HIP events on its own stream),
`roofline_hbm` (the HBM-bound regime: the same dense kernel on tapes whose arithmetic fits under the store
stream) and, at N = 1, `cpu_baseline` (the CPU oracle on a bounded sample).
"""
import argparse
import ctypes
import hashlib
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N = 512
SUBDIV_GRID = 16
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_PEAK_TFLOPS = 157.3   # vector FP32, FMA-counted
SIMDS = 1024               # 256 CUs x 4 SIMDs
CLOCK_HZ = 2.4e9           # peak engine clock
VALU_ISSUE_PEAK = SIMDS * CLOCK_HZ / 4 / 1e9   # a wave64 VALU instruction occupies its SIMD16 for 4 cycles: G wavefront-instructions / s

CONFIGS = {
    # name: (sponge depth, effective grid edge, dense leg?)
    "c3": (4, 512, True),
    "c5": (5, 2048, False),
    # BASELINE config 4: mass_properties of the planetary assembly at resolution 0.25, grid 64 (run_c4 below)
    "c4": (None, None, False),
}
C4_RESOLUTION, C4_GRID = 0.25, 64


def csrc_hash():
    """Identifies the device code a profile was taken on (profiles/*summary.json carry the same field)."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "codecad_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hpp", ".hip")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c3")
    ap.add_argument("--weak", action="store_true", help="N objects (one per GPU) instead of one object over N GPUs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hbm-leg", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip the comparison of the timed steps' output with the CPU oracle")
    ap.add_argument("--no-graph", action="store_true", help="skip the hipGraph leg (steps captured once and replayed)")
    ap.add_argument("--graph", action="store_true", help="run the hipGraph leg on more than one rank too (default: one rank only)")
    ap.add_argument("--pipelines", type=int, default=None,
                    help="traversals in flight, each with its lists and its stream (default: 2)")
    ap.add_argument("--n", type=int, default=None, help="grid edge (default: the config's)")
    ap.add_argument("--evaluator", choices=["auto", "specialised", "interpreter"], default="auto",
                    help="auto = per-tape hipRTC specialisation when it builds, else the tape interpreter")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Called directly with --gpus N (the driver launches torch.distributed.run itself): start the ranks as a
        # child job -- nothing has touched the GPU yet -- and pass its exit code on.
        import socket
        import subprocess
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    # ONE line on stdout, whatever the libraries underneath print there (RCCL announces its version on stdout when a
    # process group comes up): everything else this process writes to file descriptor 1 goes to stderr
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import numpy as np
    import torch
    import codecad_amd as cc
    from codecad_amd import hip_util, dist
    from codecad_amd.hip_util import check

    rank, world = dist.init()
    local = dist.local_device()
    torch.cuda.set_device(local)
    hip_util.manager.use_device(local)
    m = hip_util.manager
    lib = m.lib
    dev = torch.device("cuda", local)
    if args.config == "c4":
        run_c4(args, real_stdout, rank, world, dev)
        return
    depth, n_default, dense_leg = CONFIGS[args.config]
    n = args.n or n_default
    weak = args.weak and world > 1
    fptr = ctypes.POINTER(ctypes.c_float)

    shape = cc.examples.sponge(depth)
    host_tape = cc.nodes.make_program(shape)
    interp_tape = hip_util.Tape(host_tape, policy="0")   # the same program, always interpreted (reported beside the headline)
    tape = interp_tape
    evaluator = "interpreter"
    if args.evaluator != "interpreter":
        try:
            tape = hip_util.Tape(host_tape, policy="0").specialize(hip_util.SPEC_DENSE | hip_util.SPEC_BLOCKS | hip_util.SPEC_CLASSIFY)   # (no renderers in a step) ~0.5 s of hipRTC (ms from the disk cache), outside the timed region
            evaluator = "specialised"
        except RuntimeError as e:
            if args.evaluator == "specialised":
                raise
            print("bench: hipRTC specialisation unavailable, using the interpreter: %s" % str(e)[:300], file=sys.stderr)
    # A and C are enqueued on torch's current stream.  The subdivision (B) is a chain of tiny kernels and
    # (N > 1) all-gathers: latency, not work.  It runs on a second, high-priority stream -- made torch's current
    # stream while it is enqueued, so that the collectives order against it -- and hides behind A; C waits for both.
    # (its own stream, not the legacy default one: a step is also captured into a hipGraph, which the default stream cannot do)
    main_stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(main_stream)
    stream = main_stream.cuda_stream
    # One stream per traversal in flight.  A traversal is a latency chain (per level: classify, all-gather, slice); on
    # several GPUs it is longer than a rank's share of the dense grid and the leaf blocks together, so two of them are
    # kept in flight, each on its own stream with its own lists (the collectives of one process group still run in the
    # order they were enqueued, the same on every rank): a step then costs max(A + C, B / 2), not max(A + C, B) as with
    # round 2's single side stream.  More in flight was measured on one GPU with forced collectives (--pipelines 4):
    # 0.92 ms per step against 0.70 -- four high-priority traversals get in the way of the dense kernel.
    n_pipes = args.pipelines or 2
    assert n_pipes >= 2
    side_streams = [torch.cuda.Stream(device=dev, priority=-1) for _ in range(n_pipes)]

    # ---- A: dense grid (this rank's x-slab of the ONE grid; --weak: a whole grid per rank) --------------
    step_f = np.float32(1.0 / n)
    corner = np.array([-0.5 + 0.5 / n] * 3 + [0.0], dtype=np.float32)
    dims = (ctypes.c_uint32 * 3)(n, n, n)
    x0, x1 = (0, n) if (weak or world == 1) else dist.x_slab(n, rank, world)
    dense_voxels = (x1 - x0) * n * n if dense_leg else 0
    dense_out = torch.empty((max(x1 - x0, 1), n, n, 4), dtype=torch.float32, device=dev) if dense_leg else None

    # ---- B: the hierarchy ---------------------------------------------------------------------------------
    resolution = 1.0 / n
    box = shape.bounding_box().expanded_additive(resolution / 2)
    levels = cc.subdivision.calculate_block_sizes(box, 3, resolution, SUBDIV_GRID, True)
    origin = (ctypes.c_double * 3)(box.a.x, box.a.y, box.a.z)
    cells = [int(d[0]) * int(d[1]) * int(d[2]) for _, d in levels]
    n_objects = world if weak else 1
    top = torch.zeros((n_objects, 4), dtype=torch.int32, device=dev)
    top[:, 3] = torch.arange(n_objects, dtype=torch.int32, device=dev)   # the object id rides in the 4th component
    leaf_int_step, leaf_dims = levels[-1]
    leaf_cells = cells[-1]
    ld = (ctypes.c_uint32 * 3)(int(leaf_dims[0]), int(leaf_dims[1]), int(leaf_dims[2]))

    def build_pipeline(capacities, i):
        with torch.cuda.stream(side_streams[i]):
            p = dist.subdivision_pipeline(tape, levels, resolution, (box.a.x, box.a.y, box.a.z), 3, capacities, dev,
                                          side_streams[i].cuda_stream, top)
        side_streams[i].synchronize()
        return p

    # first capacities: every cell of a level while that is small, else a surface estimate; a traversal that
    # overflows says what it needed (dist.Overflow) and the pipeline is rebuilt -- during warm-up only
    capacities = cc.subdivision.first_capacities(cells[:-1], n_top=n_objects)
    # P pipelines, taken in turn: the traversal of step k + 1 (latency-bound at 8 GPUs: two levels of classify +
    # all-gather + slice) runs on its stream while step k's leaf blocks are still being evaluated from another
    # pipeline's list; it only waits for the leaf-block launch that last read ITS list (step k + 1 - P).
    pipes = [build_pipeline(capacities, i) for i in range(n_pipes)]
    list_free = [torch.cuda.Event() for _ in range(n_pipes)]   # recorded after the leaf-block launch that read pipes[i]'s list
    leaf_out = [None]

    # one set of events per step: nothing in a step waits for the host, the elapsed times are read after the timed region
    def new_events():
        evs = [ctypes.c_void_p() for _ in range(6)]
        for ev in evs:
            check(lib.hu_event_create(ctypes.byref(ev)), "event")
        return evs
    warm_events = new_events()
    # the steps whose legs are timed with events: all of a short run, eight spread evenly over a long one
    timed_steps = sorted(set(range(args.steps)) if args.steps <= 8 else {round(j * (args.steps - 1) / 7) for j in range(8)})
    step_events = {k: new_events() for k in timed_steps}

    # (the launches' arguments are fixed: pointers and converted scalars are worked out once, a launch costs the host one C call)
    a_args = (tape.device_ptr, corner.ctypes.data_as(fptr), step_f, dims, x0, x1 - x0, 0, dense_out.data_ptr() if dense_leg else None, stream)

    def launch_a():
        check(lib.hu_grid_eval_slab(*a_args), "hu_grid_eval_slab")

    c_args = {}
    leaf_step_f = np.float32(leaf_int_step * resolution)

    def launch_c(mine):
        # the launch is sized for the list's capacity, the length is read on the device
        args = c_args.get(id(mine))
        if args is None or args[0] is not leaf_out[0] or args[1] is not mine:
            cap = int(mine.shape[0]) - 1
            if leaf_out[0] is None or leaf_out[0].shape[0] < cap:
                leaf_out[0] = torch.empty((cap, leaf_cells), dtype=torch.float32, device=dev)
                c_args.clear()
            args = (leaf_out[0], mine, (tape.device_ptr, mine[1:].data_ptr(), mine.data_ptr(), cap, resolution, origin, leaf_step_f, ld, 1,
                                        leaf_out[0].data_ptr(), stream))
            c_args[id(mine)] = args
        check(lib.hu_grid_eval_blocks_indirect(*args[2]), "hu_grid_eval_blocks_indirect")

    joined = [torch.cuda.Event() for _ in range(n_pipes)]

    def one_step(evs, k):
        """`evs`: the step's six timing events, or None (most of a long run's steps: six hipEventRecord calls are 27 us of
        host time, more than all the launches of the step; the breakdown is taken from a sample of the steps)."""
        i = k % n_pipes
        pipe, side_stream = pipes[i], side_streams[i]
        if evs is not None:
            ev0, ev1, ev2, evb0, evb1, evc0 = evs
            side = side_stream.cuda_stream
        # A
        if dense_leg:
            if evs is not None:
                check(lib.hu_event_record(ev0, stream), "record")
            launch_a()
            if evs is not None:
                check(lib.hu_event_record(ev1, stream), "record")
        # B, concurrently with A, on the pipeline's stream
        side_stream.wait_event(list_free[i])       # (never recorded yet: no wait)
        if evs is not None:
            check(lib.hu_event_record(evb0, side), "record")
        with torch.cuda.stream(side_stream):       # (the pipeline's torch ops -- header resets, collectives -- follow the current stream)
            mine = pipe.enqueue()                  # [header | this rank's share of the leaf blocks], all on the device
        if evs is not None:
            check(lib.hu_event_record(evb1, side), "record")
        joined[i].record(side_stream)
        main_stream.wait_event(joined[i])
        # C, after A and B
        if evs is not None:
            check(lib.hu_event_record(evc0, stream), "record")
        launch_c(mine)
        if evs is not None:
            check(lib.hu_event_record(ev2, stream), "record")
        list_free[i].record(main_stream)              # the traversal P steps on may overwrite this list once C has read it
        return mine

    def capture_steps(n_steps):
        """n_steps whole steps -- A, the traversal with its collectives, C -- captured into ONE hipGraph, with the
        dependencies of the direct form: the traversals run on the side stream beside A and the previous step's C,
        pipelines taken in turn (traversal k waits for C of step k - 2, which read its lists); each C joins its
        traversal.  Replaying it costs the host one call per n_steps steps."""
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=main_stream):
            fork = torch.cuda.Event()
            fork.record(main_stream)
            for s_ in side_streams:
                s_.wait_event(fork)
            c_done = []
            for k in range(n_steps):
                side_stream = side_streams[k % n_pipes]
                with torch.cuda.stream(side_stream):
                    if k >= n_pipes:
                        side_stream.wait_event(c_done[k - n_pipes])
                    mine = pipes[k % n_pipes].enqueue()
                    joined = torch.cuda.Event()
                    joined.record(side_stream)
                if dense_leg:
                    launch_a()
                main_stream.wait_event(joined)
                launch_c(mine)
                done = torch.cuda.Event()
                done.record(main_stream)
                c_done.append(done)
            for s_ in side_streams:
                main_stream.wait_stream(s_)
        return g

    def elapsed(a, b):
        ms = ctypes.c_float()
        check(lib.hu_event_elapsed_ms(a, b, ctypes.byref(ms)), "elapsed")
        return ms.value

    def barrier():
        main_stream.synchronize()
        for s_ in side_streams:
            s_.synchronize()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    torch.cuda.synchronize()                       # setup (default-stream fills and copies) is complete before any stream runs
    # warm-up: the first traversals also settle the list capacities at what the lists need + 12 % (a launch is
    # sized for the capacity, and workgroups past the list's end cost their dispatch: 80 000 spare leaf blocks are 0.2 ms)
    # (tightening may be needed twice: with tight lists the small levels become REPLICATED levels -- every rank classifies them
    # in full and keeps the cells it owns, dist.py -- and a rank's share of the leaf level is then 1 / N of what it was)
    totals, settled, done = None, 0, 0
    for i in range(n_pipes * max(args.warmup, 1) + 8 * n_pipes):
        pipe = pipes[i % n_pipes]
        mine = one_step(warm_events, i)
        try:
            totals = pipe.check()
            tight = [int(v * 1.125) + 16 for v in pipe.needed]
        except dist.Overflow as e:
            totals, tight = None, [int(v * 1.125) + 16 for v in e.needed]
        if totals is None or (settled < 3 and any(c > t + t // 8 + 64 for c, t in zip(pipe.capacities, tight))):
            barrier()
            pipes[:] = [build_pipeline(tight, j) for j in range(n_pipes)]
            leaf_out[0] = None
            settled += 1 if totals is not None else 0
            totals = None
            done = 0
            continue
        done += 1
        if done >= max(args.warmup, n_pipes):   # (every pipeline has run with the settled capacities)
            break
    assert totals is not None, "list capacities did not settle"
    my_leaves = int(mine[0, 0].item())

    # the interpreter on the same dense launch, timed the same way (reported, not part of `value`)
    interp_ms = []
    if dense_leg:
        ev0, ev1 = warm_events[0], warm_events[1]
        for i in range(3):
            check(lib.hu_event_record(ev0, stream), "record")
            check(lib.hu_grid_eval_slab(interp_tape.device_ptr, corner.ctypes.data_as(fptr), step_f, dims, x0, x1 - x0, 0,
                                        dense_out.data_ptr(), stream), "hu_grid_eval_slab")
            check(lib.hu_event_record(ev1, stream), "record")
            check(lib.hu_event_synchronize(ev1), "sync")
            if i:
                interp_ms.append(elapsed(ev0, ev1))
    barrier()
    t0 = time.perf_counter()
    last_mine = None
    for k in range(args.steps):
        last_mine = one_step(step_events.get(k), k)
    enqueue_s = time.perf_counter() - t0            # host time to enqueue the steps (they run behind it)
    barrier()
    wall = time.perf_counter() - t0
    if world > 1:
        wall = float(dist.allreduce_max(torch.tensor([wall], dtype=torch.float64, device=dev)).item())
    dense_ms = [elapsed(e[0], e[1]) for e in step_events.values()] if dense_leg else []
    b_ms = [elapsed(e[3], e[4]) for e in step_events.values()]   # B: the whole traversal on its stream (kernels + all-gathers)
    c_ms = [elapsed(e[5], e[2]) for e in step_events.values()]   # C: every sample of every leaf block of this rank
    # the timed traversals were not looked at while they ran: validate now (identical work every step)
    for used in pipes[:min(args.steps, n_pipes)]:
        assert used.check() == totals, "the timed steps did not reproduce the warm-up traversal"
    # what the timed steps WROTE, against the CPU oracle, bit for bit (outside the timing; a checker, nothing measured)
    verified = None
    if not args.no_verify:
        verified = verify_outputs(np, torch, host_tape, dense_out if dense_leg else None, corner, step_f, x0, n, last_mine,
                                  leaf_out[0], resolution, (box.a.x, box.a.y, box.a.z), leaf_step_f, leaf_dims)
        if world > 1:
            bad = torch.tensor([0.0 if verified["ok"] else 1.0], dtype=torch.float64, device=dev)
            verified["ranks_ok"] = bool(float(dist.allreduce_max(bad).item()) == 0.0)
            verified["ranks"] = world

    # ---- the same steps captured into a hipGraph and replayed (reported beside the headline, never instead of it):
    # up to eight steps per graph + one graph for the remainder, EXACTLY --steps steps in all; what it shows is the
    # host's share of a step -- ~25 ctypes / torch calls enqueued directly, one graph launch per eight steps replayed
    graph_leg = None
    if not args.no_graph and (world == 1 or args.graph):
        try:
            barrier()
            unit = min(8, args.steps)
            g_unit = capture_steps(unit)
            g_rest = capture_steps(args.steps % unit) if args.steps % unit else None
            # warm: the first launch of a graph uploads it, and the kernel trace (tools/r04/graph_trace.py) shows the first ~16
            # steps after the capture running ~20 % slower whatever launches them -- the capture and instantiation keep the host
            # busy for tens of ms while the device idles and clocks down -- so the graph is replayed three times before it is timed
            for _ in range(3):
                g_unit.replay()
            if g_rest is not None:
                g_rest.replay()
            barrier()
            tg = time.perf_counter()
            for _ in range(args.steps // unit):
                g_unit.replay()
            if g_rest is not None:
                g_rest.replay()
            g_enqueue = time.perf_counter() - tg
            barrier()
            g_wall = time.perf_counter() - tg
            if world > 1:
                g_wall = float(dist.allreduce_max(torch.tensor([g_wall], dtype=torch.float64, device=dev)).item())
            for used in pipes:
                assert used.check() == totals, "the replayed steps did not reproduce the warm-up traversal"
            graph_leg = {"captured": True, "steps_per_graph": unit, "steps": args.steps,
                         "ms_per_step": round(g_wall / args.steps * 1e3, 4),
                         "host_enqueue_ms_per_step": round(g_enqueue / args.steps * 1e3, 4),
                         "collectives_in_graph": bool(dist.exchanging()) and pipes[0].replicate < len(pipes[0].capacities)}
        except Exception as e:   # (a runtime that cannot capture a collective: said, not hidden)
            graph_leg = {"captured": False, "error": ("%s: %s" % (type(e).__name__, e))[:400]}

    # samples of one step over ALL ranks: dense voxels + every classified cell + every leaf sample
    parents_per_level = [n_objects] + totals[:-1]
    subdivision_samples = sum(p * c for p, c in zip(parents_per_level, cells[:-1]))
    leaf_samples = totals[-1] * leaf_cells
    job_dense = (n ** 3 * n_objects) if dense_leg else 0
    job_samples = job_dense + subdivision_samples + leaf_samples
    value = job_samples * args.steps / wall / 1e6

    def avg(v):
        return sum(v) / len(v) if v else 0.0

    def allmax(x):
        return float(dist.allreduce_max(torch.tensor([x], dtype=torch.float64, device=dev)).item()) if world > 1 else x

    b_avg, c_avg, dense_avg_ms = allmax(avg(b_ms)), allmax(avg(c_ms)), avg(dense_ms)
    hbm_leg = None
    if rank == 0 and world == 1 and not args.no_hbm_leg:
        hbm_leg = hbm_regime(lib, check, hip_util, cc, torch, np, dev, stream, n if dense_leg else 512, evaluator)

    if rank == 0:
        flop = tape_flop(host_tape)
        adaptive_ms = b_avg + c_avg
        if dense_leg:
            kernel = "k_grid_eval<%s, 0, 2>" % ("JitEval" if evaluator == "specialised" else "InterpEval<false>")
            k_ms, k_voxels, k_bytes = dense_avg_ms, dense_voxels, dense_voxels * 16.0
            k_note = "16 B stored per voxel (float4), reads ~ 0"
        else:
            kernel = "k_grid_eval_blocks<%s, 1, 2>" % ("JitEval" if evaluator == "specialised" else "InterpEval<true>")
            k_ms, k_voxels, k_bytes = avg(c_ms), my_leaves * leaf_cells, my_leaves * (leaf_cells * 4.0 + 16.0)
            k_note = "4 B stored per sample (float) + 16 B read per leaf block"
        achieved = k_bytes / (k_ms * 1e-3) / 1e9
        prof = profile_summary(kernel, k_voxels, args.config) if world == 1 and (args.n is None or args.n == n_default) else None
        roofline = {"bound": "hbm", "kernel": kernel, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None, "kernel_ms": round(k_ms, 4),
                    "algorithmic_bytes": k_bytes, "algorithmic_bytes_note": k_note,
                    "voxels_per_s": round(k_voxels / (k_ms * 1e-3), 0), "rank": 0}
        if prof is not None:
            # Both roofs of this launch.  `bound` stays the memory roof the north star names (algorithmic bytes / this run's
            # kernel time / 8 TB/s; `traffic` = what the PMC passes of THIS device code counted, a stale profile is ignored);
            # beside it the vector ALU's issue rate from the same profile's instruction count per 128 voxels -- since round
            # 3's tables the kernel sits between the two (its stores alone and its arithmetic alone take
            # about the same time).
            issue_rate = prof["valu_insts_per_128_voxels"] * (k_voxels / 128.0) / (k_ms * 1e-3) / 1e9
            roofline.update({
                "traffic": prof.get("hbm_traffic_bytes_per_launch"),
                "valu_issue": {"achieved": round(issue_rate, 2), "peak": VALU_ISSUE_PEAK, "unit": "G wavefront-instructions/s",
                               "frac": round(issue_rate / VALU_ISSUE_PEAK, 4),
                               "note": "VALU instructions issued per second / (1024 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction)"},
                "from_profile": {"file": prof["file"], "csrc_hash": prof["csrc_hash"],
                                 "valu_insts_per_128_voxels": prof["valu_insts_per_128_voxels"],
                                 "valu_insts_per_wave": prof["valu_insts_per_wave"],
                                 "valu_issue_busy_in_profiled_run": prof.get("valu_issue_busy")}})
        else:
            roofline["note"] = ("no rocprofv3 counter profile of this device code (csrc hash %s) is committed: only the HBM "
                                "fraction is reported (DESIGN.md section 5)" % csrc_hash())
        roofline["algorithmic_flop_per_voxel_reference_formulas"] = flop
        line = {
            "metric": "SDF Mvoxels/s (grid_eval+subdivision), 512^3 menger_sponge" if args.config == "c3" else
                      "SDF Mvoxels/s (subdivision + leaf-block grid_eval), 2048^3-effective menger_sponge depth 5",
            "value": round(value, 1), "unit": "Mvoxels/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(wall / args.steps * 1e3, 3),
            "host_enqueue_ms_per_step": round(enqueue_s / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak" if weak else "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": ("menger_sponge depth=%d, %d^3%s: %sadaptive subdivision (grid %d, overlap) + grid_eval of all "
                                    "leaf blocks; %s" % (depth, n, " effective" if not dense_leg else "",
                                                         "dense float4 grid_eval + " if dense_leg else "", SUBDIV_GRID,
                                                         "one object per GPU" if weak else "ONE object over all GPUs")),
                       "baseline_config": args.config,
                       "evaluator": evaluator + (" (per-tape straight-line kernels compiled with hipRTC from the same "
                                                  "op library; bit-identical to the interpreter)" if evaluator == "specialised" else ""),
                       "traversals_in_flight": n_pipes,
                       "replicated_levels": int(pipes[0].replicate),
                       "tape_floats": int(host_tape.size), "tape_instructions": tape.n_instructions,
                       "value_registers": tape.n_registers,
                       "parallelism": ("x-slabs of the one grid; the small leading levels of the hierarchy replicated (every rank classifies them "
                                       "in full, the last one listing the cells the rank owns: no exchange), larger levels: balanced parent slices + "
                                       "one fixed-size RCCL all-gather of the survivors per level; leaf blocks: the rank's share" if world > 1 and not weak else
                                       "one object per rank; one global parent list per level, re-balanced" if weak else "single GPU")},
            "samples_per_step": {"dense": job_dense, "subdivision": subdivision_samples, "leaf_blocks": leaf_samples,
                                 "survivors_per_level_global": totals, "rank0_dense": dense_voxels,
                                 "rank0_leaf_blocks": my_leaves},
            "adaptive": {"what": "B + C, slowest rank: the whole traversal on its own stream (kernels + all-gathers, overlapping A), "
                                 "then every sample of this rank's leaf blocks",
                         "subdivision_ms": round(b_avg, 4), "leaf_blocks_ms": round(c_avg, 4), "ms": round(adaptive_ms, 4),
                         # launch-to-launch spread of the leaf-block kernel in THIS run (rank 0, HIP events, free-running)
                         "leaf_blocks_ms_min_max": [round(min(c_ms), 4), round(max(c_ms), 4)] if c_ms else None,
                         "evaluated_samples": subdivision_samples + leaf_samples,
                         "evaluated_msamples_per_s": round((subdivision_samples + leaf_samples) / adaptive_ms / 1e3, 1),
                         "effective_voxels": n ** 3 * n_objects,
                         "effective_mvoxels_per_s": round(n ** 3 * n_objects / adaptive_ms / 1e3, 1)},
            "roofline": roofline,
        }
        if graph_leg is not None:
            if graph_leg.get("captured"):
                graph_leg["value"] = round(job_samples / graph_leg["ms_per_step"] / 1e3, 1)
                graph_leg["host_enqueue_ms_per_step_direct"] = line["host_enqueue_ms_per_step"]
            line["graph_replay"] = graph_leg
        if dist.exchanging() and world == 1:
            line["config"]["forced_collectives"] = ("CODECAD_AMD_FORCE_COLLECTIVES=1: one rank, a real process group (backend %s): the %d leading "
                                                    "levels are replicated (hu_subdivision_level_owned: every rank classifies them in full and keeps the "
                                                    "cells it owns, no exchange), every other level goes through all_gather_into_tensor -> hu_slice_rows -> "
                                                    "indirect launches" % (torch.distributed.get_backend(), int(pipes[0].replicate)))
        if verified is not None:
            line["verified"] = verified
        if interp_ms:
            line["interpreter_dense_kernel_ms"] = round(avg(interp_ms), 4)
        if hbm_leg is not None:
            line["roofline_hbm"] = hbm_leg
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(host_tape, n if dense_leg else 512)
        real_stdout.write(json.dumps(line) + "\n")
        real_stdout.flush()
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        dist.barrier()
        torch.distributed.destroy_process_group()
    if verified is not None and not (verified["ok"] and verified.get("ranks_ok", True)):
        print("bench: the timed steps' output differs from the oracle: %s" % json.dumps(verified), file=sys.stderr)
        sys.exit(1)


def run_c4(args, real_stdout, rank, world, dev):
    """`--config c4`: BASELINE config 4.  A step = ONE whole `mass_properties` integration of the planetary assembly
    (reference examples/planetary.py:503-560 as its captured tape; driver mass_properties.py:30-229) at resolution 0.25,
    grid 64: levels [(16, 7x7x5), (0.25, 64^3)], every level enqueued with its lists on the device (dist.MassPipeline:
    classification -> per-level integrals; N > 1: balanced parent slices, one fixed-size all-gather per level, one final
    all-reduce of ten doubles), nothing waits for the host inside a step.  `value` = samples classified per second, all
    ranks.  The dominant kernel is the leaf level's launch of k_classify<., true, true, 2> (57.7 M samples); it stores
    next to nothing (40 B of sums per parent), so its roof is the vector ALU's issue rate: `roofline.bound` says so, the
    HBM figure beside it is there for completeness."""
    import numpy as np
    import torch
    import codecad_amd as cc
    from codecad_amd import hip_util, dist, subdivision
    from codecad_amd.hip_util import check
    from codecad_amd.mass_properties import finish, _KEYS

    lib = hip_util.manager.lib
    shape = cc.examples.planetary()
    host_tape = cc.nodes.make_program(shape)
    tape = hip_util.Tape(host_tape, policy="0")
    evaluator = "interpreter"
    if args.evaluator != "interpreter":
        try:
            tape.specialize(hip_util.SPEC_CLASSIFY)      # (the kernels a mass_properties step launches)
            evaluator = "specialised"
        except RuntimeError as e:
            if args.evaluator == "specialised":
                raise
            print("bench: hipRTC specialisation unavailable, using the interpreter: %s" % str(e)[:300], file=sys.stderr)
    main_stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(main_stream)
    stream = main_stream.cuda_stream
    box = shape.bounding_box()
    levels = [(C4_RESOLUTION * cell, tuple(int(v) for v in dims)) for cell, dims in
              subdivision.calculate_block_sizes(box, 3, C4_RESOLUTION, C4_GRID, overlap=False)]
    cells = [d[0] * d[1] * d[2] for _, d in levels]
    capacities = subdivision.first_capacities(cells[:-1], row_bytes=32 + 40)

    def barrier():
        main_stream.synchronize()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    evs = [ctypes.c_void_p() for _ in range(2)]
    for ev in evs:
        check(lib.hu_event_create(ctypes.byref(ev)), "event")
    # warm-up: settles the list capacities at what the lists need + 12 % (launches are sized for the capacity)
    pipe, totals, partial = None, None, None
    for _ in range(max(args.warmup, 1) + 3):
        if pipe is None:
            pipe = dist.MassPipeline(tape, levels, (box.a.x, box.a.y, box.a.z), capacities, dev, stream)
            pipe.timing = {len(levels) - 1: (evs[0], evs[1])}      # HIP events around the leaf level's classification launch
        pipe.enqueue()
        try:
            partial, totals = pipe.finish()
        except dist.Overflow as e:
            capacities, pipe = [int(v * 1.125) + 16 for v in e.needed], None
            continue
        tight = [int(v * 1.125) + 16 for v in pipe.pipe.needed[:-1]]
        if any(c > t for c, t in zip(capacities, tight)):
            capacities, pipe = tight, None
    assert pipe is not None and totals is not None, "list capacities did not settle"
    barrier()
    k_ms = []
    t0 = time.perf_counter()
    for k in range(args.steps):
        pipe.enqueue()
        if args.steps <= 8 or k % max(1, args.steps // 8) == 0:      # (reading an event pair waits for that step: a sample of a long run)
            check(lib.hu_event_synchronize(evs[1]), "sync")
            ms = ctypes.c_float()
            check(lib.hu_event_elapsed_ms(evs[0], evs[1], ctypes.byref(ms)), "elapsed")
            k_ms.append(ms.value)
    barrier()
    wall = time.perf_counter() - t0
    if world > 1:
        wall = float(dist.allreduce_max(torch.tensor([wall], dtype=torch.float64, device=dev)).item())
    partial, totals_timed = pipe.finish()
    assert totals_timed == totals, "the timed steps did not reproduce the warm-up traversal"
    mp = finish(dict(zip(_KEYS, dist.allreduce_sum(partial.clone()).tolist())))
    parents = [1] + [int(t) for t in totals]
    samples = sum(p * c for p, c in zip(parents, cells))
    leaf_parents_here = int(pipe.pipe.lists[-2][0, 0].item())       # this rank's share of the leaf level's parents
    verified = None
    if not args.no_verify:
        verified = verify_c4(np, torch, host_tape, pipe, levels, leaf_parents_here)
        if world > 1:
            bad = torch.tensor([0.0 if verified["ok"] else 1.0], dtype=torch.float64, device=dev)
            verified["ranks_ok"] = bool(float(dist.allreduce_max(bad).item()) == 0.0)
    if rank == 0:
        kernel = "k_classify<%s, true, true, 2>" % ("JitEval" if evaluator == "specialised" else "InterpEval<true>")
        k_avg = sum(k_ms) / len(k_ms)
        k_samples = leaf_parents_here * cells[-1]
        k_bytes = leaf_parents_here * (40.0 + 32.0)      # ten uint32 sums written + one 32-byte parent row read, per parent
        prof = profile_summary(kernel, k_samples, "c4") if world == 1 else None
        roofline = {"bound": "valu_issue", "kernel": kernel, "kernel_ms": round(k_avg, 4), "samples": k_samples,
                    "samples_per_s": round(k_samples / (k_avg * 1e-3), 0), "unit": "G wavefront-instructions/s",
                    "peak": VALU_ISSUE_PEAK, "achieved": None, "frac": None, "traffic": None, "rank": 0,
                    "why": "the kernel stores 40 B per 262 144 samples: its roof is the vector ALU's issue rate (1024 SIMDs x 2.4 GHz / 4 cycles "
                           "per wave64 instruction); the HBM figure is given for completeness",
                    "hbm": {"algorithmic_bytes": k_bytes, "achieved_GBps": round(k_bytes / (k_avg * 1e-3) / 1e9, 3), "peak_GBps": HBM_PEAK_GBS,
                            "frac": round(k_bytes / (k_avg * 1e-3) / 1e9 / HBM_PEAK_GBS, 7)}}
        if prof is not None:
            issue_rate = prof["valu_insts_per_128_voxels"] * (k_samples / 128.0) / (k_avg * 1e-3) / 1e9
            roofline.update({"achieved": round(issue_rate, 2), "frac": round(issue_rate / VALU_ISSUE_PEAK, 4),
                             "traffic": prof.get("hbm_traffic_bytes_per_launch"),
                             "from_profile": {"file": prof["file"], "csrc_hash": prof["csrc_hash"],
                                              "valu_insts_per_128_samples": prof["valu_insts_per_128_voxels"],
                                              "vgprs": prof.get("vgprs"), "waves_per_simd_by_vgprs": prof.get("waves_per_simd"),
                                              "valu_issue_busy_in_profiled_run": prof.get("valu_issue_busy")}})
        else:
            roofline["note"] = ("no rocprofv3 counter profile of this device code (csrc hash %s) is committed: the instruction count "
                                "behind `achieved` is missing" % csrc_hash())
        line = {"metric": "SDF Mvoxels/s (mass_properties), planetary assembly at resolution %.2f, grid %d" % (C4_RESOLUTION, C4_GRID),
                "value": round(samples * args.steps / wall / 1e6, 1), "unit": "Mvoxels/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": round(wall / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
                "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": "mass_properties (volume / centroid / inertia) of the planetary gearbox assembly, resolution %.2f, grid %d: "
                                       "levels %s, ONE object over all GPUs" % (C4_RESOLUTION, C4_GRID, [(s_, list(d)) for s_, d in levels]),
                           "baseline_config": "c4", "evaluator": evaluator, "tape_floats": int(host_tape.size),
                           "tape_instructions": tape.n_instructions,
                           "parallelism": "balanced parent slices + one fixed-size RCCL all-gather per level + one all-reduce of ten doubles"
                                          if world > 1 else "single GPU"},
                "samples_per_step": {"per_level": [p * c for p, c in zip(parents, cells)], "ambiguous_cells_per_level": totals,
                                     "total": samples},
                "result": {"volume": mp.volume, "centroid": [mp.centroid.x, mp.centroid.y, mp.centroid.z]},
                "roofline": roofline}
        if verified is not None:
            line["verified"] = verified
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline_c4(host_tape, levels, box)
        real_stdout.write(json.dumps(line) + "\n")
        real_stdout.flush()
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        dist.barrier()
        torch.distributed.destroy_process_group()
    if verified is not None and not (verified["ok"] and verified.get("ranks_ok", True)):
        print("bench: the timed steps' output differs from the oracle: %s" % json.dumps(verified), file=sys.stderr)
        sys.exit(1)


def verify_c4(np, torch, host_tape, pipe, levels, leaf_parents, blocks=6):
    """The uint32 index sums the last timed step left for `blocks` of this rank's leaf-level parents (64^3 samples each) and
    for the top block, against the CPU oracle's mass_properties kernel (mass_properties.cl:7-56), integer for integer."""
    import oracle
    rng = np.random.default_rng(20261005)
    lp = pipe.pipe
    out = {"ok": True, "against": "oracle/sdf_oracle.c mass_properties (CPU restatement), uint32 sums exact", "blocks": []}
    for level in sorted({0, len(levels) - 1}):
        s, dims = levels[level]
        leaf = level + 1 == len(levels)
        thr = np.float32(0.0 if leaf else s * math.sqrt(3) / 2)
        if level == 0:
            rows, picks = lp.top[1:2].view(torch.float64).cpu().numpy(), [0]
        else:
            rows = lp.lists[level - 1][1:1 + leaf_parents].view(torch.float64).cpu().numpy()
            picks = sorted(set(rng.integers(0, leaf_parents, blocks).tolist())) if leaf_parents else []
        sums = pipe.sums[level].cpu().numpy().view(np.uint32)
        for b in picks:
            # mass_properties.py:86: shifted_corner = box_corner + splat(box_step/2), fp64, cast once
            corner = (rows[b, :3] + s / 2).astype(np.float32)
            want, _, _ = oracle.mass_properties(host_tape, corner, np.float32(s), thr, dims)
            ok = bool(np.array_equal(np.asarray(want, dtype=np.uint32).reshape(-1)[:10], sums[b]))
            out["blocks"].append({"level": level, "parent": int(b), "samples": int(dims[0] * dims[1] * dims[2]), "ok": ok})
            out["ok"] = out["ok"] and ok
    out["samples_checked"] = sum(b["samples"] for b in out["blocks"])
    return out


def cpu_baseline_c4(tape, levels, box):
    """The CPU oracle (a port) on a bounded sample of C4: leaf-level blocks of 64^3 samples from the assembly's middle."""
    import numpy as np
    import oracle
    cores = effective_cores()
    s, dims = levels[-1]
    mid = np.array([(box.a.x + box.b.x) / 2, (box.a.y + box.b.y) / 2, (box.a.z + box.b.z) / 2])
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < 8.0:
        corner = (mid + np.array([(n % 3 - 1) * 16.0, ((n // 3) % 3 - 1) * 16.0, 0.0]) + s / 2).astype(np.float32)
        oracle.mass_properties(tape, corner, np.float32(s), np.float32(0.0), dims)
        n += 1
    dt = time.perf_counter() - t0
    return {"value": round(n * dims[0] * dims[1] * dims[2] / dt / 1e6, 3), "unit": "Mvoxels/s", "cores": 1, "kind": "port",
            "sample": "oracle mass_properties (C restatement, one thread; the box has %d cores) on %d leaf-level blocks of %dx%dx%d samples of this "
                      "tape, %.1f s wall" % (cores, n, dims[0], dims[1], dims[2], dt)}


def verify_outputs(np, torch, host_tape, dense_out, corner, step, x0, n, mine, leaf_out, resolution, origin, leaf_step, leaf_dims,
                   dense_samples=131072, leaf_blocks=256):
    """Compare what the last timed step left in `dense_out` (this rank's x-slab) and `leaf_out` (this rank's leaf blocks,
    block b of it = row b of `mine`) with the CPU oracle, bit for bit (-0 != +0, NaN == NaN): `dense_samples` voxels of
    the slab and `leaf_blocks` whole blocks, drawn with a fixed seed.  The oracle is the checker here -- nothing of it is
    timed or shipped (reference kernels: grid_eval.cl:2-34)."""
    import oracle

    def same(a, b):
        a, b = np.ascontiguousarray(a, dtype=np.float32), np.ascontiguousarray(b, dtype=np.float32)
        return bool(((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all())

    rng = np.random.default_rng(20261005)
    out = {"ok": True, "against": "oracle/sdf_oracle.c (CPU restatement), bit for bit"}
    if dense_out is not None:
        nx = int(dense_out.shape[0])
        idx = np.stack([rng.integers(0, nx, dense_samples), rng.integers(0, n, dense_samples), rng.integers(0, n, dense_samples)], axis=1)
        # sample position of grid_eval.cl:31: corner + step * (float)gid, in binary32
        pts = corner[:3][None, :] + np.float32(step) * (idx + np.array([x0, 0, 0])).astype(np.float32)
        ti = torch.from_numpy(idx).to(dense_out.device)
        got = dense_out[ti[:, 0], ti[:, 1], ti[:, 2]].cpu().numpy()
        ok = same(got, oracle.evaluate_points(host_tape, pts.astype(np.float32)))
        out["dense_voxels"] = {"checked": int(dense_samples), "of": nx * n * n, "ok": ok}
        out["ok"] = out["ok"] and ok
    count = int(mine[0, 0].item())
    if count and leaf_out is not None:
        cells = int(leaf_dims[0]) * int(leaf_dims[1]) * int(leaf_dims[2])
        pick = np.unique(rng.integers(0, count, min(leaf_blocks, count)))
        rows = mine[1:1 + count][torch.from_numpy(pick).to(mine.device)].cpu().numpy()
        got = leaf_out[torch.from_numpy(pick).to(leaf_out.device)].cpu().numpy()
        ok = True
        dims = tuple(int(d) for d in leaf_dims)
        for r, g in zip(rows, got):
            # subdivision.py:100: pos = int_pos * resolution + origin in fp64, cast once
            c = (r[:3].astype(np.float64) * resolution + np.array(origin, dtype=np.float64)).astype(np.float32)
            want = oracle.grid_eval_pymcubes(host_tape, c, np.float32(leaf_step), dims)
            ok = ok and same(g.reshape(-1), np.asarray(want).reshape(-1))
        out["leaf_blocks"] = {"checked": int(len(pick)), "of": count, "samples_checked": int(len(pick)) * cells, "ok": bool(ok)}
        out["ok"] = out["ok"] and bool(ok)
    return out


def hbm_regime(lib, check, hip_util, cc, torch, np, dev, stream, n, evaluator, tapes=None):
    """The HBM-bound regime, measured in this run: the SAME dense kernels on tapes whose arithmetic fits under the
    store stream (box: 5 instructions; sphere: 3; sphere + box: 9; csg_example: 28), float4 (16 B/voxel) and float (4 B/voxel),
    with the tape interpreter and with per-tape code.  Each entry: algorithmic bytes / average kernel time (HIP
    events around ten back-to-back launches, after forty warm ones) against the 8 TB/s peak."""
    out = []
    fptr = ctypes.POINTER(ctypes.c_float)
    e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
    check(lib.hu_event_create(ctypes.byref(e0)), "event")
    check(lib.hu_event_create(ctypes.byref(e1)), "event")
    buf = torch.empty((n, n, n, 4), dtype=torch.float32, device=dev)
    dims = (ctypes.c_uint32 * 3)(n, n, n)
    reps = 10
    for name, shape in tapes or (("box", cc.shapes.box(100)), ("sphere", cc.shapes.sphere(130)), ("sphere_plus_box", cc.examples.sphere_plus_box()),
                                 ("csg_example", cc.examples.csg_example())):
        host_tape = cc.nodes.make_program(shape)
        bb = shape.bounding_box()
        extent = max(bb.b.x - bb.a.x, bb.b.y - bb.a.y, bb.b.z - bb.a.z)
        step = np.float32(extent / n)
        corner = np.array([bb.a.x + extent / n / 2, bb.a.y + extent / n / 2, bb.a.z + extent / n / 2, 0.0], dtype=np.float32)
        for mode in (["interpreter", "specialised"] if evaluator == "specialised" else ["interpreter"]):
            t = hip_util.Tape(host_tape, policy="0")
            if mode == "specialised":
                t = t.specialize(hip_util.SPEC_DENSE)
            for layout, bytes_per_voxel in ((0, 16), (1, 4)):
                def launch():
                    check(lib.hu_grid_eval_slab(t.device_ptr, corner.ctypes.data_as(fptr), step, dims, 0, n, layout, buf.data_ptr(),
                                                stream), "hu_grid_eval_slab")
                # warm: the build that precedes these launches keeps the host busy for a second while the device idles and
                # clocks down; three launches did not bring it back (sponge(4): 0.49 ms here against 0.405 in the timed steps)
                for _ in range(40):
                    launch()
                check(lib.hu_event_record(e0, stream), "record")
                for _ in range(reps):
                    launch()
                check(lib.hu_event_record(e1, stream), "record")
                check(lib.hu_event_synchronize(e1), "sync")
                ms = ctypes.c_float()
                check(lib.hu_event_elapsed_ms(e0, e1, ctypes.byref(ms)), "elapsed")
                k_ms = ms.value / reps
                gbs = n ** 3 * bytes_per_voxel / (k_ms * 1e-3) / 1e9
                kernel = "k_grid_eval<%s, %d, 2>" % ("JitEval" if mode == "specialised" else "InterpEval<%s>" % ("true" if layout else "false"), layout)
                out.append({"tape": name, "tape_instructions": t.n_instructions, "evaluator": mode, "kernel": kernel,
                            "bytes": n ** 3 * bytes_per_voxel, "ms": round(k_ms, 4), "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS,
                            "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "gvoxels_per_s": round(n ** 3 / k_ms / 1e6, 1)})
    return out


# FLOPs per instruction of the reference's formulas (SURVEY.md section 8(d) convention): 1 per
# add/mul/compare-select/abs/copysign, 2 per fma, sqrt 1, divide 1, as executed on the common path.
_FLOP = {0: 0, 1: 0, 2: 0, 3: 14, 4: 9, 5: 90, 6: 0, 7: 12, 8: 1, 9: 4, 10: 120, 11: 39, 12: 39, 13: 40,
         14: 1, 15: 1, 16: 1, 17: 3, 18: 15, 19: 90, 20: 90, 21: 160, 22: 14, 23: 8, 24: 120, 25: 2,
         26: 2, 27: 6, 28: 4}
_PARAMS = {0: 0, 1: 0, 2: 0, 3: 2, 4: 1, 5: 2, 7: 1, 8: 0, 9: 0, 10: 2, 11: 7, 12: 7, 13: 4, 14: 0, 15: 0,
           16: 1, 17: 1, 18: 3, 19: 1, 20: 1, 21: 2, 22: 1, 23: 0, 24: 3, 25: 0, 26: 1, 27: 1, 28: 1}


def profile_summary(kernel, samples_per_launch, config):
    """The committed rocprofv3 PMC summary (profiles/*kernels_summary.json: tools/collect_kernels.sh +
    tools/summarize_kernels.py, counters per kernel per launch) of `kernel` in `config` -- only if it was taken on THIS
    device code (csrc hash); else None.  `samples_per_launch`: what one launch of the kernel evaluates in this config."""
    import glob
    want = csrc_hash()
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*kernels_summary.json"))):
        try:
            d = json.load(open(f))
            if d.get("csrc_hash") != want:
                continue
            entry = d[config][kernel]
            c = entry["counters_per_launch"]
            best = {"file": os.path.basename(f), "csrc_hash": want,
                    "valu_insts_per_wave": round(c["SQ_INSTS_VALU"] / c["SQ_WAVES"], 2),
                    # per 128 samples (a wavefront-instruction's worth of two-voxel lanes): the figure that compares across
                    # kernels and rounds -- round 1 dense: 721.3, round 2: 366.7
                    "valu_insts_per_128_voxels": round(c["SQ_INSTS_VALU"] / (samples_per_launch / 128.0), 2),
                    "hbm_traffic_bytes_per_launch": entry.get("hbm_traffic_bytes_per_launch"),
                    "valu_issue_busy": entry.get("valu_issue_busy"), "profiled_avg_ms": round(entry["avg_ns"] / 1e6, 4),
                    "vgprs": entry.get("vgprs"), "waves_per_simd": entry.get("waves_per_simd_by_vgprs")}
        except (ValueError, KeyError, ZeroDivisionError, TypeError):
            continue
    return best


def tape_flop(tape):
    pc, total = 0, 0
    while pc < len(tape):
        op = int(tape[pc]) // 512
        pc += 1
        if op == 6:
            cnt = int(tape[pc])
            total += 25 * cnt
            pc += 1 + 2 * cnt
        else:
            total += _FLOP[op]
            pc += _PARAMS[op]
        if op == 0:
            break
    return total


def effective_cores():
    """Host cores this process may actually use: affinity mask, capped by the cgroup CPU quota."""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return cores


def cpu_baseline(tape, n):
    """The CPU oracle (a port: the reference's OpenCL code cannot be built here) on a bounded
    sample of workload A: the first `planes` x-planes of the same n^3 grid, all host cores."""
    import numpy as np
    import oracle
    cores = effective_cores()
    step = np.float32(1.0 / n)
    corner = [-0.5 + 0.5 / n] * 3
    oracle.grid_eval(tape, corner, step, (1, 64, 64), threads=1)  # load + warm
    t0 = time.perf_counter()
    oracle.grid_eval(tape, corner, step, (2, n, n), threads=1)
    one = 2 * n * n / (time.perf_counter() - t0)
    planes = max(cores, min(n, int(one * cores * 6.0 / (n * n))))   # aim for ~6 s wall
    t0 = time.perf_counter()
    oracle.grid_eval(tape, corner, step, (planes, n, n), threads=cores)
    dt = time.perf_counter() - t0
    return {"value": round(planes * n * n / dt / 1e6, 2), "unit": "Mvoxels/s", "cores": cores, "kind": "port",
            "sample": "oracle grid_eval (C restatement, OpenMP over x) of the first %d of %d x-planes of the "
                      "%d^3 grid of this tape, %.1f s wall" % (planes, n, n, dt),
            "single_thread_mvoxels_s": round(one / 1e6, 3)}


if __name__ == "__main__":
    main()
