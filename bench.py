#!/usr/bin/env python3
"""bench.py -- SDF Mvoxels/s (grid_eval + subdivision) on the 512^3 Menger sponge.

One "step" is one pass of the hot path over one batch of synthetic input (no RNG: the only
input is the CSG tree), per GPU:
  A. dense grid_eval (float4) of sponge(4) on the 512^3 cell-centred grid          [k_grid_eval<0>]
  B. adaptive subdivision of sponge(4) at resolution 1/512, grid 16, overlapping
     leaf samples: levels [(240,3^3),(15,16^3),(1,16^3)], survivors compacted by the
     wavefront ballot scan                                                      [k_classify<0,1>]
  C. grid_eval of ALL surviving 16^3 leaf blocks in one launch (float, PyMCubes layout,
     what the reference's mesh pipeline does block by block)                [k_grid_eval_blocks<1>]
`value` counts SDF samples actually evaluated (A + B + C) per second over all ranks; nothing
is cached between steps and the output buffers are rewritten every step.  B is latency (tiny
kernels, counter reads, all-gathers), so it runs on a second HIP stream concurrently with A; C
waits for both.

N > 1 (torchrun, one rank per GPU, RCCL): weak scaling -- the job is N sponges; the dense
grid is x-slab sharded (rank r owns object r's 512^3 slab, no collective); the subdivision
hierarchy of all N objects is ONE global parent list per level, cut into balanced slices,
with a variable-length RCCL all-gather of the survivors between levels (codecad_amd/dist.py).

The JSON line also carries `roofline` for the dominant kernel (k_grid_eval<0>, measured with
HIP events on its own stream) and `cpu_baseline` (the CPU oracle on a bounded sample).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N = 512
SPONGE_DEPTH = 4
SUBDIV_GRID = 16
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_PEAK_TFLOPS = 157.3   # vector FP32, FMA-counted


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--n", type=int, default=N, help="grid edge (default 512)")
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

    import numpy as np
    import torch
    import codecad_amd as cc
    from codecad_amd import hip_util, dist
    from codecad_amd.hip_util import check
    import ctypes

    rank, world = dist.init()
    local = dist.local_device()
    torch.cuda.set_device(local)
    hip_util.manager.use_device(local)
    m = hip_util.manager
    lib = m.lib
    dev = torch.device("cuda", local)
    n = args.n

    shape = cc.examples.sponge(SPONGE_DEPTH)
    tape = cc.nodes.make_program_buffer(shape)
    interp_tape = hip_util.Tape(tape.host_tape, policy="0")   # the same program, always interpreted (reported beside the headline)
    tape = interp_tape
    evaluator = "interpreter"
    if args.evaluator != "interpreter":
        try:
            tape = hip_util.Tape(tape.host_tape, policy="0").specialize()   # ~1 s of hipRTC (ms from the disk cache), outside the timed region
            evaluator = "specialised"
        except RuntimeError as e:
            if args.evaluator == "specialised":
                raise
            print("bench: hipRTC specialisation unavailable, using the interpreter: %s" % str(e)[:300], file=sys.stderr)
    # Everything is enqueued on torch's current stream: the survivor lists travel through
    # torch.distributed (RCCL orders its collectives against that stream), so one stream gives
    # the kernel -> all-gather -> kernel dependencies without extra synchronisation.
    queue = m.wrap_stream(torch.cuda.current_stream().cuda_stream)
    stream = queue.handle
    # The subdivision (B) is a chain of tiny kernels, 4-byte counter reads and (N > 1) all-gathers: latency,
    # not work.  It runs on a second, high-priority stream so that this latency hides behind the dense
    # kernel (A) instead of following it; C waits for both.
    main_stream = torch.cuda.current_stream()
    side_stream = torch.cuda.Stream(device=dev, priority=-1)
    side = side_stream.cuda_stream

    # ---- A: dense grid ------------------------------------------------------------------
    step_f = np.float32(1.0 / n)
    corner = np.array([-0.5 + 0.5 / n] * 3 + [0.0], dtype=np.float32)
    dims = (ctypes.c_uint32 * 3)(n, n, n)
    dense_out = torch.empty((n, n, n, 4), dtype=torch.float32, device=dev)
    dense_voxels = n ** 3

    # ---- B: subdivision hierarchy ---------------------------------------------------------
    resolution = 1.0 / n
    box = shape.bounding_box().expanded_additive(resolution / 2)
    levels = cc.subdivision.calculate_block_sizes(box, 3, resolution, SUBDIV_GRID, True)
    origin = (ctypes.c_double * 3)(box.a.x, box.a.y, box.a.z)
    counter = torch.zeros(1, dtype=torch.int32, device=dev)
    stats = {"samples": 0, "leaves": 0, "level_counts": None}
    capacity = [1 << 16] * len(levels)

    def classify(level, parents):
        int_step, ldims = levels[level]
        k = int(parents.shape[0])
        cells = int(ldims[0]) * int(ldims[1]) * int(ldims[2])
        stats["samples"] += k * cells
        if k == 0:
            return parents[:0]
        d = (ctypes.c_uint32 * 3)(int(ldims[0]), int(ldims[1]), int(ldims[2]))
        box_step = int_step * resolution
        thr = box_step * math.sqrt(3) / 2
        parents = parents.contiguous()
        while True:
            children = torch.empty((capacity[level], 4), dtype=torch.int32, device=dev)
            check(lib.hu_memset(counter.data_ptr(), 0, 4, side), "memset")
            check(lib.hu_subdivision_level(tape.device_ptr, parents.data_ptr(), k, int(int_step), d, 3,
                                           resolution, origin, np.float32(box_step), np.float32(thr),
                                           counter.data_ptr(), children.data_ptr(), capacity[level], side),
                  "hu_subdivision_level")
            count = int(counter.item())   # synchronises the side stream only
            if count <= capacity[level]:
                return children[:count]
            capacity[level] = int(count * 1.25)

    top = torch.zeros((world, 4), dtype=torch.int32, device=dev)
    top[:, 3] = torch.arange(world, dtype=torch.int32, device=dev)   # one hierarchy per object
    leaf_int_step, leaf_dims = levels[-1]
    leaf_cells = int(leaf_dims[0]) * int(leaf_dims[1]) * int(leaf_dims[2])
    ld = (ctypes.c_uint32 * 3)(int(leaf_dims[0]), int(leaf_dims[1]), int(leaf_dims[2]))
    leaf_out = [None]
    level_hints = []   # per-level survivor counts of the previous step: one collective per level (dist.allgather_rows)

    ev0, ev1, ev2, evb0, evb1, evc0 = (ctypes.c_void_p() for _ in range(6))
    for ev in (ev0, ev1, ev2, evb0, evb1, evc0):
        check(lib.hu_event_create(ctypes.byref(ev)), "event")
    dense_ms, adaptive_ms = [], []

    def one_step(timed):
        # A
        check(lib.hu_event_record(ev0, stream), "record")
        check(lib.hu_grid_eval(tape.device_ptr, corner.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), step_f,
                               dims, dense_out.data_ptr(), stream), "hu_grid_eval")
        check(lib.hu_event_record(ev1, stream), "record")
        # B, concurrently with A, on the side stream (torch ops and collectives follow the stream context)
        stats["samples"] = 0
        with torch.cuda.stream(side_stream):
            check(lib.hu_event_record(evb0, side), "record")
            leaves, counts = dist.run_levels(top, len(levels) - 1, classify, hints=level_hints)
            # this rank's balanced share of the global leaf list
            b, e = dist.balanced_slice(int(leaves.shape[0]), rank, world)
            mine = leaves[b:e].contiguous()
            check(lib.hu_event_record(evb1, side), "record")
        stats["level_counts"] = counts
        mine.record_stream(main_stream)       # allocated on the side stream, consumed on the main one
        main_stream.wait_stream(side_stream)
        # C, after A and B
        k = int(mine.shape[0])
        stats["leaves"] = k
        if leaf_out[0] is None or leaf_out[0].shape[0] < k:
            leaf_out[0] = torch.empty((int(k * 1.1) + 1, leaf_cells), dtype=torch.float32, device=dev)
        check(lib.hu_event_record(evc0, stream), "record")
        check(lib.hu_grid_eval_blocks(tape.device_ptr, mine.data_ptr(), k, resolution, origin,
                                      np.float32(leaf_int_step * resolution), ld, 1, leaf_out[0].data_ptr(), stream),
              "hu_grid_eval_blocks")
        check(lib.hu_event_record(ev2, stream), "record")
        if timed:
            check(lib.hu_event_synchronize(ev2), "sync")
            ms = ctypes.c_float()
            check(lib.hu_event_elapsed_ms(ev0, ev1, ctypes.byref(ms)), "elapsed")
            dense_ms.append(ms.value)
            check(lib.hu_event_elapsed_ms(evb0, evb1, ctypes.byref(ms)), "elapsed")
            b_ms = ms.value                    # B: subdivision with its per-level counter reads (and all-gathers)
            check(lib.hu_event_elapsed_ms(evc0, ev2, ctypes.byref(ms)), "elapsed")
            adaptive_ms.append(b_ms + ms.value)   # + C: every sample of every leaf block

    def barrier():
        queue.synchronize()
        side_stream.synchronize()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step(False)
    # the interpreter on the same dense grid, timed the same way (reported, not part of `value`)
    interp_ms = []
    for i in range(3):
        check(lib.hu_event_record(ev0, stream), "record")
        check(lib.hu_grid_eval(interp_tape.device_ptr, corner.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), step_f,
                               dims, dense_out.data_ptr(), stream), "hu_grid_eval")
        check(lib.hu_event_record(ev1, stream), "record")
        check(lib.hu_event_synchronize(ev1), "sync")
        ms = ctypes.c_float()
        check(lib.hu_event_elapsed_ms(ev0, ev1, ctypes.byref(ms)), "elapsed")
        if i:
            interp_ms.append(ms.value)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step(True)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        elapsed = float(dist.allreduce_max(torch.tensor([elapsed], dtype=torch.float64, device=dev)).item())

    per_rank_samples = dense_voxels + stats["samples"] + stats["leaves"] * leaf_cells
    job_samples = float(dist.allreduce_sum(torch.tensor([per_rank_samples], dtype=torch.float64, device=dev)).item())
    value = job_samples * args.steps / elapsed / 1e6

    if rank == 0:
        flop = tape_flop(tape.host_tape)
        dense_avg_ms = sum(dense_ms) / len(dense_ms)
        alg_bytes = dense_voxels * 16.0
        achieved = alg_bytes / (dense_avg_ms * 1e-3) / 1e9
        line = {
            "metric": "SDF Mvoxels/s (grid_eval+subdivision), 512^3 menger_sponge",
            "value": round(value, 1), "unit": "Mvoxels/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "menger_sponge depth=4, %d^3: dense float4 grid_eval + adaptive subdivision "
                                   "(grid %d, overlap) + grid_eval of all leaf blocks; one object per GPU" % (n, SUBDIV_GRID),
                       "evaluator": evaluator + (" (per-tape straight-line kernels compiled with hipRTC from the same "
                                                  "op library; bit-identical to the interpreter)" if evaluator == "specialised" else ""),
                       "tape_floats": int(tape.host_tape.size), "tape_instructions": tape.n_instructions,
                       "value_registers": tape.n_registers, "parallelism": "x-slab/object per rank, "
                       "balanced parent slices + RCCL all-gather of survivors per level" if world > 1 else "single GPU"},
            "samples_per_step_per_gpu": {"dense": dense_voxels, "subdivision": stats["samples"],
                                         "leaf_blocks": stats["leaves"] * leaf_cells,
                                         "survivors_per_level_global": stats["level_counts"]},
            "interpreter_dense_kernel_ms": round(sum(interp_ms) / len(interp_ms), 4),
            # adaptive runs report effective voxels/s (N^3 / time) beside evaluated samples/s
            "adaptive": {"what": "B + C on this rank: subdivision to the leaf blocks (on its own stream, overlapping A), then every "
                                 "sample of every leaf block; ms = B's stream time + C's kernel time",
                         "ms": round(sum(adaptive_ms) / len(adaptive_ms), 4),
                         "evaluated_samples": stats["samples"] + stats["leaves"] * leaf_cells,
                         "evaluated_msamples_per_s": round((stats["samples"] + stats["leaves"] * leaf_cells)
                                                           / (sum(adaptive_ms) / len(adaptive_ms)) / 1e3, 1),
                         "effective_voxels": n ** 3,
                         "effective_mvoxels_per_s": round(n ** 3 / (sum(adaptive_ms) / len(adaptive_ms)) / 1e3, 1)},
            "roofline": {"bound": "hbm", "kernel": "k_grid_eval<%s, 0, 2>" % ("JitEval" if evaluator == "specialised" else "InterpEval<false>"),
                         "achieved": round(achieved, 2),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": measured_traffic(n, evaluator), "kernel_ms": round(dense_avg_ms, 4),
                         "voxels_per_s": round(dense_voxels / (dense_avg_ms * 1e-3), 0),
                         "note": "this tape is FP32-VALU-bound, not HBM-bound: see valu_* fields; valu_flop_per_voxel is the "
                                 "ALGORITHMIC count of the reference's formulas (SURVEY.md section 8(d) convention, FMA = 2), "
                                 "not instructions executed: the kernel's reduced transformation forms execute fewer",
                         "valu_flop_per_voxel": flop,
                         "valu_achieved_tflops": round(dense_voxels * flop / (dense_avg_ms * 1e-3) / 1e12, 2),
                         "valu_peak_tflops": FP32_PEAK_TFLOPS,
                         "valu_frac": round(dense_voxels * flop / (dense_avg_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4),
                         # issue-slot utilisation of the vector ALUs from the committed rocprofv3 counters (like `traffic`)
                         "valu_busy_measured": measured_valu_busy(n, evaluator)},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(tape.host_tape, n)
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        torch.distributed.destroy_process_group()


# FLOPs per instruction of the canonical arithmetic (DESIGN.md "Algorithmic FLOPs"): 1 per
# add/mul/compare-select/abs/copysign, 2 per fma, sqrt 1, divide 1, as executed on the common path.
_FLOP = {0: 0, 1: 0, 2: 0, 3: 14, 4: 9, 5: 90, 6: 0, 7: 12, 8: 1, 9: 4, 10: 120, 11: 39, 12: 39, 13: 40,
         14: 1, 15: 1, 16: 1, 17: 3, 18: 15, 19: 90, 20: 90, 21: 160, 22: 14, 23: 8, 24: 120, 25: 2,
         26: 2, 27: 6, 28: 4}
_PARAMS = {0: 0, 1: 0, 2: 0, 3: 2, 4: 1, 5: 2, 7: 1, 8: 0, 9: 0, 10: 2, 11: 7, 12: 7, 13: 4, 14: 0, 15: 0,
           16: 1, 17: 1, 18: 3, 19: 1, 20: 1, 21: 2, 22: 1, 23: 0, 24: 3, 25: 0, 26: 1, 27: 1, 28: 1}


def measured_valu_busy(n, evaluator="specialised"):
    """Fraction of the VALU issue slots the dense kernel used, from the same committed PMC passes:
    SQ_ACTIVE_INST_VALU * 4 / (1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs); None without a matching profile."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*summary.json"))):
        try:
            d = json.load(open(f))
            c = d["dense_kernel_counters_per_launch"]
            if d.get("grid_edge", 512) == n and d.get("evaluator", "interpreter") == evaluator:
                best = round(c["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (c["GRBM_GUI_ACTIVE"] / 8), 3)
        except (ValueError, KeyError, ZeroDivisionError):
            continue
    return best


def measured_traffic(n, evaluator="specialised"):
    """HBM bytes per launch of the dense kernel from the committed rocprofv3 PMC passes
    (profiles/*_summary.json, produced by tools/collect_profiles.sh for this same kernel and
    grid); None when no profile of this grid size is present."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*summary.json"))):
        try:
            d = json.load(open(f))
        except ValueError:
            continue
        if d.get("grid_edge", 512) == n and d.get("evaluator", "interpreter") == evaluator and \
                "hbm_traffic_bytes_per_launch" in d:
            best = d["hbm_traffic_bytes_per_launch"]
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
                      "same %d^3 sponge(4) grid, %.1f s wall" % (planes, n, n, dt),
            "single_thread_mvoxels_s": round(one / 1e6, 3)}


if __name__ == "__main__":
    main()
