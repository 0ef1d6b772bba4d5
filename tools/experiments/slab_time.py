import ctypes, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import codecad_amd as cc
from codecad_amd import hip_util
from codecad_amd.hip_util import check
hip = hip_util.manager
lib = hip.lib
shape = cc.examples.sponge(4)
tape = hip_util.Tape(cc.nodes.make_program(shape), policy="0").specialize()
n = 512
step = np.float32(1.0 / n)
corner = np.array([-0.5 + 0.5 / n] * 3 + [0.0], np.float32)
dims = (ctypes.c_uint32 * 3)(n, n, n)
fptr = ctypes.POINTER(ctypes.c_float)
out = torch.empty((n, n, n, 4), dtype=torch.float32, device="cuda")
for count in (512, 256, 128, 64, 32):
    for _ in range(3):
        check(lib.hu_grid_eval_slab(tape.device_ptr, corner.ctypes.data_as(fptr), step, dims, 64, count if count < 512 else 512 - 64, 0, out.data_ptr(), None), "slab")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    x0 = 0 if count == 512 else 64
    e0.record()
    for _ in range(20):
        check(lib.hu_grid_eval_slab(tape.device_ptr, corner.ctypes.data_as(fptr), step, dims, x0, count, 0, out.data_ptr(), None), "slab")
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("slab of %3d planes: %.4f ms  (%.1f Gvoxel/s, %.2f of the full grid's rate)" % (count, ms, count * n * n / ms / 1e6, (count * n * n / ms) / (n ** 3 / 0.41)), flush=True)
