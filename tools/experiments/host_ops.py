#!/usr/bin/env python3
"""Run on the GPU box: what the host-side operations of a bench step cost each (us per call, 2000 calls)."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from codecad_amd import hip_util  # noqa: E402
from codecad_amd.hip_util import check  # noqa: E402

lib = hip_util.manager.lib
dev = torch.device("cuda", 0)
main, side = torch.cuda.current_stream(dev), torch.cuda.Stream(device=dev, priority=-1)
ev = ctypes.c_void_p()
check(lib.hu_event_create(ctypes.byref(ev)), "event")
tev = torch.cuda.Event()
buf = torch.zeros((1024, 4), dtype=torch.int32, device=dev)
head = buf[:1]


def timed(name, fn, n=2000):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    dt = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    print("%-44s %6.2f us" % (name, dt), flush=True)


def ctx():
    with torch.cuda.stream(side):
        pass


timed("hu_event_record (ctypes)", lambda: lib.hu_event_record(ev, side.cuda_stream))
timed("torch Event.record(stream)", lambda: tev.record(main))
timed("stream.wait_event", lambda: side.wait_event(tev))
timed("main.wait_stream(side)", lambda: main.wait_stream(side))
timed("with torch.cuda.stream(side): pass", ctx)
timed("head.zero_() (pre-sliced view)", head.zero_)
timed("buf[:1].zero_()", lambda: buf[:1].zero_())
timed("buf.data_ptr()", buf.data_ptr)
timed("side.cuda_stream", lambda: side.cuda_stream)
