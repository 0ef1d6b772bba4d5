"""`hip_util.manager`: the process-wide device context and kernel launcher.

Same shape as the reference's `cl_util.opencl_manager` singleton (reference
cl_util/opencl_manager.py:87-144): `.queue`, `.k.<kernel>(global_size, local_size, *args,
wait_for=None) -> Event`.  Differences that follow from the hardware mapping:
  * nothing is JIT-compiled: the kernels are ahead-of-time gfx950 code in libhip_util.so,
    so there are no compile units to register and the first launch costs nothing extra;
  * the device is opened lazily on first use, not at import (one process per GPU: the
    ordinal comes from LOCAL_RANK, see codecad_amd.dist);
  * the queue is an in-order HIP stream; `wait_for=` events from other streams are honoured
    with hipStreamWaitEvent.
There is no CPU fallback: without the library or without a GPU every entry point raises.
"""
import ctypes
import os

import numpy

from . import _lib


class HipError(RuntimeError):
    pass


def check(rc, what=""):
    if rc != 0:
        msg = _lib.load().hu_last_error()
        raise HipError("%s failed (%d): %s" % (what or "hip_util call", rc, msg.decode() if msg else "?"))


# kernel families of per-tape code (include/hip_util.h hu_spec_group)
from ._lib import SPEC_DENSE, SPEC_BLOCKS, SPEC_CLASSIFY, SPEC_RENDER, SPEC_ALL, SPEC_KERNELS  # noqa: E402,F401


def _note(scene, global_size, factor=1, group=SPEC_DENSE):
    """Tell a Tape how many samples a launch is about to evaluate, and with which kernel family (Tape.note_samples:
    tiered specialisation builds the family in use first)."""
    note = getattr(scene, "note_samples", None)
    if note is not None:
        n = factor
        for v in global_size:
            n *= int(v)
        note(n, group)


def _ptr(obj):
    """Device pointer of a Buffer / Tape / torch tensor / int."""
    if obj is None:
        return None
    if hasattr(obj, "device_ptr"):
        return obj.device_ptr
    if hasattr(obj, "data_ptr"):
        return obj.data_ptr()
    return int(obj)


def _float4(corner):
    a = numpy.zeros(4, dtype=numpy.float32)
    if isinstance(corner, numpy.ndarray) and corner.dtype.names:
        vals = [corner[n] for n in corner.dtype.names]
    else:
        vals = list(numpy.asarray(corner).reshape(-1))
    a[:min(4, len(vals))] = vals[:4]
    return a


def _dims3(global_size):
    try:
        dims = [int(v) for v in global_size]
    except TypeError:
        dims = [int(global_size)]
    if not 1 <= len(dims) <= 3 or any(d < 1 for d in dims):
        raise ValueError("global size must be 1-3 positive integers, got %r" % (global_size,))
    return (ctypes.c_uint32 * 3)(*(dims + [1, 1])[:3])


class Stream:
    """An in-order HIP stream (the reference's command queue)."""

    def __init__(self, manager, handle=None, owned=True):
        self.manager = manager
        if handle is None:
            h = ctypes.c_void_p()
            check(manager.lib.hu_stream_create(ctypes.byref(h)), "hu_stream_create")
            handle = h.value
        self.handle = handle
        self._owned = owned

    @property
    def context(self):
        return self.manager

    def synchronize(self):
        check(self.manager.lib.hu_stream_synchronize(self.handle), "hu_stream_synchronize")

    finish = synchronize  # pyopencl spelling


class _Profile:
    def __init__(self, event):
        self._event = event

    @property
    def start(self):
        return self._event._ns(self._event._start)

    @property
    def end(self):
        return self._event._ns(self._event._stop)


class Event:
    """Completion marker of one enqueued operation: `.wait()`, `.profile.start/.end` (ns)."""

    def __init__(self, manager, stream):
        self.manager = manager
        self.stream = stream
        self._start = manager._new_event()
        self._stop = manager._new_event()
        check(manager.lib.hu_event_record(self._start, stream.handle), "hu_event_record")
        self.profile = _Profile(self)

    def _done(self):
        check(self.manager.lib.hu_event_record(self._stop, self.stream.handle), "hu_event_record")
        return self

    def wait(self):
        check(self.manager.lib.hu_event_synchronize(self._stop), "hu_event_synchronize")

    def elapsed_ms(self):
        self.wait()
        ms = ctypes.c_float()
        check(self.manager.lib.hu_event_elapsed_ms(self._start, self._stop, ctypes.byref(ms)), "hu_event_elapsed_ms")
        return ms.value

    def _ns(self, ev):
        self.wait()
        ms = ctypes.c_float()
        check(self.manager.lib.hu_event_elapsed_ms(self.manager._epoch, ev, ctypes.byref(ms)), "hu_event_elapsed_ms")
        return int(ms.value * 1e6)

    def __del__(self):
        try:
            self.manager._recycle_event(self._start)
            self.manager._recycle_event(self._stop)
        except Exception:
            pass


class _Kernels:
    """`manager.k.<name>(global_size, local_size, *args, wait_for=None)`.

    Positional arguments are those of the reference OpenCL kernels
    (grid_eval.cl:2-4,23-25; subdivision.cl:12-16; mass_properties.cl:7-12).
    `local_size` is accepted and ignored (the reference always passes None; the workgroup
    shape is chosen from the tape's register count).
    """

    def __init__(self, manager):
        self._m = manager

    def _launch(self, wait_for, queue, fn):
        m = self._m
        stream = queue or m.queue
        for ev in (wait_for or ()):
            if getattr(ev, "stream", None) is not stream and hasattr(ev, "_stop"):
                check(m.lib.hu_stream_wait_event(stream.handle, ev._stop), "hu_stream_wait_event")
        ev = Event(m, stream)
        fn(stream.handle)
        return ev._done()

    def grid_eval(self, global_size, local_size, scene, box_corner, box_step, output, wait_for=None, queue=None):
        _note(scene, global_size)
        c, d = _float4(box_corner), _dims3(global_size)
        return self._launch(wait_for, queue, lambda s: check(self._m.lib.hu_grid_eval(
            _ptr(scene), c.ctypes.data_as(_lib._f4), float(box_step), d, _ptr(output), s), "hu_grid_eval"))

    def grid_eval_pymcubes(self, global_size, local_size, scene, box_corner, box_step, output, wait_for=None, queue=None):
        _note(scene, global_size)
        c, d = _float4(box_corner), _dims3(global_size)
        return self._launch(wait_for, queue, lambda s: check(self._m.lib.hu_grid_eval_pymcubes(
            _ptr(scene), c.ctypes.data_as(_lib._f4), float(box_step), d, _ptr(output), s), "hu_grid_eval_pymcubes"))

    def subdivision_step(self, global_size, local_size, scene, box_corner, box_step, distance_threshold,
                         intersecting_counter, list_buffer, wait_for=None, queue=None):
        _note(scene, global_size, group=SPEC_CLASSIFY)
        c, d = _float4(box_corner), _dims3(global_size)
        return self._launch(wait_for, queue, lambda s: check(self._m.lib.hu_subdivision_step(
            _ptr(scene), c.ctypes.data_as(_lib._f4), float(box_step), float(distance_threshold), d,
            _ptr(intersecting_counter), _ptr(list_buffer), s), "hu_subdivision_step"))

    def mass_properties(self, global_size, local_size, shape, box_corner, box_step, distance_threshold,
                        sums, intersecting_counter, list_buffer, wait_for=None, queue=None):
        _note(shape, global_size, group=SPEC_CLASSIFY)
        c, d = _float4(box_corner), _dims3(global_size)
        return self._launch(wait_for, queue, lambda s: check(self._m.lib.hu_mass_properties(
            _ptr(shape), c.ctypes.data_as(_lib._f4), float(box_step), float(distance_threshold), d,
            _ptr(sums), _ptr(intersecting_counter), _ptr(list_buffer), s), "hu_mass_properties"))

    def ray_caster(self, global_size, local_size, scene, origin, forward, up, right, pixel_tolerance, box_radius,
                   min_distance, max_distance, floor_z, render_options, output, wait_for=None, queue=None):
        """rendering/ray_caster.cl:146-159; global_size = (width, height)."""
        _note(scene, global_size, 50, SPEC_RENDER)   # a march is tens of evaluations per pixel
        v = [_float4(x) for x in (origin, forward, up, right)]
        return self._launch(wait_for, queue, lambda s: check(self._m.lib.hu_ray_caster(
            _ptr(scene), *[x.ctypes.data_as(_lib._f4) for x in v], float(pixel_tolerance), float(box_radius),
            float(min_distance), float(max_distance), float(floor_z), int(render_options), int(global_size[0]),
            int(global_size[1]), _ptr(output), s), "hu_ray_caster"))

    def bitmap(self, global_size, local_size, scene, origin, step_size, output, wait_for=None, queue=None):
        """rendering/bitmap.cl:1-4; global_size = (width, height)."""
        _note(scene, global_size, group=SPEC_RENDER)
        o = _float4(origin)
        return self._launch(wait_for, queue, lambda s: check(self._m.lib.hu_bitmap(
            _ptr(scene), o.ctypes.data_as(_lib._f4), float(step_size), int(global_size[0]), int(global_size[1]),
            _ptr(output), s), "hu_bitmap"))

    def process_polygon(self, global_size, local_size, box_corner, box_step, corners, vertices, links, starts,
                        start_counter, wait_for=None, queue=None):
        """rendering/polygon2d.cl:82-93; global_size = (gx-1, gy-1, 2)."""
        c = _float4(box_corner)
        assert int(global_size[2]) == 2
        g = (ctypes.c_uint32 * 2)(int(global_size[0]), int(global_size[1]))
        return self._launch(wait_for, queue, lambda s: check(self._m.lib.hu_process_polygon(
            c.ctypes.data_as(_lib._f4), float(box_step), _ptr(corners), g, _ptr(vertices), _ptr(links), _ptr(starts),
            _ptr(start_counter), s), "hu_process_polygon"))


class _BlockPool:
    """Recycles device (or pinned host) allocations.

    hipMalloc / hipFree cost 0.1-1 ms each and hipFree synchronises the device; the level-synchronous
    drivers allocate a handful of lists per level, which made `mass_properties` spend two thirds of its
    wall time in the allocator (sponge(4) at 1/512: 1.1 ms of kernels, 3.4 ms wall).  Blocks are
    rounded up to a size class and returned to the pool of the STREAM they were used on: the next user
    enqueues on the same in-order stream, so work still in flight on the block finishes first.
    CODECAD_AMD_POOL_MB caps the cached bytes (default 4096, 0 disables pooling)."""

    def __init__(self, alloc, free, limit_bytes, fence=None, wait=None):
        """fence(stream_key) -> token, wait(token): for memory the HOST writes (pinned blocks), where handing a
        block to the next user of the same stream is not enough -- the host does not run in stream order.  A
        block is given back together with a fence recorded on its stream and taken only after the fence passed."""
        self._alloc, self._free, self.limit = alloc, free, limit_bytes
        self._fence, self._wait = fence, wait
        self.blocks = {}   # (stream key, size class) -> [(pointer, fence token or None)]
        self.cached = 0

    @staticmethod
    def size_class(nbytes):
        nbytes = max(int(nbytes), 1)
        if nbytes <= (1 << 20):
            return max(256, 1 << (nbytes - 1).bit_length())
        return (nbytes + (1 << 20) - 1) & ~((1 << 20) - 1)

    def take(self, stream_key, nbytes):
        cls = self.size_class(nbytes)
        stack = self.blocks.get((stream_key, cls))
        if stack:
            self.cached -= cls
            ptr, token = stack.pop()
            if token is not None:
                self._wait(token)   # copies that were still queued from / into the block when it came back
            return ptr, cls
        try:
            return self._alloc(cls), cls
        except RuntimeError:
            self.trim()          # out of memory with blocks cached: give them back and retry once
            return self._alloc(cls), cls

    def give(self, stream_key, ptr, cls, in_flight=False):
        """in_flight: asynchronous copies may still be queued on the block (pools with a fence only)."""
        if self.cached + cls <= self.limit:
            token = self._fence(stream_key) if (in_flight and self._fence is not None) else None
            self.blocks.setdefault((stream_key, cls), []).append((ptr, token))
            self.cached += cls
        else:
            self._free(ptr)     # hipHostFree / hipFree wait for the device themselves

    def trim(self):
        for stack in self.blocks.values():
            for ptr, token in stack:
                if token is not None:
                    self._wait(token)
                self._free(ptr)
        self.blocks, self.cached = {}, 0


class HipManager:
    """Lazy singleton: `.lib`, `.device`, `.queue`, `.k`, `.device_name`."""

    max_register_count = 512  # tape format limit (reference nodes/__init__.py:6)

    def __init__(self):
        self._lib = None
        self._queue = None
        self._epoch = None
        self._free_events = []
        self.device = None
        self.k = _Kernels(self)
        limit = int(float(os.environ.get("CODECAD_AMD_POOL_MB", "4096")) * (1 << 20))
        self.device_pool = _BlockPool(self._raw_malloc, self._raw_free, limit)
        self.pinned_pool = _BlockPool(self._raw_host_alloc, self._raw_host_free, min(limit, 1 << 30),
                                      fence=self._pinned_fence, wait=self._pinned_wait)

    # -- lifecycle -----------------------------------------------------------------------
    @property
    def lib(self):
        if self._lib is None:
            self._open()
        return self._lib

    def _open(self, device=None):
        lib = _lib.load()
        n = ctypes.c_int()
        rc = lib.hu_device_count(ctypes.byref(n))
        if rc != 0 or n.value < 1:
            msg = lib.hu_last_error()
            raise HipError("no HIP device available (%s); codecad_amd has no CPU fallback"
                           % (msg.decode() if msg else "device count 0"))
        if device is None:
            device = int(os.environ.get("CODECAD_AMD_DEVICE", os.environ.get("LOCAL_RANK", "0")))
            device %= n.value
        check(lib.hu_set_device(device), "hu_set_device")
        self._lib = lib
        self.device = device
        self.device_count = n.value
        self._epoch = self._new_event()
        check(lib.hu_event_record(self._epoch, None), "hu_event_record")

    # -- raw allocation (the pools call these) -----------------------------------------------
    def _raw_malloc(self, nbytes):
        p = ctypes.c_void_p()
        check(self.lib.hu_malloc(ctypes.byref(p), nbytes), "hu_malloc")
        return p.value

    def _raw_free(self, ptr):
        check(self.lib.hu_free(ptr), "hu_free")

    def _raw_host_alloc(self, nbytes):
        p = ctypes.c_void_p()
        check(self.lib.hu_host_alloc(ctypes.byref(p), nbytes), "hu_host_alloc")
        return p.value

    def _raw_host_free(self, ptr):
        check(self.lib.hu_host_free(ptr), "hu_host_free")

    def _pinned_fence(self, stream_key):
        ev = self._new_event()
        check(self.lib.hu_event_record(ev, stream_key), "hu_event_record")
        return ev

    def _pinned_wait(self, ev):
        check(self.lib.hu_event_synchronize(ev), "hu_event_synchronize")
        self._free_events.append(ev)

    def empty_cache(self):
        """Return every pooled block to the driver."""
        self.device_pool.trim()
        self.pinned_pool.trim()

    def use_device(self, device):
        """Select the GPU ordinal for this process (before any allocation)."""
        if self._lib is not None and device != self.device:
            self.synchronize()
            self.empty_cache()   # pooled blocks belong to the device we are leaving
            self._queue = None
            self._free_events = []
        self._open(device)

    @property
    def available(self):
        try:
            return self.lib is not None
        except RuntimeError:
            return False

    @property
    def context(self):
        return self

    @property
    def queue(self):
        if self._queue is None:
            self._queue = Stream(self)
        return self._queue

    @property
    def device_name(self):
        buf = ctypes.create_string_buffer(256)
        check(self.lib.hu_device_name(self.device, buf, 256), "hu_device_name")
        return buf.value.decode()

    def new_stream(self):
        return Stream(self)

    def wrap_stream(self, raw_handle):
        """Adopt an existing hipStream_t (e.g. torch.cuda.current_stream().cuda_stream)."""
        return Stream(self, handle=raw_handle, owned=False)

    def synchronize(self):
        check(self.lib.hu_synchronize(), "hu_synchronize")

    # -- event pool ----------------------------------------------------------------------
    def _new_event(self):
        if self._free_events:
            return self._free_events.pop()
        h = ctypes.c_void_p()
        check(self.lib.hu_event_create(ctypes.byref(h)), "hu_event_create")
        return h.value

    def _recycle_event(self, h):
        if h is not None and len(self._free_events) < 256:
            self._free_events.append(h)


instance = HipManager()
