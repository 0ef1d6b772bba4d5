"""Device buffers, tape handles and the two-deep job interleavers.

`Buffer` has the surface of the reference's `cl_util.Buffer` (reference
cl_util/cl_buffer.py:9-131): numpy dtype + shape, a host shadow array, `read`,
`enqueue_read`, `enqueue_write`, `map`, indexing on the shadow.  The shadow lives in pinned
host memory so enqueue_* are genuinely asynchronous on the buffer's stream.
"""
import contextlib
import ctypes
import os

import numpy

from .manager import instance as _instance
from .manager import check, Event


class mem_flags:
    """Accepted for source compatibility with pyopencl.mem_flags; HBM has no such modes."""
    READ_WRITE = 1
    WRITE_ONLY = 2
    READ_ONLY = 4
    ALLOC_HOST_PTR = 16
    COPY_HOST_PTR = 32


class map_flags:
    READ = 1
    WRITE = 2
    WRITE_INVALIDATE_REGION = 4


class _PinnedBlock:
    """Owns one pinned host allocation (from the manager's pool); given back when the last numpy view of
    it dies, so an array returned by Buffer.read() stays valid after the Buffer itself is gone."""

    def __init__(self, manager, nbytes, stream_key):
        self.manager = manager
        self.stream_key = stream_key
        self.in_flight = False   # an asynchronous copy from / into the block was enqueued (Buffer.enqueue_*)
        self.ptr, self.size_class = manager.pinned_pool.take(stream_key, nbytes)

    def view(self, dtype, nitems, shape, nbytes):
        raw = (ctypes.c_char * max(nbytes, 1)).from_address(self.ptr)
        raw._owner = self  # numpy keeps `raw` as the array's base, `raw` keeps us
        return numpy.frombuffer(raw, dtype=dtype, count=nitems).reshape(shape)

    def __del__(self):
        try:
            if self.ptr:
                # the host is not stream-ordered: the next owner may only write the block after copies still
                # queued on it have run (the pool records a fence on the stream now and waits for it on reuse)
                self.manager.pinned_pool.give(self.stream_key, self.ptr, self.size_class, self.in_flight)
                self.ptr = None
        except Exception:
            pass


class Buffer:
    @staticmethod
    def dual_dtype(scalar):
        return numpy.dtype([(n, scalar) for n in "xy"])

    @staticmethod
    def quad_dtype(scalar):
        return numpy.dtype([(n, scalar) for n in "xyzw"])

    def __init__(self, dtype, shape, mem_flags=None, queue=None):
        m = _instance
        self.manager = m
        self.queue = queue if queue is not None else m.queue
        self.dtype = numpy.dtype(dtype)
        try:
            self.shape = tuple(int(s) for s in shape)
        except TypeError:
            self.shape = (int(shape),)
        self.nitems = 1
        for s in self.shape:
            self.nitems *= s
        self.size = self.nitems * self.dtype.itemsize  # bytes
        self.array = None
        self._pinned = None
        self._stream_key = self.queue.handle
        self.device_ptr, self._size_class = m.device_pool.take(self._stream_key, self.size)

    # -- host shadow ---------------------------------------------------------------------
    def create_host_side_array(self):
        """Allocate the pinned shadow array `self.array` (uninitialised)."""
        self._pinned = _PinnedBlock(self.manager, self.size, self._stream_key)
        self.array = self._pinned.view(self.dtype, self.nitems, self.shape, self.size)

    def _host(self, array):
        if array is not None:
            return array
        if self.array is None:
            self.create_host_side_array()
        return self.array

    def _wait_all(self, wait_for):
        for ev in (wait_for or ()):
            if getattr(ev, "stream", None) is not self.queue and hasattr(ev, "_stop"):
                check(self.manager.lib.hu_stream_wait_event(self.queue.handle, ev._stop), "hu_stream_wait_event")

    # -- transfers -----------------------------------------------------------------------
    def enqueue_read(self, out=None, wait_for=None):
        """Device -> host (self.array or `out`), asynchronous; returns an Event."""
        host = self._host(out)
        if host.nbytes < self.size:
            raise RuntimeError("Not enough space to store contents of the buffer")
        self._wait_all(wait_for)
        ev = Event(self.manager, self.queue)
        if host is self.array and self._pinned is not None:
            self._pinned.in_flight = True
        check(self.manager.lib.hu_memcpy_d2h(host.ctypes.data, self.device_ptr, self.size, self.queue.handle), "hu_memcpy_d2h")
        return ev._done()

    def read(self, out=None, wait_for=None):
        """Blocking device -> host; returns the array."""
        host = self._host(out)
        self.enqueue_read(out=host, wait_for=wait_for).wait()
        return host

    def enqueue_write(self, a=None, wait_for=None):
        """Host (self.array or `a`) -> device, asynchronous; returns an Event."""
        host = self._host(a)
        from_shadow = host is self.array and self._pinned is not None
        host = numpy.ascontiguousarray(host)
        if host.nbytes > self.size:
            raise RuntimeError("Not enough space to store contents in the buffer")
        self._wait_all(wait_for)
        ev = Event(self.manager, self.queue)
        self._keepalive = host  # the copy is asynchronous: keep the source alive
        if from_shadow:
            self._pinned.in_flight = True
        check(self.manager.lib.hu_memcpy_h2d(self.device_ptr, host.ctypes.data, host.nbytes, self.queue.handle), "hu_memcpy_h2d")
        if not from_shadow:
            self.queue.synchronize()  # any other source (pageable, or the caller's to reuse): consumed before we return
        return ev._done()

    def enqueue_zero_fill_compatible(self, wait_for=None):
        return self.enqueue_fill(0, wait_for=wait_for)

    def enqueue_fill(self, byte_value=0, wait_for=None):
        """hipMemsetAsync on the buffer's stream (the reference writes zeros from the host)."""
        self._wait_all(wait_for)
        ev = Event(self.manager, self.queue)
        check(self.manager.lib.hu_memset(self.device_ptr, byte_value, self.size, self.queue.handle), "hu_memset")
        return ev._done()

    @contextlib.contextmanager
    def map(self, flags, offset=None, shape=None, wait_for=None):
        """Context manager giving a host view; written back on exit when mapped for writing."""
        if offset not in (None, 0) or (shape is not None and tuple(numpy.atleast_1d(shape)) != self.shape):
            raise NotImplementedError("partial maps are not supported")
        host = self._host(None)
        for ev in (wait_for or ()):
            ev.wait()
        if flags & map_flags.READ or flags & map_flags.WRITE:
            self.read()
        yield host
        if flags & (map_flags.WRITE | map_flags.WRITE_INVALIDATE_REGION):
            self.enqueue_write().wait()

    # -- container protocol ----------------------------------------------------------------
    def __getitem__(self, key):
        return self.array[key]

    def __setitem__(self, key, value):
        self.array[key] = value

    def __len__(self):
        return self.nitems

    def release(self):
        """Give the device memory back now (idempotent): to the manager's pool, for the next buffer used on
        the same stream.  The pinned shadow follows when the last array that views it is gone."""
        if self.device_ptr:
            self.manager.device_pool.give(self._stream_key, self.device_ptr, self._size_class)
            self.device_ptr = None
        self._pinned = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class BufferList:
    """Release many buffers together (reference cl_buffer.py:134-158)."""

    def __init__(self, buffers=()):
        self.buffers = list(buffers)

    def add(self, buff):
        self.buffers.append(buff)

    def release(self):
        try:
            for b in self.buffers:
                b.release()
        finally:
            self.buffers = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.release()


def cache_dir():
    """Directory of the on-disk cache of per-tape code objects (hu_tape_specialize_cached), or None:
    CODECAD_AMD_CACHE=<dir> selects it, "" or "0" disables it; default $XDG_CACHE_HOME/codecad_amd or
    ~/.cache/codecad_amd.  Created here if possible (the library only creates the last level)."""
    v = os.environ.get("CODECAD_AMD_CACHE")
    if v is not None:
        if v in ("", "0"):
            return None
        path = v
    else:
        base = os.environ.get("XDG_CACHE_HOME") or os.path.join(os.path.expanduser("~"), ".cache")
        if not os.path.isabs(base):
            return None     # no home directory to put it in
        path = os.path.join(base, "codecad_amd")
    try:
        os.makedirs(path, exist_ok=True)
    except OSError:
        pass    # best effort: the library ignores a directory it cannot use
    return path


class _BackgroundCompiler:
    """Worker threads that turn tapes into per-tape code objects with hipRTC while the interpreter serves their launches.
    The builds themselves run in PROCESSES of their own (_compile_server.py, one per worker, started at its first request,
    host only): the HIP runtime and hipRTC share a lock, so a build on a thread of this process would stall a module load
    or a kernel's first launch for its whole duration.  A finished build sits in the on-disk cache -- or, when that is
    switched off, in a private directory of this process -- where the tape's next launch finds it in milliseconds.  A
    server's first build also pays hipRTC's one-off start (~1.5 s in a fresh process), so no launch ever waits for that.
    Several workers (CODECAD_AMD_RTC_SERVERS, default: up to eight, half the cores) build a tape's KERNELS side by side,
    one image per kernel (round 3: all ten kernels in one build; round 4 first its four families): the kernel a launch
    is waiting for is ready after its own compilation, the whole tape after its slowest kernel, and the servers -- which
    run the hipRTC of the ROCm installation, with its clang next to it -- skip the headers through a precompiled one
    (hu_spec_pch_prepare).  CODECAD_AMD_RTC_SERVER=0 (or a server that cannot be started) builds on the worker thread
    instead."""

    REPLY_TIMEOUT = 900.0      # seconds a build may take before its server is given up (and replaced at the next job)

    def __init__(self, servers=None):
        if servers is None:
            # (under torchrun the ranks of a node share its cores: LOCAL_WORLD_SIZE)
            ranks = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1))
            servers = int(os.environ.get("CODECAD_AMD_RTC_SERVERS", "0") or 0) or min(8, max(1, (os.cpu_count() or 2) // (2 * ranks)))
        self.queue, self.threads, self.private_dir = None, [], None
        self.slots = [{"server": None} for _ in range(max(1, int(servers)))]

    @property
    def server(self):
        """the first worker's server: None (not started yet), a live process, or False (cannot be used: builds run on the thread)"""
        return self.slots[0]["server"]

    def remote(self):
        """may builds run in server processes (CODECAD_AMD_RTC_SERVER, and no worker has given up on its server)?"""
        return os.environ.get("CODECAD_AMD_RTC_SERVER", "1") != "0" and all(slot["server"] is not False for slot in self.slots)

    def directory(self):
        d = cache_dir()
        if d:
            return d
        if self.private_dir is None:
            import atexit
            import shutil
            import tempfile
            self.private_dir = tempfile.mkdtemp(prefix="codecad_amd_jit_")
            atexit.register(shutil.rmtree, self.private_dir, True)
        return self.private_dir

    def submit(self, lib, host_tape, include_dir, groups):
        """-> a job: {"done": threading.Event, "error": None or str, "directory": where the image is, "groups": the
        kernel families it builds (hu_spec_group bits)}"""
        import queue
        import threading
        if not self.threads:
            self.queue = queue.Queue()
            for k, slot in enumerate(self.slots):
                th = threading.Thread(target=self._run, args=(slot,), name="codecad_amd-hiprtc-%d" % k, daemon=True)
                th.start()
                self.threads.append(th)
        job = {"done": threading.Event(), "error": None, "directory": self.directory(), "lib": lib, "groups": int(groups),
               "tape": numpy.array(host_tape, dtype=numpy.float32, copy=True), "include": include_dir}
        self.queue.put(job)
        return job

    def _start_server(self, lib):
        import atexit
        import subprocess
        import sys
        if os.environ.get("CODECAD_AMD_RTC_SERVER", "1") == "0":
            return None
        try:
            path = getattr(lib, "_name", None) or ""
            script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_compile_server.py")
            server = subprocess.Popen([sys.executable, script, path], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                                      stderr=subprocess.DEVNULL, text=True, bufsize=1)
        except OSError:
            return None

        def stop(p=server):
            try:
                p.stdin.close()
                p.terminate()
            except Exception:
                pass
        atexit.register(stop)
        return server

    @staticmethod
    def _stop(server):
        try:
            server.stdin.close()
        except Exception:
            pass
        try:
            server.kill()
            server.wait(10)
        except Exception:
            pass

    def _build_remote(self, job, slot):
        """-> None (done) / an error string / False (no server for this job: build here)"""
        import base64
        import json
        import select
        if slot["server"] is None:
            slot["server"] = self._start_server(job["lib"]) or False
        server = slot["server"]
        if not server:
            return False
        try:
            req = {"tape": base64.b64encode(job["tape"].tobytes()).decode(), "include": job["include"], "dir": job["directory"],
                   "groups": job["groups"]}
            server.stdin.write(json.dumps(req) + "\n")
            server.stdin.flush()
            ready, _, _ = select.select([server.stdout], [], [], self.REPLY_TIMEOUT)
            if not ready:
                raise OSError("the compile server did not answer in %d s" % self.REPLY_TIMEOUT)
            line = server.stdout.readline()
            if not line:
                raise OSError("the compile server ended")
            reply = json.loads(line)
            return None if reply.get("rc") == 0 else str(reply.get("error") or "hipRTC failed (%s)" % reply.get("rc"))
        except (OSError, ValueError):
            # gone, hung or talking nonsense: this job is built here; the server is put away and the NEXT job starts a new one
            # (a permanent fallback to this thread would bring back the stalls the server exists to avoid)
            self._stop(server)
            slot["server"] = None if slot.get("restarts", 0) < 3 else False
            slot["restarts"] = slot.get("restarts", 0) + 1
            return False

    def _run(self, slot):
        while True:
            job = self.queue.get()
            try:
                result = self._build_remote(job, slot)
                if result is False:
                    t = job["tape"]
                    size, hit = ctypes.c_size_t(0), ctypes.c_int(0)
                    rc = job["lib"].hu_tape_compile_groups(t.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), t.size, job["include"].encode(),
                                                           job["directory"].encode(), job["groups"], ctypes.byref(size), ctypes.byref(hit))
                    if rc != 0:
                        msg = job["lib"].hu_last_error()
                        job["error"] = msg.decode() if msg else "hipRTC failed (%d)" % rc
                else:
                    job["error"] = result
            except Exception as e:      # (never let the worker die: the tape just stays interpreted)
                job["error"] = "%s: %s" % (type(e).__name__, e)
            job["done"].set()


_background = _BackgroundCompiler()
from ._lib import SPEC_ALL, SPEC_KERNELS  # noqa: E402


def _kernels_of(groups, first=0):
    """The single kernels (hu_spec_group bits) of the set `groups` in the order they are built: the kernels of the set `first`
    -- the family in use -- over whole bricks, the mask kernel, that family's kernels for boxes that end anywhere; then the
    other families' the same way; the run-form kernels (grids that are mostly padding: 2D) last."""
    def rank(i):
        bit, mine = 1 << i, bool((1 << i) & int(first))
        if i >= 15:
            return 5                                   # k_grid_eval_runs / k_grid_eval_blocks_runs
        if i == 10:
            return 1                                   # k_box_masks
        if i >= 11:
            return 2 if mine else 4                    # ..._ragged
        return 0 if mine else 3
    bits = [i for i in range(SPEC_KERNELS) if (1 << i) & int(groups)]
    return [1 << i for i in sorted(bits, key=lambda i: (rank(i), i))]


class Tape:
    """A decoded instruction tape resident in HBM (replaces the reference's program buffer,
    nodes/program.py:79-84).  Accepted as the `scene` argument of every kernel."""

    def __init__(self, tape, policy=None):
        """policy: "auto" / "0" / "1" for this tape, overriding CODECAD_AMD_SPECIALIZE (see below)."""
        m = _instance
        self.manager = m
        t = numpy.ascontiguousarray(tape, dtype=numpy.float32)
        self.host_tape = t
        h = ctypes.c_void_p()
        check(m.lib.hu_tape_create(t.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), t.size, ctypes.byref(h)), "hu_tape_create")
        self.device_ptr = h.value
        self.device = m.device
        n, r, f = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        check(m.lib.hu_tape_info(h, ctypes.byref(n), ctypes.byref(r), ctypes.byref(f)), "hu_tape_info")
        self.n_instructions, self.n_registers, self.flags = n.value, r.value, f.value
        self.specialized = False
        # CODECAD_AMD_SPECIALIZE: "1" = compile per-tape kernels at upload (and wait for them), "0" = never on its
        # own, otherwise (default "auto"): the interpreter serves every launch at once, and as soon as it has done a
        # little work on this tape -- a tape evaluated once on a small grid never costs a compilation -- hipRTC
        # builds the per-tape kernels IN THE BACKGROUND (_BackgroundCompiler); the first launch after the build has
        # finished loads them (milliseconds) and runs 3-5x faster from then on.  No launch ever waits for the
        # compiler.  A program found in the on-disk cache (cache_dir) is taken at upload.  Same bytes either way.
        self._policy = policy if policy is not None else os.environ.get("CODECAD_AMD_SPECIALIZE", "auto")
        self._work = 0.0
        self._jobs = []          # background builds in flight, in the order they were asked for
        self.groups = 0          # the kernels whose per-tape code is loaded (hu_spec_group bits)
        self.from_cache = False
        if self._policy == "1":
            self.specialize()
        elif self._policy == "auto":
            # a program compiled before costs milliseconds: take it now (an image of all kernels, or one per kernel --
            # what the background builds leave behind)
            if cache_dir():
                self._specialize(only_if_cached=True)

    # The interpreter retires ~2.5e12 (tape instruction x sample) per second whatever the tape (measured on MI355X:
    # sponge(4): 85 x 29e9; planetary: 467 x 6.2e9).  A background build starts once the interpreter has spent
    # _START_SECONDS on the tape: enough that the tape is clearly not a one-off on a small grid, little against the
    # seconds of interpretation that the build then saves.
    _INTERPRETER_RATE = 2.5e12
    _START_SECONDS = 0.0005

    def note_samples(self, n, group=1):
        """Called by the launch wrappers with the number of samples about to be evaluated with this tape and the kernel
        family (hu_spec_group bit) that is about to run."""
        if self._policy != "auto" or self.groups == SPEC_ALL:
            return
        if self._jobs:
            if any(job["done"].is_set() for job in self._jobs):
                self._take_background_builds()
            return
        if self.groups:
            return      # (what was asked for is loaded; the other families follow when they are used)
        self._work += float(n) * self.n_instructions
        if self._work / self._INTERPRETER_RATE >= self._START_SECONDS:
            # one build per KERNEL, those of the family in use first; the workers build them side by side (their images are
            # also what a later process finds in the on-disk cache at upload)
            from . import builder
            self._jobs = [_background.submit(self.manager.lib, self.host_tape, builder.CSRC, k) for k in _kernels_of(SPEC_ALL, group)]

    def _take_background_builds(self):
        for job in [j for j in self._jobs if j["done"].is_set()]:      # (in any order: the families are built side by side)
            if job not in self._jobs:
                return
            self._jobs.remove(job)
            if job["error"] is not None:
                self._policy = "0"      # hipRTC cannot build this tape: stay with the interpreter
                self.build_error = job["error"]
                self._jobs = []
                return
            self._specialize(only_if_cached=True, directory=job["directory"], groups=job["groups"])
            if self.groups & job["groups"] != job["groups"]:
                self._policy = "0"      # (the image vanished or does not load: do not try again and again)
                self._jobs = []
                return

    def wait_specialized(self, timeout=None):
        """Wait for the background builds in flight (if any) and switch to them; returns self.specialized.  Launches never
        need this -- they use whatever is ready --; measurements and tests do."""
        for job in list(self._jobs):
            if not job["done"].wait(timeout):
                break
        self._take_background_builds()
        return self.specialized

    def specialize(self, groups=SPEC_ALL):
        """Compile straight-line kernels for this tape with hipRTC (seconds, once) and wait for them; afterwards
        every launch with this tape uses them.  Same results as the interpreter.  Raises
        RuntimeError (with the compiler log) if hipRTC cannot build it.  `groups`: the kernels to build
        (hu_spec_group bits: SPEC_DENSE | SPEC_BLOCKS | SPEC_CLASSIFY | SPEC_RENDER; default all of them).
        The kernels are built side by side in the compile servers, one image per kernel: the whole tape is there after
        its slowest kernel (measured on the MI355X box: sponge(4) 0.53 s, planetary 3.6 s; as one image in this process:
        1.42 s, 10.9 s).  CODECAD_AMD_SPECIALIZE_POOL=0 -- the default under torch.distributed with more than one rank,
        where every rank would start its own servers for the same tape -- builds the set in this process, as one image."""
        self._jobs = []
        groups = int(groups)
        directory = None
        pool = os.environ.get("CODECAD_AMD_SPECIALIZE_POOL") or ("1" if int(os.environ.get("WORLD_SIZE", "1") or 1) <= 1 else "0")
        if pool == "1" and _background.remote():
            from . import builder
            # what the cache holds already is taken first; the rest goes to the servers, one kernel per job
            self._specialize(only_if_cached=True, groups=groups)
            jobs = [_background.submit(self.manager.lib, self.host_tape, builder.CSRC, k) for k in _kernels_of(groups & ~self.groups)]
            for job in jobs:
                job["done"].wait()
            directory = _background.directory()     # (a kernel whose job failed is built -- and its error reported -- below)
            self._specialize(only_if_cached=False, directory=directory, groups=groups)
            self.from_cache = self.from_cache and not jobs      # (from the cache = nothing had to be compiled)
            return self
        self._specialize(only_if_cached=False, directory=directory, groups=groups)
        return self

    def _specialize(self, only_if_cached, directory=None, groups=SPEC_ALL):
        if self.groups & groups == groups:
            return
        from . import builder
        if directory is None:
            directory = cache_dir()
        hit, flag = ctypes.c_int(0), ctypes.c_int(0)
        check(self.manager.lib.hu_tape_specialize_groups(self.device_ptr, builder.CSRC.encode(),
                                                         directory.encode() if directory else None,
                                                         1 if only_if_cached else 0, int(groups), ctypes.byref(hit)), "hu_tape_specialize_groups")
        check(self.manager.lib.hu_tape_specialized(self.device_ptr, ctypes.byref(flag)), "hu_tape_specialized")
        self.groups = int(flag.value)
        self.specialized = self.groups != 0     # some kernel runs per-tape code (all of them: groups == SPEC_ALL)
        self.from_cache = bool(hit.value)

    @property
    def alive(self):
        return self.device_ptr is not None

    def release(self):
        if self.device_ptr:
            check(self.manager.lib.hu_tape_destroy(self.device_ptr), "hu_tape_destroy")
            self.device_ptr = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


# ---------------------------------------------------------------------------------------
# Two-deep software pipelines over events.  Semantics pinned by the reference's
# tests/test_clutil.py:188-248 (restated in tests/test_hip_util_host.py): children are only
# started after their parent finished, at most two jobs are in flight, LIFO job stack.
# The level-synchronous drivers in this package do not need them (one launch per level),
# they exist for callers written against the reference's per-block style.
# ---------------------------------------------------------------------------------------
def interleave(initial_jobs, helper1, helper2):
    """Alternate `enqueue(*job)` / `process_result(event)` between two helpers
    (reference cl_buffer.py:161-198)."""
    pending = list(initial_jobs)
    if not pending:
        raise AssertionError("There must be at least one job to start")

    class _Slot:
        def __init__(self, helper):
            self.helper, self.event = helper, None

        def start(self, job):
            self.event = self.helper.enqueue(*job)

        def finish(self):
            return self.helper.process_result(self.event)

    busy, idle = _Slot(helper1), _Slot(helper2)
    busy.start(pending.pop())
    while True:
        had_more = bool(pending)
        if had_more:
            idle.start(pending.pop())
        pending.extend(busy.finish())
        if not had_more:
            if not pending:
                return
            idle.start(pending.pop())
        busy, idle = idle, busy


def interleave2(job_func, initial_jobs):
    """Run generator jobs two at a time, switching whenever one yields an event
    (anything with .wait()); a job's return value is an iterable of follow-up job specs
    (reference cl_util/__init__.py:8-67)."""

    class _Ready:
        @staticmethod
        def wait():
            pass

    pending = list(initial_jobs)
    current = other = None  # each: [generator, event]
    while True:
        if current is None:
            if pending:
                current = [job_func(pending.pop()), _Ready]
            elif other is None:
                return
            else:
                current, other = other, None
                continue
        current[1].wait()
        try:
            current[1] = current[0].send(None)
        except StopIteration as stop:
            if stop.value is not None:
                pending.extend(stop.value)
            current = None
        else:
            current, other = other, current
