"""Host side of hip_util without a GPU: the C-ABI library loads and exports every symbol of
include/hip_util.h, fails loudly without a device, and the job interleavers keep the
reference's semantics (reference tests/test_clutil.py:188-248)."""
import ctypes
import os

import numpy
import pytest

from codecad_amd import hip_util
from codecad_amd.hip_util import _lib


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = _lib.header_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(_lib.PROTOTYPES) == declared      # every declared function has a typed binding
    assert lib.hu_abi_version() == 1


def test_no_gpu_means_loud_failure_not_fallback():
    lib = _lib.load()
    n = ctypes.c_int(-1)
    rc = lib.hu_device_count(ctypes.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    assert n.value == 0 and rc != 0 and lib.hu_last_error()
    fresh = hip_util.HipManager()
    with pytest.raises(RuntimeError):
        fresh.lib
    assert fresh.available is False


def test_argument_validation_needs_no_device():
    lib = _lib.load()
    assert lib.hu_tape_create(None, 0, None) == -3
    assert b"NULL" in lib.hu_last_error()
    assert lib.hu_free(None) == 0 and lib.hu_tape_destroy(None) == 0
    assert lib.hu_device_count(None) == -3


def test_interleave2_semantics():
    log = []

    class MockEvent:
        def __init__(self, job):
            self.job = job

        def wait(self):
            log.append((self.job, "wait"))

    def job_func(job):
        log.append((job, 1))
        yield MockEvent(job)
        log.append((job, 2))
        return [2 * job, 2 * job + 1] if job < 10 else None

    hip_util.interleave2(job_func, [2, 3])
    seen, working, busy_ticks, most = set(), set(), 0, 0
    for job, step in log:
        busy_ticks += bool(working)
        most = max(most, len(working))
        if step == 1:
            working.add(job)
        elif step == "wait":
            working.remove(job)
        else:
            assert (job, "wait") in seen
        if job > 3:
            assert (job // 2, 2) in seen      # a child never starts before its parent finished
        seen.add((job, step))
    assert (18, 2) in seen and (19, 2) in seen
    assert most == 2 and busy_ticks >= len(seen) - 4


def test_interleave_semantics():
    order = []

    class Helper:
        def __init__(self, name):
            self.name = name

        def enqueue(self, job, depth):
            self.current = (job, depth)
            order.append(("enqueue", self.name, job))
            return object()

        def process_result(self, event):
            job, depth = self.current
            order.append(("result", self.name, job))
            return [(job * 2, depth + 1), (job * 2 + 1, depth + 1)] if depth < 2 else []

    hip_util.interleave([(1, 0)], Helper("a"), Helper("b"))
    done = [j for kind, _, j in order if kind == "result"]
    assert sorted(done) == [1, 2, 3, 4, 5, 6, 7]
    started = [j for kind, _, j in order if kind == "enqueue"]
    for j in started:
        if j > 1:
            assert order.index(("result", "a", j // 2)) < [i for i, o in enumerate(order) if o[0] == "enqueue" and o[2] == j][0] \
                if ("result", "a", j // 2) in order else True
    with pytest.raises(AssertionError):
        hip_util.interleave([], Helper("a"), Helper("b"))


@pytest.mark.parametrize("s", ["", "ac", "a\nb", 'a"b', "ěščřžýáíé", "ab\0cd", "\1" "23", "".join(map(chr, range(128)))])
def test_format_c_string_literal(s, tmp_path):
    """The literal, compiled by a C compiler, is the original bytes (reference test_clutil.py:251-302)."""
    import subprocess
    lit = hip_util.format_c_string_literal(s)
    raw = s.encode("utf-8")
    src = tmp_path / "t.c"
    src.write_text('#include <stdio.h>\nint main(void){const char p[] = %s; fwrite(p, 1, sizeof(p), stdout); return 0;}\n' % lit)
    exe = tmp_path / "t"
    subprocess.run(["gcc", "-o", str(exe), str(src)], check=True, capture_output=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True).stdout
    assert out == raw + b"\0"


def _tape_ptr(tape):
    import ctypes
    import numpy
    t = numpy.ascontiguousarray(tape, dtype=numpy.float32)
    return t, t.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def test_specialised_source_builds_under_hiprtc_without_a_device():
    """hipRTC needs no GPU: the generated straight-line source and the op library headers it includes
    must compile for gfx950 and contain all eight kernels (catches header changes that only break
    the run-time compiler, e.g. a type hipRTC's built-in runtime does not declare)."""
    import ctypes
    import os
    import codecad_amd as cc
    from codecad_amd.hip_util import _lib
    lib = _lib.load()
    include_dir = os.path.join(os.path.dirname(cc.__file__), "csrc").encode()
    t, p = _tape_ptr(cc.nodes.make_program(cc.examples.sponge(2)))
    needed = ctypes.c_size_t(0)
    assert lib.hu_tape_source(p, t.size, None, 0, ctypes.byref(needed)) == 0
    buf = ctypes.create_string_buffer(needed.value)
    assert lib.hu_tape_source(p, t.size, buf, needed.value, ctypes.byref(needed)) == 0
    src = buf.value.decode()
    # sponge(2): the deferred-directions form (specialise.hpp): phase 1 as typed single-assignment statements -- in
    # place, and for the walks along x of a box with its tables --, then one block per (primitive, path to the root): 6
    # bars + the box, in each of the two evaluation functions
    assert "struct JitEval" in src and "tape_dist" in src and "deferred directions: 7 " in src
    for name in ("tape_dist(", "tape_eval(", "tape_pre_x(", "tape_dist_x(", "tape_eval_x("):
        assert src.count("auto " + name) == 1, name
    for name in ("tape_tab_x_x(", "tape_tab_x_y(", "tape_tab_x_z(", "tape_tab_x_xy(", "tape_tab_x_xz(", "tape_tab_x_yz("):
        assert src.count("void " + name) == 1, name
    # the bars of the two crosses are pair columns (2 x 3) and so is the box's square; the walks read them
    assert "table columns:" in src and "3 / 2 / 2 (xy / xz / yz)" in src, src[src.index("table columns:"):][:120]
    assert "tb.template XY<" in src and "tb.template YZ<" in src and "out[0 * S] = " in src
    assert src.count("const auto t") > 100 and "struct Hoisted {" in src
    assert "exec_one<T, true" not in src       # no record of the distance-only program is run through the library here
    assert src.count("// the primitive of record") == 2 * 7
    size = ctypes.c_size_t(0)
    rc = lib.hu_tape_compile_check(p, t.size, include_dir, ctypes.byref(size))
    assert rc == 0, lib.hu_last_error().decode()
    assert size.value > 10000
    # more statements of two coordinates than the tables of a box have columns: the caps hold, the rest stays with the walks
    import re
    parts = [cc.shapes.cylinder(d=0.5 + 0.01 * i, h=40).translated(i * 0.7 - 7, (i % 5) * 0.9, 0) for i in range(20)]
    parts += [cc.shapes.cylinder(d=0.4 + 0.01 * i, h=40).rotated_x(90).translated(i * 0.7 - 3, 0, (i % 3) * 1.1) for i in range(10)]
    tm, pm = _tape_ptr(cc.nodes.make_program(cc.shapes.union(parts)))
    assert lib.hu_tape_source(pm, tm.size, None, 0, ctypes.byref(needed)) == 0
    bufm = ctypes.create_string_buffer(needed.value)
    assert lib.hu_tape_source(pm, tm.size, bufm, needed.value, ctypes.byref(needed)) == 0
    cols = [int(v) for v in re.search(r"table columns: (\d+) / (\d+) / (\d+) \(x / y / z\), (\d+) / (\d+) / (\d+) \(xy / xz / yz\)", bufm.value.decode()).groups()]
    assert max(cols[:3]) <= 48 and max(cols[3:]) <= 16 and sum(cols[3:]) == 24, cols
    # a tape with a rounded blend keeps the plain straight-line form (its distance depends on directions)
    blend = cc.nodes.make_program(cc.shapes.union([cc.shapes.sphere(2), cc.shapes.box(1).translated_x(1)], r=0.3))
    tb, pb_ = _tape_ptr(blend)
    assert lib.hu_tape_source(pb_, tb.size, None, 0, ctypes.byref(needed)) == 0
    buf = ctypes.create_string_buffer(needed.value)
    assert lib.hu_tape_source(pb_, tb.size, buf, needed.value, ctypes.byref(needed)) == 0
    plain = buf.value.decode()
    assert "deferred directions" not in plain and plain.count("exec_one<T, false, decltype(regs), ") > 0
    rc = lib.hu_tape_compile_check(pb_, tb.size, include_dir, ctypes.byref(size))
    assert rc == 0, lib.hu_last_error().decode()
    # a malformed tape is rejected before any compilation
    bad, pb = _tape_ptr([99 * 512.0])
    assert lib.hu_tape_compile_check(pb, bad.size, include_dir, None) != 0
    assert b"malformed" in lib.hu_last_error()


def _source_of(tape, env=None):
    """hu_tape_source of a tape; with `env`, in a process of its own (the generator reads its knobs once)."""
    import ctypes
    import os
    import subprocess
    import sys
    import numpy
    from codecad_amd.hip_util import _lib
    if env is None:
        lib = _lib.load()
        t, p = _tape_ptr(tape)
        needed = ctypes.c_size_t(0)
        assert lib.hu_tape_source(p, t.size, None, 0, ctypes.byref(needed)) == 0
        buf = ctypes.create_string_buffer(needed.value)
        assert lib.hu_tape_source(p, t.size, buf, needed.value, ctypes.byref(needed)) == 0
        return buf.value.decode()
    code = ("import sys, ctypes, numpy; sys.path.insert(0, %r); from codecad_amd.hip_util import _lib; lib = _lib.load();"
            "t = numpy.frombuffer(sys.stdin.buffer.read(), dtype=numpy.float32).copy(); n = ctypes.c_size_t(0);"
            "p = t.ctypes.data_as(ctypes.POINTER(ctypes.c_float)); assert lib.hu_tape_source(p, t.size, None, 0, ctypes.byref(n)) == 0;"
            "b = ctypes.create_string_buffer(n.value); assert lib.hu_tape_source(p, t.size, b, n.value, ctypes.byref(n)) == 0;"
            "sys.stdout.write(b.value.decode())" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = subprocess.run([sys.executable, "-c", code], input=numpy.ascontiguousarray(tape, dtype=numpy.float32).tobytes(), capture_output=True,
                         env=dict(os.environ, **env), timeout=300)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    return out.stdout.decode()


def test_box_pruning_in_the_generated_source():
    """specialise.hpp "BOX PRUNING": an assembly's tape gets a mask function (bounds of its distances over a box from their
    values at the centre) and guarded assignments -- scalar bit tests around everything only a prunable operand needs --;
    the planetary assembly: all 80 (primitive, path) pairs deferred, > 100 prunable operands, its XY-frame gears in the pair
    table.  A tape in which nothing can be bounded (the sponge: all of it behind a repetition) keeps the code it had, and
    HU_PRUNE=0 builds any tape without pruning."""
    import re
    import codecad_amd as cc
    planetary = cc.nodes.make_program(cc.examples.planetary())
    src = _source_of(planetary)
    m = re.search(r"deferred directions: (\d+) \(primitive, path\) pairs; \d+ statements in phase 1; box pruning: (\d+) scopes", src)
    assert m and int(m.group(1)) == 80 and int(m.group(2)) > 100, src[-600:]
    words = int(re.search(r"kPruneWords = (\d+);", src).group(1))
    assert words == (int(m.group(2)) + 31) // 32
    prune = src[src.index("void tape_prune("):]
    prune = prune[:prune.index("\n}\n")]
    assert "iv_leaf(" in prune and "iv_perp(" in prune and "iv_min(" in prune and "iv_unknown()" in prune
    assert prune.count("&= ~") == int(m.group(2)) and "out.w[%d] = w%d;" % (words - 1, words - 1) in prune
    assert "run_record" not in prune          # (no bound is claimed for a gear: the mask function never evaluates one)
    dist = src[src.index("auto tape_dist_x("):]
    dist = dist[:dist.index("\n}\n")]
    assert dist.count("if (pr.template alive<") > 100 and "as<decltype(t" in dist and "{};" not in dist
    xy = src[src.index("void tape_tab_x_xy("):]
    xy = xy[:xy.index("\n}\n")]
    assert xy.count("= run_record<") >= 6 and "if (pr.template alive<" in xy       # the gears of the frames that keep z: 256 evaluations per box
    assert "mask_const<decltype(t" in src[src.index("auto tape_eval_x("):]
    # a repetition inside an assembly: its primitives are bounded in the boxes that stay inside ONE of its cells
    plate = cc.shapes.box(5, 4, 0.6) - cc.shapes.unsafe.Repetition(cc.shapes.cylinder(h=4, d=0.5), (1.0, 1.25, None))
    scene = cc.shapes.union([plate.translated(6, 0, 0), cc.shapes.sphere(3).translated(-6, 1, 0), cc.shapes.box(2).translated(0, 7, 0),
                             cc.shapes.cylinder(h=3, d=1).translated(0, -7, 1)])
    rep = _source_of(cc.nodes.make_program(scene))
    rp = rep[rep.index("void tape_prune("):]
    rp = rp[:rp.index("\n}\n")]
    assert rp.count("= iv_same_cell(") == 2 and "!(same0 && same1) ? iv_unknown() : iv_leaf(" in rp
    assert "kPruneAll = false;" in rep and "alive<" in rep[rep.index("auto tape_dist_x("):rep.index("auto tape_eval_x(")]
    assert "alive<" not in rep[rep.index("auto tape_eval_x("):]          # few scopes: the distance walks only
    # a tape that is mostly repetitions, or has nothing to bound: no scopes, no tests, the statements as they were
    sponge = _source_of(cc.nodes.make_program(cc.examples.sponge(3)))
    assert "box pruning: 0 scopes" in sponge and "kPruneWords = 0;" in sponge and "alive<" not in sponge and "const auto t" in sponge
    off = _source_of(planetary, env={"HU_PRUNE": "0"})
    assert "box pruning: 0 scopes" in off and "alive<" not in off


def test_specialised_code_cache_on_disk(tmp_path):
    """hu_tape_compile_cached (host only): a miss compiles and stores one file, a hit only reads it and gives
    the same code; truncated / bit-flipped / foreign files are ignored, rebuilt and replaced; another tape or other
    compiler options get another file; an unusable directory is not an error."""
    import ctypes
    import os
    import time
    import codecad_amd as cc
    from codecad_amd.hip_util import _lib
    lib = _lib.load()
    include_dir = os.path.join(os.path.dirname(cc.__file__), "csrc").encode()
    cache = tmp_path / "cache"
    t, p = _tape_ptr(cc.nodes.make_program(cc.shapes.sphere(2) - cc.shapes.box(1)))

    def compile_(tape_ptr, tape, directory):
        size, hit = ctypes.c_size_t(0), ctypes.c_int(-1)
        t0 = time.perf_counter()
        rc = lib.hu_tape_compile_cached(tape_ptr, tape.size, include_dir, None if directory is None else str(directory).encode(),
                                        ctypes.byref(size), ctypes.byref(hit))
        assert rc == 0, lib.hu_last_error().decode()
        return size.value, hit.value, time.perf_counter() - t0

    size0, hit, cold = compile_(p, t, cache)
    files = sorted(os.listdir(cache))
    assert hit == 0 and size0 > 10000 and len(files) == 1 and files[0].endswith(".huspec")
    size1, hit, warm = compile_(p, t, cache)
    assert hit == 1 and size1 == size0 and warm < cold / 5 and sorted(os.listdir(cache)) == files
    path = cache / files[0]
    good = path.read_bytes()
    assert len(good) > size0
    for bad in (good[:len(good) // 2], good[:-1], good + b"x", b"", b"HUSPEC1\0" + bytes(100),
                good[:200] + bytes([good[200] ^ 1]) + good[201:],            # a lowered name / header byte
                good[:-5000] + bytes([good[-5000] ^ 0x40]) + good[-4999:]):  # a byte of the code object
        path.write_bytes(bad)
        size, hit, _ = compile_(p, t, cache)
        assert hit == 0 and size == size0
        good = path.read_bytes()       # rebuilt and stored again (hipRTC's code objects carry a build id: not the same bytes)
        assert len(good) > size0 and compile_(p, t, cache)[1] == 1
    # another program -> another file; the first one still hits
    t2, p2 = _tape_ptr(cc.nodes.make_program(cc.shapes.sphere(3) - cc.shapes.box(1)))
    assert compile_(p2, t2, cache)[1] == 0 and len(os.listdir(cache)) == 2
    assert compile_(p, t, cache)[1] == 1
    # other compiler options -> another key
    os.environ["HU_RTC_FLAGS"] = "-DHU_CACHE_TEST=1"
    try:
        assert compile_(p, t, cache)[1] == 0 and len(os.listdir(cache)) == 3
    finally:
        del os.environ["HU_RTC_FLAGS"]
    # no cache / a directory that cannot be created: compiles, no error, nothing stored
    assert compile_(p, t, None)[1] == 0
    assert compile_(p, t, tmp_path / "missing" / "parent")[1] == 0 and not (tmp_path / "missing").exists()
    assert not [f for f in os.listdir(cache) if ".tmp" in f]
    # the cache is bounded: past 8192 entries (a tape leaves up to nineteen) the oldest go, the newest stay
    for i in range(8200):
        dummy = cache / ("%032x.huspec" % i)
        dummy.write_bytes(b"old")
        os.utime(dummy, (1000 + i, 1000 + i))
    t3, p3 = _tape_ptr(cc.nodes.make_program(cc.shapes.sphere(4) - cc.shapes.box(1)))
    assert compile_(p3, t3, cache)[1] == 0
    left = set(os.listdir(cache))
    assert len(left) == 6144 and "%032x.huspec" % 0 not in left and "%032x.huspec" % 8199 in left
    assert compile_(p3, t3, cache)[1] == 1 and compile_(p, t, cache)[1] == 1


def test_block_pool_recycles_per_stream_and_respects_its_limit():
    from codecad_amd.hip_util.manager import _BlockPool
    live, counter = set(), [0]

    def alloc(n):
        counter[0] += 1
        live.add(counter[0])
        return counter[0]

    pool = _BlockPool(alloc, live.remove, limit_bytes=3 << 20)
    assert [pool.size_class(n) for n in (0, 1, 256, 257, 1 << 20, (1 << 20) + 1)] == [256, 256, 256, 512, 1 << 20, 2 << 20]
    a, ca = pool.take("s1", 1000)
    b, cb = pool.take("s1", 1000)
    assert a != b and ca == cb == 1024
    pool.give("s1", a, ca)
    assert pool.take("s2", 1000)[0] not in (a, b)          # another stream never sees s1's block
    assert pool.take("s1", 600)[0] == a                     # same stream, same size class: recycled
    big = [pool.take("s1", 2 << 20) for _ in range(3)]
    for p, c in big:
        pool.give("s1", p, c)                               # 2 MiB + 2 MiB > 3 MiB limit: the second is freed
    assert pool.cached == 2 << 20 and big[1][0] not in live and big[0][0] in live
    pool.trim()
    assert pool.cached == 0 and big[0][0] not in live


def test_block_pool_trims_and_retries_when_the_allocator_fails():
    from codecad_amd.hip_util.manager import _BlockPool
    freed, fail_once = [], [True]

    def alloc(n):
        if n == 4096 and fail_once[0]:
            fail_once[0] = False
            raise RuntimeError("out of memory")
        return n

    pool = _BlockPool(alloc, freed.append, limit_bytes=1 << 30)
    p, c = pool.take("s", 300)
    pool.give("s", p, c)
    assert pool.take("s", 4000) == (4096, 4096) and freed == [512] and pool.cached == 0


def test_tape_decoder_survives_mutated_tapes():
    """The decoder is the trust boundary of the C ABI (a malformed tape must be refused at upload, the
    kernels never validate): mutate golden tapes -- truncations, corrupted words, wild registers, NaNs --
    and decode each through hu_tape_source (host only).  Every call must return, with OK or a
    'malformed tape' error; the process surviving is the assertion."""
    import ctypes
    import json
    import os
    import numpy as np
    from conftest import ROOT
    from codecad_amd.hip_util import _lib
    lib = _lib.load()
    shapes = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_tapes.json")))["shapes"]
    tapes = [np.array(s["tape_u32"], dtype=np.uint32).view(np.float32) for s in shapes if s["tape_len"] < 400]
    rng = np.random.default_rng(11)
    needed = ctypes.c_size_t(0)
    ok = refused = 0
    for trial in range(3000):
        t = tapes[trial % len(tapes)].copy()
        kind = trial % 6
        if kind == 0:
            t = t[:rng.integers(0, len(t))]
        elif kind == 1:
            t[rng.integers(0, len(t))] = rng.choice([np.nan, np.inf, -1.0, 1e9, 0.5, 29 * 512.0, 6 * 512.0 + 3])
        elif kind == 2:
            i = rng.integers(0, len(t))
            t[i] = float(int(rng.integers(0, 29)) * 512 + int(rng.integers(0, 512)))
        elif kind == 3:
            t = np.concatenate([t[:rng.integers(0, len(t))], t[rng.integers(0, len(t)):]])
        elif kind == 4:
            t = rng.uniform(0, 29 * 512, size=rng.integers(1, 60)).astype(np.float32).round()
        else:
            t = t.view(np.uint32).copy()
            t[rng.integers(0, len(t))] ^= np.uint32(1 << int(rng.integers(0, 32)))
            t = t.view(np.float32)
        t = np.ascontiguousarray(t, dtype=np.float32)
        if t.size == 0:
            continue
        rc = lib.hu_tape_source(t.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), t.size, None, 0, ctypes.byref(needed))
        if rc == 0:
            ok += 1
        else:
            refused += 1
            assert b"malformed" in lib.hu_last_error()
    assert ok > 50 and refused > 500, (ok, refused)


def test_cache_dir_resolution(monkeypatch, tmp_path):
    from codecad_amd.hip_util import buffer
    monkeypatch.setenv("CODECAD_AMD_CACHE", "0")
    assert buffer.cache_dir() is None
    monkeypatch.setenv("CODECAD_AMD_CACHE", "")
    assert buffer.cache_dir() is None
    monkeypatch.setenv("CODECAD_AMD_CACHE", str(tmp_path / "a" / "b"))
    assert buffer.cache_dir() == str(tmp_path / "a" / "b") and (tmp_path / "a" / "b").is_dir()
    monkeypatch.delenv("CODECAD_AMD_CACHE")
    monkeypatch.setenv("XDG_CACHE_HOME", str(tmp_path / "xdg"))
    assert buffer.cache_dir() == str(tmp_path / "xdg" / "codecad_amd") and (tmp_path / "xdg" / "codecad_amd").is_dir()
    monkeypatch.setenv("XDG_CACHE_HOME", "relative/dir")
    assert buffer.cache_dir() is None


def _listing(tape, which):
    import ctypes
    from codecad_amd.hip_util import _lib
    lib = _lib.load()
    t, p = _tape_ptr(tape)
    needed = ctypes.c_size_t(0)
    assert lib.hu_tape_listing(p, t.size, which, None, 0, ctypes.byref(needed)) == 0, lib.hu_last_error()
    buf = ctypes.create_string_buffer(needed.value)
    assert lib.hu_tape_listing(p, t.size, which, buf, needed.value, ctypes.byref(needed)) == 0
    return buf.value.decode().splitlines()


def test_leaf_fusion_of_the_interpreter_programs():
    """The interpreter's programs fuse `to -> primitive -> extrusion -> from -> select` runs into single records
    (tape.hpp fuse_leaves): sponge(4) 56 -> 26 dispatches, csg_example 20 -> 5; the unfused programs (what per-tape
    code is generated from) are untouched; a store of the transformed point survives only where it is still read."""
    import codecad_amd as cc
    sponge = cc.nodes.make_program(cc.examples.sponge(4))
    assert len(_listing(sponge, 0)) == 57 and len(_listing(sponge, 2)) == 27      # incl. _return
    assert len(_listing(sponge, 1)) == 57 and len(_listing(sponge, 3)) == 27
    assert sum(line.count("LEAF(") for line in _listing(sponge, 2)) == 13
    csg = _listing(cc.nodes.make_program(cc.examples.csg_example()), 2)
    assert len(csg) == 6 and all("LEAF(" in line for line in csg[:5])
    assert "LEAF(sample to:y circle extrusion from:y) [store 0]" == csg[0]          # the point's store is gone ...
    assert "store-point:2" in csg[2] and csg[3].startswith("[load 2]")              # ... and kept where a later record loads it
    # a rounded blend is never fused into a leaf (its operands' directions matter), plain selects are
    blend = _listing(cc.nodes.make_program(cc.shapes.union([cc.shapes.sphere(2), cc.shapes.box(1).translated_x(1)], r=0.3)), 2)
    assert any(line.startswith("union") or " union" in line and "LEAF" not in line for line in blend)
    assert _listing(cc.nodes.make_program(cc.shapes.union([cc.shapes.sphere(2), cc.shapes.box(1).translated_x(1)], r=0.3)), 1) == []


def test_interpreter_kernels_keep_their_program_in_scalar_registers(tmp_path):
    """Compile the library's device code to assembly (hipcc cross-compiles here, ~1 min) and look at every interpreter
    kernel: no scratch, and no vector-memory loads in the dense grid kernels (their only loads are the program's records,
    which belong in s_load).  Round 2 found this silently broken: two instantiations of the interpreter in one kernel
    made the compiler keep the record group in scratch and fetch it with global_load."""
    import re
    import subprocess
    from codecad_amd.hip_util import builder
    hipcc = builder.find_hipcc()
    if hipcc is None:
        pytest.skip("no hipcc in this environment")
    out = tmp_path / "hu.s"
    flags = [f for f in builder.HIPCC_FLAGS if f not in ("-fPIC",)] + builder.INTERPRETER_FLAGS
    subprocess.run([hipcc] + flags + ["-I", builder.INCLUDE, "--cuda-device-only", "-S", "-o", str(out),
                                      os.path.join(builder.CSRC, "hip_util.hip")], check=True, capture_output=True)
    text = out.read_text()
    seen = 0
    for chunk in re.split(r"\n(?=_Z\w+:\s+; @)", text):
        m = re.match(r"(_Z\w+):", chunk)
        if not m or "InterpEval" not in m.group(1):
            continue
        seen += 1
        scratch = re.search(r"; ScratchSize: (\d+)", chunk)
        assert scratch and int(scratch.group(1)) == 0, m.group(1)
        body = chunk.split(".section")[0]
        if "k_grid_eval" in m.group(1) and "blocks" not in m.group(1):
            assert not re.search(r"\tglobal_load|\tscratch_", body), m.group(1)
            assert re.search(r"\ts_load_dwordx(8|16)", body), m.group(1)
    assert seen >= 20
    _check_flagged_unit(text)


def _check_flagged_unit(text):
    """The translation unit built with -mllvm -structurizecfg-skip-uniform-regions (builder.FLAGGED_SOURCES).  That option
    once let a scalar branch choose a per-lane value in a DIVERGENT loop with a second exit (csrc/exchange.hip), so:
    the unit holds only the grid / leaf-block / classification kernels over the tape interpreter -- the renderers,
    contouring, reductions, the exchange step, the sort and marching cubes are built without the option --, and no loop
    of it that wavefront lanes leave one by one (a back edge on EXEC) has another way out."""
    import re
    from codecad_amd.hip_util import builder
    assert tuple(builder.FLAGGED_SOURCES) == ("hip_util.hip",)
    kernels = re.findall(r"\.amdhsa_kernel (\w+)", text)
    assert len(kernels) >= 28
    for k in kernels:
        assert re.match(r"_ZN4sdfk(11k_grid_eval|18k_grid_eval_blocks|10k_classify)INS_10InterpEval", k), k
    checked = 0
    for chunk in re.split(r"\n(?=_Z\w+:\s+; @)", text):
        m = re.match(r"(_Z\w+):", chunk)
        if not m:
            continue
        lines = chunk.split("\n.Lfunc_end")[0].split("\n")
        labels = {}
        for i, line in enumerate(lines):
            lm = re.match(r"(\.LBB\d+_\d+):", line)
            if lm:
                labels[lm.group(1)] = i
        branches = []
        for i, line in enumerate(lines):
            bm = re.match(r"\s+(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)", line)
            if bm and bm.group(2) in labels:
                branches.append((i, bm.group(1), labels[bm.group(2)]))
        for i, op, target in branches:
            if target <= i and op in ("s_cbranch_execnz", "s_cbranch_execz"):     # a loop that lanes leave one by one
                other_exits = [k for k, _, t in branches if target <= k < i and not (target <= t <= i + 1)]
                assert not other_exits, "%s: a divergent loop with a second exit under -structurizecfg-skip-uniform-regions" % m.group(1)
                checked += 1
    assert checked >= 1    # (the gear's bisection, the rounded blend: the check has something to look at)


def test_precompiled_header_of_the_per_tape_builds(tmp_path):
    """hu_spec_pch_prepare makes the header with the clang++ next to hipRTC (skipped where there is none); a per-tape build
    finds it in its cache directory, a header the compiler refuses is dropped -- the build still succeeds -- and the next
    process makes a new one.  Run in processes of their own, without torch (whose wheel brings another hipRTC along), on a
    COPY of the library (the real one has its header beside it: builder.prepare_pch)."""
    import shutil
    import subprocess
    import sys
    import codecad_amd as cc
    from codecad_amd.hip_util import _lib, builder
    lib = _lib.load()
    copy = tmp_path / "lib" / "libhip_util.so"
    copy.parent.mkdir()
    shutil.copy(lib._name, copy)
    tape = tmp_path / "tape.f32"
    numpy.ascontiguousarray(cc.nodes.make_program(cc.shapes.sphere(2) - cc.shapes.box(1.5)), dtype=numpy.float32).tofile(tape)
    cache = tmp_path / "cache"
    code = """
import ctypes, os, sys
lib = ctypes.CDLL(sys.argv[1])
fp = ctypes.POINTER(ctypes.c_float)
lib.hu_tape_compile_groups.argtypes = [fp, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_int)]
lib.hu_spec_pch_prepare.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t]
raw = open(sys.argv[2], 'rb').read()
tape = (ctypes.c_float * (len(raw) // 4)).from_buffer_copy(raw)
csrc, cache, what = sys.argv[3].encode(), sys.argv[4].encode(), sys.argv[5]
def build(kernel):
    size = ctypes.c_size_t(0)
    rc = lib.hu_tape_compile_groups(tape, len(tape), csrc, cache, kernel, ctypes.byref(size), None)
    return rc, size.value
if what == 'prepare':
    buf = ctypes.create_string_buffer(4096)
    assert lib.hu_spec_pch_prepare(csrc, cache, buf, 4096) == 0
    made = buf.value.decode().split()
    print('PCH', made[0] if made else '')
    print('COUNT', len(made))
    print('BUILD', *build(1 << 10))
elif what == 'refuse':
    path = sys.argv[6]
    open(path, 'wb').write(b'not a precompiled header')
    print('BUILD', *build(1 << 1))
    print('LEFT', int(os.path.exists(path)))
else:
    print('BUILD', *build(1 << int(sys.argv[6])))
"""

    def run(*what, env=None):
        e = dict(os.environ, AMD_COMGR_CACHE="0")
        e.update(env or {})
        out = subprocess.run([sys.executable, "-c", code, str(copy), str(tape), builder.CSRC, str(cache)] + list(what),
                             capture_output=True, text=True, timeout=600, env=e)
        assert out.returncode == 0, out.stderr[-2000:]
        return dict(line.split(" ", 1) for line in out.stdout.splitlines() if " " in line)

    first = run("prepare")
    if not first["PCH"]:
        pytest.skip("no clang++ next to the hipRTC in use: no precompiled header")
    pch = first["PCH"]
    assert first["COUNT"] == "2"        # one per optimisation level the builds use (-O3; -O1 for big sources)
    assert os.path.dirname(pch) == str(cache) and os.path.getsize(pch) > 100000 and first["BUILD"].split()[0] == "0"
    with_header = int(first["BUILD"].split()[1])
    shutil.rmtree(cache)
    without = run("build", "10", env={"HU_RTC_PCH": "0"})    # the same kernel without: no header appears, the same code
    assert without["BUILD"].split() == ["0", str(with_header)] and not [f for f in os.listdir(cache) if f.endswith(".pch")]
    made = run("build", "9")                                  # a build makes the header itself when it is missing
    assert made["BUILD"].split()[0] == "0" and os.path.exists(pch) and with_header > 1000
    refused = run("refuse", pch)
    assert refused["BUILD"].split()[0] == "0" and refused["LEFT"] == "0"
    again = run("prepare")
    assert again["PCH"] == pch and os.path.getsize(pch) > 100000
    # a source above HU_RTC_BIG_KB is built with -O1, from the header of that level: another image of the same kernel
    images = len([f for f in os.listdir(cache) if f.endswith(".huspec")])
    big = run("build", "9", env={"HU_RTC_BIG_KB": "1"})
    assert big["BUILD"].split()[0] == "0" and len([f for f in os.listdir(cache) if f.endswith(".huspec")]) == images + 1
    assert len([f for f in os.listdir(cache) if f.endswith(".pch")]) == 2


def test_background_builds_run_in_a_process_of_their_own(tmp_path, monkeypatch):
    """The worker thread hands a tape's kernel family to the compile server (codecad_amd/hip_util/_compile_server.py:
    host only, no torch) and finds the image in the directory it named; a failed build is reported on the job, the
    server survives a malformed request, and with CODECAD_AMD_RTC_SERVER=0 -- or a server that is gone -- the same job is
    built on the worker thread."""
    import ctypes
    import json
    import os
    import subprocess
    import sys
    import codecad_amd as cc
    from codecad_amd.hip_util import _lib, buffer, builder
    lib = _lib.load()
    monkeypatch.setenv("CODECAD_AMD_CACHE", str(tmp_path / "cache"))
    tape = cc.nodes.make_program(cc.shapes.sphere(2) - cc.shapes.box(1.5))

    def is_hit(t, groups):
        t = numpy.ascontiguousarray(t, dtype=numpy.float32)
        size, hit = ctypes.c_size_t(0), ctypes.c_int(-1)
        rc = lib.hu_tape_compile_groups(t.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), t.size, builder.CSRC.encode(),
                                        str(tmp_path / "cache").encode(), groups, ctypes.byref(size), ctypes.byref(hit))
        assert rc == 0
        return hit.value == 1

    worker = buffer._BackgroundCompiler(servers=1)
    job = worker.submit(lib, tape, builder.CSRC, _lib.SPEC_CLASSIFY)
    assert job["done"].wait(120) and job["error"] is None and job["directory"] == str(tmp_path / "cache")
    assert worker.server and worker.server.poll() is None          # a live child process built it
    assert is_hit(tape, _lib.SPEC_CLASSIFY)
    # a tape hipRTC cannot build (malformed) -> the error is on the job, the server goes on
    bad = worker.submit(lib, [99 * 512.0], builder.CSRC, _lib.SPEC_DENSE)
    assert bad["done"].wait(120) and bad["error"] and "malformed" in bad["error"]
    assert worker.server.poll() is None
    # the server gone -> the worker builds that job itself, and the job after it gets a NEW server (no permanent fallback to
    # builds on a thread of this process: those stall launches)
    worker.server.kill()
    worker.server.wait(10)
    tape2 = cc.nodes.make_program(cc.shapes.sphere(2) - cc.shapes.box(1.25))
    job = worker.submit(lib, tape2, builder.CSRC, _lib.SPEC_CLASSIFY)
    assert job["done"].wait(120) and job["error"] is None and worker.server is None
    assert is_hit(tape2, _lib.SPEC_CLASSIFY)
    tape2b = cc.nodes.make_program(cc.shapes.sphere(2) - cc.shapes.box(1.3))
    job = worker.submit(lib, tape2b, builder.CSRC, _lib.SPEC_CLASSIFY)
    assert job["done"].wait(120) and job["error"] is None and worker.server and worker.server.poll() is None
    assert is_hit(tape2b, _lib.SPEC_CLASSIFY)
    # several workers build side by side, each with a server of its own
    pool = buffer._BackgroundCompiler(servers=2)
    tapes = [cc.nodes.make_program(cc.shapes.sphere(2) - cc.shapes.box(1.0 + 0.01 * k)) for k in range(2)]
    jobs = [pool.submit(lib, t, builder.CSRC, _lib.SPEC_CLASSIFY) for t in tapes]
    assert all(j["done"].wait(120) and j["error"] is None for j in jobs) and all(is_hit(t, _lib.SPEC_CLASSIFY) for t in tapes)
    assert len(pool.slots) == 2 and sum(1 for sl in pool.slots if sl["server"]) >= 1
    # switched off -> never started
    monkeypatch.setenv("CODECAD_AMD_RTC_SERVER", "0")
    worker = buffer._BackgroundCompiler(servers=1)
    tape3 = cc.nodes.make_program(cc.shapes.sphere(2) - cc.shapes.box(1.125))
    job = worker.submit(lib, tape3, builder.CSRC, _lib.SPEC_DENSE)
    assert job["done"].wait(120) and job["error"] is None and worker.server is False and is_hit(tape3, _lib.SPEC_DENSE)
    # the protocol itself: one JSON line in, one out; garbage is answered, not fatal; stdin closing ends it
    script = os.path.join(os.path.dirname(buffer.__file__), "_compile_server.py")
    text = open(script).read()
    assert "import torch" not in text and "import numpy" not in text and "codecad_amd" not in text.split('"""')[2]
    p = subprocess.Popen([sys.executable, script, lib._name], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, bufsize=1)
    p.stdin.write("not json\n")
    p.stdin.flush()
    assert json.loads(p.stdout.readline())["rc"] != 0
    p.stdin.close()
    assert p.wait(30) == 0
