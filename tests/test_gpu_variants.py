"""Every interpreter variant against the oracle: {1, 2} voxels per lane x {distance-only, full}.

The library picks a variant per kernel (csrc/hip_util.hip launch_shape / distance_only); the
environment switches read at first use force one, so each combination runs in its own
process over the kernel-level parity tests."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("voxels_per_lane", ["1", "2"])
@pytest.mark.parametrize("full_interpreter", ["0", "1"])
def test_forced_variant_matches_oracle(hip, voxels_per_lane, full_interpreter):
    env = dict(os.environ, HU_VOXELS_PER_LANE=voxels_per_lane, HU_FULL_INTERPRETER=full_interpreter)
    cmd = [sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
           os.path.join(ROOT, "tests", "test_gpu_parity.py"),
           os.path.join(ROOT, "tests", "test_gpu_drivers.py"), "-k",
           "grid_eval or subdivision_step or mass_properties_kernel or level_batched or leaf_block or degenerate"]
    proc = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-2000:]
    assert " passed" in proc.stdout


SPEC_SHAPES = ["sphere_plus_box", "csg_example", "sponge4", "gear", "mirror_3d", "kat_circle", "nonconvex_shell2",
               "torus", "extreme_twisted_revolve", "revolved_pentagon", "rotated_pattern_2d", "planetary"]


@pytest.mark.parametrize("name", SPEC_SHAPES + sorted(__import__("shapes_zoo").rounded_shapes))
def test_specialised_tape_is_bit_identical_to_the_interpreter(hip, name):
    """hu_tape_specialize (hipRTC straight-line kernels) vs the interpreter, byte for byte, through
    all four reference-shaped kernels and one level-batched launch."""
    import math
    import ctypes
    import numpy as np
    import shapes_zoo
    from conftest import load_golden_tapes
    from codecad_amd import hip_util, nodes
    from codecad_amd.hip_util import check
    if name in shapes_zoo.rounded_shapes:
        shape = shapes_zoo.rounded_shapes[name]
        tape_f, dim = nodes.make_program(shape), shape.dimension()
        bb = shape.bounding_box()
        a, b = np.array(bb.a, float), np.array(bb.b, float)
    else:
        ref = load_golden_tapes()[name]
        tape_f, dim = ref["tape"], ref["dimension"]
        a, b = np.array(ref["bbox_a"]), np.array(ref["bbox_b"])
    a, b = np.where(np.isfinite(a), a, -2.0), np.where(np.isfinite(b), b, 2.0)
    n = 21
    size = float(np.max(b - a)) * 1.2 + 1e-3
    step = np.float32(size / n)
    c4 = np.zeros(4, np.float32)
    c4[:3] = (a + b) / 2 - size / 2 + size / n / 2
    dims = (n, n, n) if dim == 3 else (n, n, 1)
    if dim == 2:
        c4[2] = 0
    thr = np.float32(float(step) * math.sqrt(dim) / 2)
    cells = dims[0] * dims[1] * dims[2]
    results = []
    for specialise in (False, True):
        tape = hip_util.Tape(tape_f)
        if specialise:
            tape.specialize()
            flag = ctypes.c_int()
            check(hip.lib.hu_tape_specialized(tape.device_ptr, ctypes.byref(flag)), "hu_tape_specialized")
            assert flag.value == hip_util.SPEC_ALL      # every kernel family runs per-tape code
        got = []
        g4 = hip_util.Buffer(hip_util.Buffer.quad_dtype(np.float32), dims)
        hip.k.grid_eval(dims, None, tape, c4, step, g4).wait()
        got.append(g4.read().view(np.uint32).copy())
        g1 = hip_util.Buffer(np.float32, dims)
        hip.k.grid_eval_pymcubes(dims, None, tape, c4, step, g1).wait()
        got.append(g1.read().view(np.uint32).copy())
        counter = hip_util.Buffer(np.uint32, 1)
        lst = hip_util.Buffer(hip_util.Buffer.quad_dtype(np.uint8), cells)
        counter.enqueue_fill(0)
        hip.k.subdivision_step(dims, None, tape, c4, step, thr, counter, lst).wait()
        k = int(counter.read()[0])
        got.append(np.array(sorted(map(tuple, lst.read().view(np.uint8).reshape(-1, 4)[:k].tolist())), dtype=np.uint8))
        sums = hip_util.Buffer(np.uint32, 10)
        sums.enqueue_fill(0)
        counter.enqueue_fill(0)
        hip.k.mass_properties(dims, None, tape, c4, step, thr, sums, counter, lst).wait()
        got.append(sums.read().copy())
        got.append(counter.read().copy())
        results.append(got)
    for x, y in zip(*results):
        assert np.array_equal(x, y)


@pytest.mark.gpu
def test_tiered_specialisation_policy(hip, monkeypatch):
    """Default policy: the interpreter serves every launch at once; once it has done a little work on a tape, hipRTC
    builds the per-tape kernels in the background, and the first launch after the build switches to them.  No launch
    waits for the compiler; the bytes are the same before and after."""
    import time
    import numpy as np
    from codecad_amd import hip_util, examples, nodes, grid_eval
    tape = nodes.make_program(examples.sponge(2))
    c = np.zeros(4, np.float32)
    c[:3] = -0.5
    out = hip_util.Buffer(grid_eval.FLOAT4, (24, 24, 24))

    t = hip_util.Tape(tape)
    assert t._policy == "auto" and not t.specialized and not t._jobs
    hip.k.grid_eval((24, 24, 24), None, t, c, np.float32(1 / 24), out).wait()
    assert not t.specialized and not t._jobs and t._work == 24 ** 3 * t.n_instructions      # too little work to bother
    interpreted = out.read().copy()
    t._START_SECONDS = 0.0                                                   # "enough work" from now on
    t0 = time.perf_counter()
    hip.k.grid_eval((24, 24, 24), None, t, c, np.float32(1 / 24), out).wait()
    assert time.perf_counter() - t0 < 0.5 and len(t._jobs) == hip_util.SPEC_KERNELS and not t.specialized   # interpreted, the builds (one per kernel) are under way
    assert np.array_equal(out.read().view(np.uint32), interpreted.view(np.uint32))
    assert t.wait_specialized(timeout=120) and t.specialized and t.groups == hip_util.SPEC_ALL and not t._jobs
    hip.k.grid_eval((24, 24, 24), None, t, c, np.float32(1 / 24), out).wait()
    assert np.array_equal(out.read().view(np.uint32), interpreted.view(np.uint32))
    # a second tape: its launches pick the finished build up by themselves
    u = hip_util.Tape(nodes.make_program(examples.sponge(1)))
    u._START_SECONDS = 0.0
    deadline = time.perf_counter() + 120
    first_groups = 0
    while u.groups != hip_util.SPEC_ALL and time.perf_counter() < deadline:
        hip.k.grid_eval((24, 24, 24), None, u, c, np.float32(1 / 24), out).wait()
        first_groups = first_groups or u.groups
        time.sleep(0.02)
    # (the kernels are built side by side, those of the family in use are asked for first: whichever lands first, all arrive)
    assert u.groups == hip_util.SPEC_ALL and first_groups != 0

    monkeypatch.setenv("CODECAD_AMD_SPECIALIZE", "0")
    never = hip_util.Tape(tape)
    never._START_SECONDS = 0.0
    hip.k.grid_eval((24, 24, 24), None, never, c, np.float32(1 / 24), out).wait()
    assert not never.specialized and not never._jobs
    out.release()


@pytest.mark.gpu
def test_specialised_code_cache(hip, monkeypatch, tmp_path):
    """CODECAD_AMD_CACHE: the first specialisation compiles and stores, the next tape with the same program
    loads the stored code object (no hipRTC), under the default policy already at upload; results are the
    same bytes as the interpreter's; a damaged file is rebuilt; policy "0" never looks at the cache."""
    import os
    import time
    import numpy as np
    from codecad_amd import hip_util, examples, nodes, grid_eval
    monkeypatch.setenv("CODECAD_AMD_CACHE", str(tmp_path / "cache"))
    monkeypatch.setenv("CODECAD_AMD_SPECIALIZE_POOL", "0")      # specialize() in this process: ONE image (the servers' form: below)
    tape = nodes.make_program(examples.sponge(2))
    c = np.zeros(4, np.float32)
    c[:3] = -0.5
    out = hip_util.Buffer(grid_eval.FLOAT4, (24, 24, 24))

    def run(t):
        hip.k.grid_eval((24, 24, 24), None, t, c, np.float32(1 / 24), out).wait()
        return out.read().view(np.uint32).copy()

    first = hip_util.Tape(tape)
    assert not first.specialized                    # nothing cached yet: interpreted
    interpreted = run(first)
    t0 = time.perf_counter()
    first.specialize()
    cold = time.perf_counter() - t0
    files = os.listdir(tmp_path / "cache")
    assert first.specialized and not first.from_cache and len(files) == 1
    assert np.array_equal(run(first), interpreted)

    t0 = time.perf_counter()
    second = hip_util.Tape(tape)                    # default policy: taken from the cache at upload
    warm = time.perf_counter() - t0
    assert second.specialized and second.from_cache and warm < cold / 5
    assert np.array_equal(run(second), interpreted)

    never = hip_util.Tape(tape, policy="0")
    assert not never.specialized and np.array_equal(run(never), interpreted)

    path = tmp_path / "cache" / files[0]
    blob = path.read_bytes()
    path.write_bytes(blob[:len(blob) // 2])
    third = hip_util.Tape(tape)
    assert not third.specialized                    # the damaged file is ignored: interpreted until compiled
    third.specialize()
    assert third.specialized and not third.from_cache and len(path.read_bytes()) == len(blob)
    assert np.array_equal(run(third), interpreted)
    assert hip_util.Tape(tape).from_cache

    # a file that passes the container's checks but holds no loadable code object (e.g. written by another
    # runtime): at upload it is skipped silently, an explicit specialize() rebuilds and replaces it
    blob = bytearray(path.read_bytes())
    elf_header_wrecked = bytearray(blob)
    start = bytes(blob).find(b"\x7fELF")
    assert start > 0
    elf_header_wrecked[start:start + 64] = bytes(64)
    h = 0xcbf29ce484222325
    for byte in elf_header_wrecked[:-8]:
        h = ((h ^ byte) * 0x100000001b3) & 0xffffffffffffffff
    elf_header_wrecked[-8:] = h.to_bytes(8, "little")
    path.write_bytes(bytes(elf_header_wrecked))
    quiet = hip_util.Tape(tape)
    assert not quiet.specialized and np.array_equal(run(quiet), interpreted)
    quiet.specialize()
    assert quiet.specialized and not quiet.from_cache and np.array_equal(run(quiet), interpreted)
    assert hip_util.Tape(tape).from_cache

    # specialize() through the compile servers (the default): one image per kernel, found again kernel by kernel
    monkeypatch.setenv("CODECAD_AMD_SPECIALIZE_POOL", "1")
    other = nodes.make_program(examples.sponge(1))
    before = len(os.listdir(tmp_path / "cache"))
    pooled = hip_util.Tape(other)
    reference = run(pooled)
    assert not pooled.specialized
    pooled.specialize()
    assert pooled.groups == hip_util.SPEC_ALL and not pooled.from_cache
    assert len([f for f in os.listdir(tmp_path / "cache") if f.endswith(".huspec")]) == before + hip_util.SPEC_KERNELS
    assert np.array_equal(run(pooled), reference)
    found = hip_util.Tape(other)
    assert found.groups == hip_util.SPEC_ALL and found.from_cache and np.array_equal(run(found), reference)
    part = hip_util.Tape(other, policy="0").specialize(hip_util.SPEC_RENDER)        # a part of them: nothing to compile
    assert part.groups == hip_util.SPEC_RENDER and part.from_cache
    out.release()


@pytest.mark.gpu
def test_a_reported_hip_error_does_not_resurface(hip):
    """A HIP failure handed back through the return code (here an impossible allocation, which the block pools
    answer by trimming and retrying) must not stay behind as the runtime's sticky last error: the next launch
    checks hipGetLastError() and would report it as its own."""
    import ctypes
    import numpy as np
    from codecad_amd import hip_util, examples, nodes, grid_eval
    p = ctypes.c_void_p()
    assert hip.lib.hu_malloc(ctypes.byref(p), 1 << 50) != 0
    assert b"hipMalloc" in hip.lib.hu_last_error()
    t = hip_util.Tape(nodes.make_program(examples.sponge(1)))
    out = hip_util.Buffer(grid_eval.FLOAT4, (8, 8, 8))
    c = np.zeros(4, np.float32)
    hip.k.grid_eval((8, 8, 8), None, t, c, np.float32(0.1), out).wait()      # raises if the old error resurfaces
    out.release()


@pytest.mark.gpu
@pytest.mark.parametrize("specialise", [False, True])
def test_union_distances_at_signed_zeros_and_nans(hip, specialise):
    """The distance of a union / intersection / subtraction is the hardware's v_min_f32 / v_max_f32, which the
    oracle restates: where the two operands are zeros of opposite sign (samples exactly on both surfaces) and
    where one of them is NaN (a NaN sample coordinate that only one operand reads), kernels and oracle must still
    agree bit for bit, in the full program (float4) and in the distance-only one (float)."""
    import numpy as np
    import oracle
    from conftest import same_bits
    from codecad_amd import hip_util, nodes, grid_eval
    from codecad_amd.shapes import half_space, sphere, box
    solid, plane = box(4), half_space().translated(0, 2, 0)     # on the face y = 2: distances +0 and -0
    far = sphere(1).translated(50, 0, 0)
    cases = {"union": solid + plane, "intersection": solid & plane, "subtraction": solid - plane,
             "subtraction_swapped": plane - solid, "nested": (plane + far) - (solid & box(6))}
    n = 9
    step = np.float32(0.5)
    for label, shape in cases.items():
        tape = nodes.make_program(shape)
        t = hip_util.Tape(tape, policy="0")
        if specialise:
            t.specialize()
        for corner in ([-2.0, -2.0, -2.0], [float("nan"), -2.0, -2.0], [-2.0, -2.0, float("nan")]):
            c4 = np.zeros(4, np.float32)
            c4[:3] = corner
            full = hip_util.Buffer(grid_eval.FLOAT4, (n, n, n))
            hip.k.grid_eval((n, n, n), None, t, c4, step, full).wait()
            got = full.read().view(np.float32).reshape(n, n, n, 4)
            want = oracle.grid_eval(tape, np.array(corner, np.float32), step, (n, n, n))
            assert same_bits(got, want), (label, corner, "float4")
            scalar = hip_util.Buffer(np.float32, (n, n, n))
            hip.k.grid_eval_pymcubes((n, n, n), None, t, c4, step, scalar).wait()
            want_s = oracle.grid_eval_pymcubes(tape, np.array(corner, np.float32), step, (n, n, n))
            assert same_bits(scalar.read().reshape(-1), np.asarray(want_s).reshape(-1)), (label, corner, "float")
            full.release()
            scalar.release()
        if label == "union":   # the case this test is about really occurs: +0 and -0 meet on the face
            w = oracle.grid_eval(tape, np.array([-2.0, -2.0, -2.0], np.float32), step, (n, n, n))[2:7, 8, 2:7, 3]
            a = oracle.grid_eval(nodes.make_program(solid), np.array([-2.0, -2.0, -2.0], np.float32), step, (n, n, n))[2:7, 8, 2:7, 3]
            b = oracle.grid_eval(nodes.make_program(plane), np.array([-2.0, -2.0, -2.0], np.float32), step, (n, n, n))[2:7, 8, 2:7, 3]
            assert np.all(a == 0) and not np.any(np.signbit(a)) and np.all(b == 0) and np.all(np.signbit(b))
            assert np.all(w == 0) and np.all(np.signbit(w))          # the minimum of +0 and -0 is -0


@pytest.mark.gpu
def test_external_tape_ending_in_load_return(hip):
    """hu_tape_create accepts tapes our scheduler never emits.  One that ends `..., _load r; _return` returns the
    LOADED value: the decoder must not fold that load into the return record (the per-tape code generator stops at
    the return record and would drop it) -- interpreter, per-tape code and oracle agree."""
    import numpy as np
    import oracle
    from conftest import same_bits
    from codecad_amd import hip_util, grid_eval
    from codecad_amd.nodes import node as nm
    code = {name: spec[2] for name, spec in nm.Node.node_types.items()}

    def word(op, reg=0):
        return float(code[op] * 512 + reg)

    # sphere(1) -> r1; box-ish rectangle -> last; then _load r1; _return  => the sphere, not the rectangle
    tape = np.array([word("initial_transformation_to"), 0, 0, 0, 1, 0, 0, 0, word("_store", 0),
                     word("sphere"), 1.0, word("_store", 1),
                     word("_load", 0), word("rectangle"), 0.5, 0.25,
                     word("_load", 1), word("_return")], dtype=np.float32)
    n = 8
    corner, step = np.array([-1.1, -0.9, -1.0], np.float32), np.float32(0.27)
    want = oracle.grid_eval(tape, corner, step, (n, n, n))
    sphere_only = oracle.grid_eval(tape[:12].tolist() + [word("_return")], corner, step, (n, n, n))
    assert same_bits(want, sphere_only)
    c4 = np.zeros(4, np.float32)
    c4[:3] = corner
    for specialise in (False, True):
        t = hip_util.Tape(tape, policy="0")
        if specialise:
            t.specialize()
        out = hip_util.Buffer(grid_eval.FLOAT4, (n, n, n))
        hip.k.grid_eval((n, n, n), None, t, c4, step, out).wait()
        assert same_bits(out.read().view(np.float32).reshape(n, n, n, 4), want), specialise
        w = hip_util.Buffer(np.float32, (n, n, n))
        hip.k.grid_eval_pymcubes((n, n, n), None, t, c4, step, w).wait()
        assert same_bits(w.read().reshape(-1), oracle.grid_eval_pymcubes(tape, corner, step, (n, n, n)).reshape(-1)), specialise
        out.release()
        w.release()


@pytest.mark.gpu
def test_kernel_families_are_built_and_loaded_separately(hip, tmp_path):
    """hu_tape_specialize_groups: per-tape code for one kernel family at a time -- a launch of a family that is not
    loaded yet runs the interpreter, one that is loaded runs per-tape code, the bytes are the oracle's either way -- and
    hu_tape_compile_groups builds a family on the host alone (what the background thread does)."""
    import ctypes
    import numpy as np
    import oracle
    from conftest import same_bits
    from codecad_amd import hip_util, examples, nodes, grid_eval
    from codecad_amd.hip_util import builder
    tape = nodes.make_program(examples.sponge(2))
    corner = np.array([-0.52, -0.49, -0.5], np.float32)
    step, dims = np.float32(1 / 32), (16, 16, 32)
    c4 = np.zeros(4, np.float32)
    c4[:3] = corner
    want = oracle.grid_eval(tape, corner, step, dims)
    count_want, cells_want = oracle.subdivision_step(tape, corner, step, np.float32(step * 0.87), dims)

    def check_both(t):
        out = hip_util.Buffer(grid_eval.FLOAT4, dims)
        hip.k.grid_eval(dims, None, t, c4, step, out).wait()
        assert same_bits(out.read().view(np.float32).reshape(dims + (4,)), want)
        counter, lst = hip_util.Buffer(np.uint32, 1), hip_util.Buffer(np.uint8, (dims[0] * dims[1] * dims[2], 4))
        counter.enqueue_fill(0)
        hip.k.subdivision_step(dims, None, t, c4, step, np.float32(step * 0.87), counter, lst).wait()
        n = int(counter.read()[0])
        assert n == count_want and sorted(map(tuple, lst.read()[:n].tolist())) == sorted(map(tuple, cells_want.tolist()))
        for b in (out, counter, lst):
            b.release()

    t = hip_util.Tape(tape, policy="0")
    check_both(t)
    assert t.groups == 0
    t._specialize(only_if_cached=False, groups=1)          # one kernel: the dense float4 grid over boxes
    assert t.groups == 1 and t.specialized
    check_both(t)                                          # (ragged extents and everything else: interpreter)
    t._specialize(only_if_cached=False, groups=hip_util.SPEC_DENSE)     # the dense family: the missing four as one image
    assert t.groups == hip_util.SPEC_DENSE
    check_both(t)                                          # grid_eval: per-tape code; subdivision_step: interpreter
    t._specialize(only_if_cached=False, groups=hip_util.SPEC_CLASSIFY)
    assert t.groups == hip_util.SPEC_DENSE | hip_util.SPEC_CLASSIFY
    check_both(t)
    t.specialize()
    assert t.groups == hip_util.SPEC_ALL
    check_both(t)
    # the host-only build of a set of kernels, and its image found again by the load -- as a whole, and kernel by kernel
    cache = tmp_path / "cache"
    cache.mkdir()
    size, hit = ctypes.c_size_t(0), ctypes.c_int(-1)
    ptr = t.host_tape.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
    blocks = hip_util.SPEC_BLOCKS
    assert hip.lib.hu_tape_compile_groups(ptr, t.host_tape.size, builder.CSRC.encode(), str(cache).encode(), blocks, ctypes.byref(size), ctypes.byref(hit)) == 0
    assert hit.value == 0 and size.value > 5000
    u = hip_util.Tape(tape, policy="0")
    u._specialize(only_if_cached=True, directory=str(cache), groups=blocks)
    assert u.groups == blocks and u.from_cache
    u._specialize(only_if_cached=True, directory=str(cache), groups=1)      # nothing cached for that kernel: stays as it is
    assert u.groups == blocks
    for kernel in (1, 2):                                                   # what the background builds leave: single kernels
        assert hip.lib.hu_tape_compile_groups(ptr, t.host_tape.size, builder.CSRC.encode(), str(cache).encode(), kernel, None, None) == 0
    v = hip_util.Tape(tape, policy="0")
    v._specialize(only_if_cached=True, directory=str(cache), groups=hip_util.SPEC_DENSE)    # no image of the family: its two cached kernels
    assert v.groups == 3 and not v.from_cache
    check_both(v)
    assert hip.lib.hu_tape_compile_groups(ptr, t.host_tape.size, builder.CSRC.encode(), str(cache).encode(), 1 << hip_util.SPEC_KERNELS, None, None) != 0
