"""ctypes loader for oracle/_build/liboracle.so (built by oracle/Makefile).

The functions mirror the reference kernels' argument lists (reference grid_eval.cl:2-4,
23-25; subdivision.cl:12-16; mass_properties.cl:7-12) with numpy arrays for buffers.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")
_LITERAL_PATH = os.path.join(_HERE, "_build", "libliteral.so")
_LITERAL64_PATH = os.path.join(_HERE, "_build", "libliteral64.so")
_lib = None
_literal = {}

__all__ = ["build", "lib", "evaluate_points", "grid_eval", "grid_eval_pymcubes", "ray_caster", "bitmap", "process_polygon", "marching_cubes", "stl_records", "STL_RECORD",
           "subdivision_step", "mass_properties", "det_math", "evaluate_points_literal", "grid_distance_literal",
           "literal_scene", "mass_properties_literal"]

_f32p = ctypes.POINTER(ctypes.c_float)
_u32p = ctypes.POINTER(ctypes.c_uint32)
_u8p = ctypes.POINTER(ctypes.c_uint8)


def build(force=False):
    """Compile the oracle with gcc (a few seconds).  Building the checker is not using it."""
    src = [os.path.join(_HERE, f) for f in ("sdf_oracle.c", "sdf_literal.c", "det_math.h", "mc_table.h", "Makefile")]
    libs = (_LIB_PATH, _LITERAL_PATH, _LITERAL64_PATH)
    if (not force and all(os.path.exists(b) for b in libs)
            and all(min(os.path.getmtime(b) for b in libs) >= os.path.getmtime(s) for s in src)):
        return _LIB_PATH
    subprocess.run(["make", "-C", _HERE, "-B"], check=True, capture_output=True)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        for name in ("oracle_evaluate_points", "oracle_grid_eval", "oracle_grid_eval_pymcubes",
                     "oracle_subdivision_step", "oracle_mass_properties", "oracle_det_math"):
            getattr(_lib, name).restype = ctypes.c_int
    return _lib


def literal_lib(double=False):
    """The frozen formula-for-formula restatement (sdf_literal.c), its own library; double=True: the same
    formulas evaluated in binary64."""
    if double not in _literal:
        path = _LITERAL64_PATH if double else _LITERAL_PATH
        if not os.path.exists(path):
            build()
        _literal[double] = ctypes.CDLL(path)
        for name in ("oracle_evaluate_points_literal", "oracle_grid_distance_literal"):
            getattr(_literal[double], name).restype = ctypes.c_int
    return _literal[double]


def _tape(tape):
    t = np.ascontiguousarray(tape, dtype=np.float32)
    return t, t.ctypes.data_as(_f32p), ctypes.c_int(t.size)


def _corner(corner):
    c = np.zeros(4, dtype=np.float32)
    c[:3] = np.asarray(corner, dtype=np.float32).reshape(-1)[:3]
    return c


def _dims(dims):
    d = np.ones(3, dtype=np.uint32)
    dims = tuple(int(x) for x in dims)
    d[:len(dims)] = dims
    return d


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("oracle %s failed with code %d (malformed tape?)" % (what, rc))


def evaluate_points(tape, points):
    """evaluate() at each row of `points` (n,3) -> (n,4) float32 (nx, ny, nz, distance)."""
    t, tp, tn = _tape(tape)
    pts = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 3)
    out = np.empty((pts.shape[0], 4), dtype=np.float32)
    _check(lib().oracle_evaluate_points(tp, tn, pts.ctypes.data_as(_f32p),
                                        ctypes.c_int(pts.shape[0]),
                                        out.ctypes.data_as(_f32p)), "evaluate_points")
    return out


def evaluate_points_literal(tape, points, double=False):
    """evaluate() at each row of `points` with the reference's formulas verbatim (sdf_literal.c) -> (n,4)
    float32, or float64 with double=True (the same formulas and binary32 constants evaluated in binary64)."""
    t, tp, tn = _tape(tape)
    pts = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 3)
    out = np.empty((pts.shape[0], 4), dtype=np.float64 if double else np.float32)
    _check(literal_lib(double).oracle_evaluate_points_literal(tp, tn, pts.ctypes.data_as(_f32p), ctypes.c_int(pts.shape[0]),
                                                              ctypes.c_void_p(out.ctypes.data)), "evaluate_points_literal")
    return out


def grid_distance_literal(tape, corner, step, dims, double=False):
    """Distances of the reference's formulas at the kernels' sample points (corner + step * gid) -> dims."""
    t, tp, tn = _tape(tape)
    c, d = _corner(corner), _dims(dims)
    out = np.empty(tuple(int(x) for x in d), dtype=np.float64 if double else np.float32)
    _check(literal_lib(double).oracle_grid_distance_literal(tp, tn, c.ctypes.data_as(_f32p), ctypes.c_float(step),
                                                            d.ctypes.data_as(_u32p), ctypes.c_void_p(out.ctypes.data)),
           "grid_distance_literal")
    return out


def grid_eval(tape, corner, step, dims, threads=1):
    """Reference kernel grid_eval: float4 per voxel, shape dims+(4,), index z + sz*(y + sy*x)."""
    t, tp, tn = _tape(tape)
    c, d = _corner(corner), _dims(dims)
    out = np.empty(tuple(int(x) for x in d) + (4,), dtype=np.float32)
    _check(lib().oracle_grid_eval(tp, tn, c.ctypes.data_as(_f32p), ctypes.c_float(step),
                                  d.ctypes.data_as(_u32p), out.ctypes.data_as(_f32p),
                                  ctypes.c_int(threads)), "grid_eval")
    return out


def grid_eval_pymcubes(tape, corner, step, dims, threads=1):
    """Reference kernel grid_eval_pymcubes: flat float array, index z + (x + (sy-1-y)*sx)*sz."""
    t, tp, tn = _tape(tape)
    c, d = _corner(corner), _dims(dims)
    out = np.empty(int(d[0]) * int(d[1]) * int(d[2]), dtype=np.float32)
    _check(lib().oracle_grid_eval_pymcubes(tp, tn, c.ctypes.data_as(_f32p), ctypes.c_float(step),
                                           d.ctypes.data_as(_u32p), out.ctypes.data_as(_f32p),
                                           ctypes.c_int(threads)), "grid_eval_pymcubes")
    return out


def subdivision_step(tape, corner, step, thr, dims):
    """Reference kernel subdivision_step -> (count, uchar4 list[:count]) in gid order."""
    t, tp, tn = _tape(tape)
    c, d = _corner(corner), _dims(dims)
    n = int(d[0]) * int(d[1]) * int(d[2])
    counter = np.zeros(1, dtype=np.uint32)
    lst = np.zeros((n, 4), dtype=np.uint8)
    _check(lib().oracle_subdivision_step(tp, tn, c.ctypes.data_as(_f32p), ctypes.c_float(step),
                                         ctypes.c_float(thr), d.ctypes.data_as(_u32p),
                                         counter.ctypes.data_as(_u32p),
                                         lst.ctypes.data_as(_u8p)), "subdivision_step")
    return int(counter[0]), lst[:int(counter[0])]


def mass_properties(tape, corner, step, thr, dims):
    """Reference kernel mass_properties -> (sum[10] uint32, count, uchar4 list[:count])."""
    t, tp, tn = _tape(tape)
    c, d = _corner(corner), _dims(dims)
    n = int(d[0]) * int(d[1]) * int(d[2])
    sums = np.zeros(10, dtype=np.uint32)
    counter = np.zeros(1, dtype=np.uint32)
    lst = np.zeros((n, 4), dtype=np.uint8)
    _check(lib().oracle_mass_properties(tp, tn, c.ctypes.data_as(_f32p), ctypes.c_float(step),
                                        ctypes.c_float(thr), d.ctypes.data_as(_u32p),
                                        sums.ctypes.data_as(_u32p),
                                        counter.ctypes.data_as(_u32p),
                                        lst.ctypes.data_as(_u8p)), "mass_properties")
    return sums, int(counter[0]), lst[:int(counter[0])]


_DET_OPS = {"atan2": 0, "sincos": 1, "tan": 2, "acos": 3, "fmod": 4, "remainder": 5, "hypot": 6}


def det_math(op, a, b=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(a if b is None else b, dtype=np.float32)
    out = np.empty_like(a)
    out2 = np.empty_like(a)
    _check(lib().oracle_det_math(ctypes.c_int(_DET_OPS[op]), a.ctypes.data_as(_f32p),
                                 b.ctypes.data_as(_f32p), ctypes.c_int(a.size),
                                 out.ctypes.data_as(_f32p), out2.ctypes.data_as(_f32p)), "det_math")
    return (out, out2) if op == "sincos" else out


class literal_scene:
    """with literal_scene(tape): the renderers below run over the frozen literal-formula evaluate() of sdf_literal.c
    instead of the canonical one (oracle_set_scene_evaluator); binary32.  Not re-entrant."""

    def __init__(self, tape):
        self.tape = np.ascontiguousarray(tape, dtype=np.float32)

    def __enter__(self):
        lit = literal_lib()
        lit.oracle_literal_open.restype = ctypes.c_void_p
        lit.oracle_literal_open.argtypes = [_f32p, ctypes.c_int]
        lit.oracle_literal_close.argtypes = [ctypes.c_void_p]
        self.handle = lit.oracle_literal_open(self.tape.ctypes.data_as(_f32p), ctypes.c_int(self.tape.size))
        if not self.handle:
            raise MemoryError("oracle_literal_open")
        lib().oracle_set_scene_evaluator.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        lib().oracle_set_scene_evaluator.restype = None
        lib().oracle_set_scene_evaluator(ctypes.cast(lit.oracle_literal_eval, ctypes.c_void_p), ctypes.c_void_p(self.handle))
        return self

    def __exit__(self, *exc):
        lib().oracle_set_scene_evaluator(None, None)
        literal_lib().oracle_literal_close(ctypes.c_void_p(self.handle))


def mass_properties_literal(tape, corner, step, thr, dims):
    """The reference kernel mass_properties (mass_properties.cl:7-56) over the LITERAL evaluate(): distances from
    sdf_literal.c (grid_distance_literal, binary32), classification and the integer moment sums in numpy.  Same
    return value as mass_properties(); the list in (x, y, z) scan order."""
    dims = tuple(int(v) for v in dims)
    w = grid_distance_literal(tape, corner, step, dims).astype(np.float32).reshape(dims)
    thr = np.float32(thr)
    inside = w <= -thr                                    # mass_properties.cl:31
    ambiguous = ~inside & (w < thr)                       # :43
    x, y, z = np.nonzero(inside)
    x, y, z = x.astype(np.uint64), y.astype(np.uint64), z.astype(np.uint64)
    sums = np.array([(x * x).sum(), (x * y).sum(), (x * z).sum(), x.sum(), (y * y).sum(), (y * z).sum(), y.sum(),
                     (z * z).sum(), z.sum(), len(x)], dtype=np.uint64)
    assert (sums < 2 ** 32).all()
    cells = np.stack(np.nonzero(ambiguous) + (np.zeros(int(ambiguous.sum()), dtype=np.int64),), axis=1).astype(np.uint8)
    return sums.astype(np.uint32), len(cells), cells


def ray_caster(tape, origin, forward, up, right, pixel_tolerance, box_radius, min_distance, max_distance, floor_z,
               options, size, threads=1):
    """Reference kernel ray_caster (rendering/ray_caster.cl:146-256): uchar RGB in the kernel's
    layout (w, h, 3) with index (y + h*x)*3."""
    t, tp, tn = _tape(tape)
    w, h = int(size[0]), int(size[1])
    out = np.zeros((w, h, 3), dtype=np.uint8)
    vec = [np.ascontiguousarray(np.asarray(v, dtype=np.float64)[:3], dtype=np.float32) for v in (origin, forward, up, right)]
    lib().oracle_ray_caster.restype = ctypes.c_int
    _check(lib().oracle_ray_caster(tp, tn, *[v.ctypes.data_as(_f32p) for v in vec], ctypes.c_float(pixel_tolerance),
                                   ctypes.c_float(box_radius), ctypes.c_float(min_distance),
                                   ctypes.c_float(max_distance), ctypes.c_float(floor_z), ctypes.c_uint32(int(options)),
                                   ctypes.c_uint32(w), ctypes.c_uint32(h), out.ctypes.data_as(_u8p),
                                   ctypes.c_int(threads)), "ray_caster")
    return out


def bitmap(tape, origin, step_size, size):
    """Reference kernel bitmap (rendering/bitmap.cl:1-18), layout as ray_caster."""
    t, tp, tn = _tape(tape)
    w, h = int(size[0]), int(size[1])
    out = np.zeros((w, h, 3), dtype=np.uint8)
    o = np.ascontiguousarray(np.asarray(origin, dtype=np.float64)[:3], dtype=np.float32)
    lib().oracle_bitmap.restype = ctypes.c_int
    _check(lib().oracle_bitmap(tp, tn, o.ctypes.data_as(_f32p), ctypes.c_float(step_size), ctypes.c_uint32(w),
                               ctypes.c_uint32(h), out.ctypes.data_as(_u8p)), "bitmap")
    return out


def process_polygon(corners, box_corner, box_step):
    """Reference kernel process_polygon (rendering/polygon2d.cl:82-175) over a float4 corner grid of shape
    (gx, gy, 4) -> (vertices float32 (cells, 2), links uint32 (cells,), starts uint32 (n,)), cells =
    (gx-1)*(gy-1)*2, cell index t + 2*(y + (gy-1)*x).  Vertices of empty cells (link 0xffffffff) are NaN."""
    c = np.ascontiguousarray(corners, dtype=np.float32)
    gx, gy = int(c.shape[0]), int(c.shape[1])
    assert c.shape == (gx, gy, 4)
    cells = (gx - 1) * (gy - 1) * 2
    vertices = np.full((cells, 2), np.nan, dtype=np.float32)
    links = np.zeros(cells, dtype=np.uint32)
    starts = np.zeros(max((gx - 1) + (gy - 1), 1) * 2, dtype=np.uint32)
    count = ctypes.c_uint32(0)
    o = np.ascontiguousarray(np.asarray(box_corner, dtype=np.float64)[:2], dtype=np.float32)
    lib().oracle_process_polygon.restype = ctypes.c_int
    _check(lib().oracle_process_polygon(c.ctypes.data_as(_f32p), ctypes.c_uint32(gx), ctypes.c_uint32(gy),
                                        o.ctypes.data_as(_f32p), ctypes.c_float(box_step),
                                        vertices.ctypes.data_as(_f32p), links.ctypes.data_as(_u32p),
                                        starts.ctypes.data_as(_u32p), ctypes.byref(count)), "process_polygon")
    return vertices, links, starts[:count.value].copy()


def marching_cubes(field):
    """Marching cubes of one block (3D float32 array, inside = value <= 0) -> (vertices float64 (n, 3) in array
    coordinates, triangles uint32 (m, 3)); ordering and orientation as documented in sdf_oracle.c."""
    f = np.ascontiguousarray(field, dtype=np.float32)
    assert f.ndim == 3
    fn = lib().oracle_marching_cubes
    fn.restype = ctypes.c_int
    nv, nt = ctypes.c_uint64(0), ctypes.c_uint64(0)
    dims = [ctypes.c_uint32(int(d)) for d in f.shape]
    _f64p = ctypes.POINTER(ctypes.c_double)
    _check(fn(f.ctypes.data_as(_f32p), *dims, ctypes.cast(None, _f64p), ctypes.c_uint64(0), ctypes.cast(None, _u32p),
              ctypes.c_uint64(0), ctypes.byref(nv), ctypes.byref(nt)), "marching_cubes")
    vertices = np.zeros((nv.value, 3), dtype=np.float64)
    triangles = np.zeros((nt.value, 3), dtype=np.uint32)
    _check(fn(f.ctypes.data_as(_f32p), *dims, vertices.ctypes.data_as(_f64p), ctypes.c_uint64(nv.value),
              triangles.ctypes.data_as(_u32p), ctypes.c_uint64(nt.value), ctypes.byref(nv), ctypes.byref(nt)),
           "marching_cubes")
    return vertices, triangles


STL_RECORD = np.dtype([("normal", "<f4", 3), ("vectors", "<f4", (3, 3)), ("attr", "<u2")])


def stl_records(vertices, triangles):
    """Binary STL records of an indexed mesh, as the reference's exporter produces them (reference
    rendering/stl_renderer.py:14-24): every corner assigned into numpy-stl 1.8.0's float32 `vectors`
    (round to nearest), normals from `Mesh.update_normals` on save = numpy.cross(v1 - v0, v2 - v0) in float32,
    unnormalised, attribute word 0.  numpy-stl is a pinned dependency (requirements.txt:12) that is not in the
    reference tree: this restates its published record layout (50 bytes, little endian)."""
    vertices = np.asarray(vertices, dtype=np.float64).reshape(-1, 3)
    triangles = np.asarray(triangles, dtype=np.int64).reshape(-1, 3)
    rec = np.zeros(len(triangles), dtype=STL_RECORD)
    rec["vectors"] = vertices[triangles].astype(np.float32)
    v = rec["vectors"]
    rec["normal"] = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0])
    return rec
