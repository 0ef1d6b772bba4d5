"""ctypes binding of include/hip_util.h (one prototype per exported function)."""
import ctypes
import os
import re

from . import builder as _build

_c = ctypes
_vp, _i, _u32, _f, _d, _sz = _c.c_void_p, _c.c_int, _c.c_uint32, _c.c_float, _c.c_double, _c.c_size_t
_f4 = _c.POINTER(_c.c_float)
_u3 = _c.POINTER(_c.c_uint32)
_d3 = _c.POINTER(_c.c_double)
_pvp = _c.POINTER(_c.c_void_p)

# name -> argtypes; every function returns int except hu_last_error
PROTOTYPES = {
    "hu_abi_version": [],
    "hu_last_error": [],
    "hu_device_count": [_c.POINTER(_i)],
    "hu_set_device": [_i],
    "hu_device_name": [_i, _c.c_char_p, _sz],
    "hu_synchronize": [],
    "hu_malloc": [_pvp, _sz],
    "hu_free": [_vp],
    "hu_host_alloc": [_pvp, _sz],
    "hu_host_free": [_vp],
    "hu_memcpy_h2d": [_vp, _vp, _sz, _vp],
    "hu_memcpy_d2h": [_vp, _vp, _sz, _vp],
    "hu_memcpy_d2d": [_vp, _vp, _sz, _vp],
    "hu_memset": [_vp, _i, _sz, _vp],
    "hu_stream_create": [_pvp],
    "hu_stream_destroy": [_vp],
    "hu_stream_synchronize": [_vp],
    "hu_stream_wait_event": [_vp, _vp],
    "hu_event_create": [_pvp],
    "hu_event_destroy": [_vp],
    "hu_event_record": [_vp, _vp],
    "hu_event_synchronize": [_vp],
    "hu_event_elapsed_ms": [_vp, _vp, _c.POINTER(_f)],
    "hu_tape_create": [_f4, _sz, _pvp],
    "hu_tape_destroy": [_vp],
    "hu_tape_info": [_vp, _c.POINTER(_i), _c.POINTER(_i), _c.POINTER(_i)],
    "hu_grid_eval": [_vp, _f4, _f, _u3, _vp, _vp],
    "hu_grid_eval_pymcubes": [_vp, _f4, _f, _u3, _vp, _vp],
    "hu_subdivision_step": [_vp, _f4, _f, _f, _u3, _vp, _vp, _vp],
    "hu_mass_properties": [_vp, _f4, _f, _f, _u3, _vp, _vp, _vp, _vp],
    "hu_grid_eval_slab": [_vp, _f4, _f, _u3, _u32, _u32, _i, _vp, _vp],
    "hu_grid_eval_blocks": [_vp, _vp, _u32, _d, _d3, _f, _u3, _i, _vp, _vp],
    "hu_subdivision_level": [_vp, _vp, _u32, _c.c_int32, _u3, _i, _d, _d3, _f, _f, _vp, _vp, _u32, _vp],
    "hu_subdivision_level_indirect": [_vp, _vp, _vp, _u32, _c.c_int32, _u3, _i, _d, _d3, _f, _f, _vp, _vp, _u32, _vp],
    "hu_subdivision_level_owned": [_vp, _vp, _vp, _u32, _c.c_int32, _u3, _i, _d, _d3, _f, _f, _vp, _vp, _u32, _u32, _u32, _vp],
    "hu_grid_eval_blocks_indirect": [_vp, _vp, _vp, _u32, _d, _d3, _f, _u3, _i, _vp, _vp],
    "hu_slice_rows": [_vp, _u32, _u32, _u32, _u32, _vp, _u32, _vp, _vp],
    "hu_slice_rows_of": [_vp, _u32, _u32, _u32, _u32, _vp, _u32, _vp, _vp],
    "hu_mass_properties_level": [_vp, _vp, _u32, _d, _u3, _f, _f, _vp, _vp, _vp, _u32, _vp],
    "hu_mass_integrals": [_vp, _vp, _u32, _d, _vp, _u32, _vp],
    "hu_mass_properties_level_indirect": [_vp, _vp, _vp, _u32, _d, _u3, _f, _f, _vp, _vp, _vp, _u32, _vp],
    "hu_mass_properties_level_owned": [_vp, _vp, _vp, _u32, _d, _u3, _f, _f, _vp, _vp, _vp, _u32, _u32, _u32, _vp],
    "hu_mass_integrals_indirect": [_vp, _vp, _vp, _u32, _d, _vp, _u32, _vp],
    "hu_ray_caster": [_vp, _f4, _f4, _f4, _f4, _f, _f, _f, _f, _f, _u32, _u32, _u32, _vp, _vp],
    "hu_bitmap": [_vp, _f4, _f, _u32, _u32, _vp, _vp],
    "hu_process_polygon": [_f4, _f, _vp, _u3, _vp, _vp, _vp, _vp, _vp],
    "hu_process_polygon_blocks": [_vp, _vp, _u32, _d, _d3, _f, _u3, _vp, _vp, _vp, _vp, _vp],
    "hu_mesh_workgroups": [_u32, _u3, _c.POINTER(_c.c_uint64), _c.POINTER(_c.c_uint64), _c.POINTER(_c.c_uint64)],
    "hu_mesh_count": [_vp, _u32, _u3, _vp, _vp, _vp],
    "hu_mesh_emit": [_vp, _vp, _u32, _d, _d3, _d, _u3, _d, _vp, _vp, _vp, _vp, _vp, _vp],
    "hu_mesh_stl": [_vp, _vp, _c.c_uint64, _vp, _vp],
    "hu_sort_blocks": [_vp, _u32, _vp, _sz, _c.POINTER(_sz), _vp],
    "hu_tape_specialize": [_vp, _c.c_char_p],
    "hu_tape_specialize_cached": [_vp, _c.c_char_p, _c.c_char_p, _i, _c.POINTER(_i)],
    "hu_tape_specialize_groups": [_vp, _c.c_char_p, _c.c_char_p, _i, _u32, _c.POINTER(_i)],
    "hu_tape_specialized": [_vp, _c.POINTER(_i)],
    "hu_tape_prune_info": [_vp, _c.POINTER(_i), _c.POINTER(_i)],
    "hu_tape_compile_check": [_f4, _sz, _c.c_char_p, _c.POINTER(_sz)],
    "hu_tape_compile_cached": [_f4, _sz, _c.c_char_p, _c.c_char_p, _c.POINTER(_sz), _c.POINTER(_i)],
    "hu_tape_compile_groups": [_f4, _sz, _c.c_char_p, _c.c_char_p, _u32, _c.POINTER(_sz), _c.POINTER(_i)],
    "hu_spec_pch_prepare": [_c.c_char_p, _c.c_char_p, _c.c_char_p, _sz],
    "hu_selftest_math": [_c.POINTER(_c.c_uint64)],
    "hu_selftest_minmax3": [_c.POINTER(_c.c_uint64)],
    "hu_tape_source": [_f4, _sz, _c.c_char_p, _sz, _c.POINTER(_sz)],
    "hu_tape_listing": [_f4, _sz, _i, _c.c_char_p, _sz, _c.POINTER(_sz)],
}

# hu_spec_group (include/hip_util.h): the kernel families of per-tape code
SPEC_DENSE, SPEC_BLOCKS, SPEC_CLASSIFY, SPEC_RENDER, SPEC_ALL = 0x19c03, 0x6640c, 0x04f0, 0x0300, 0x7ffff   # sets of kernel bits (hip_util.h)
SPEC_KERNELS = 19

HEADER = os.path.normpath(os.path.join(os.path.dirname(__file__), "..", "..", "include", "hip_util.h"))

_lib = None


def header_symbols():
    """Function names declared in include/hip_util.h."""
    with open(HEADER) as f:
        text = f.read()
    return sorted(set(re.findall(r"\b(hu_[a-z0-9_]+)\s*\(", text)))


def load():
    """Load (building first if the sources are newer) and type the library.

    Raises RuntimeError -- never falls back to anything else -- when the library cannot be
    built or loaded.
    """
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch wheels bundle their own copy of the HIP runtime.  A process must not initialise the
    # system runtime first and torch's afterwards (torch then reports "No HIP GPUs are available"),
    # and device pointers / streams are only interchangeable when both sides resolved the same
    # runtime.  Importing torch first -- when it is installed -- gives one fixed, tested order.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    try:
        path = os.environ.get("CODECAD_AMD_LIB") or _build.build()
    except RuntimeError:
        if os.path.exists(_build.LIB_PATH):
            path = _build.LIB_PATH  # no hipcc here, but a prebuilt library travelled with us
        else:
            raise
    try:
        lib = ctypes.CDLL(path)
    except OSError as e:
        raise RuntimeError("cannot load the HIP extension %s: %s" % (path, e))
    for name, argtypes in PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = ctypes.c_char_p if name == "hu_last_error" else ctypes.c_int
    if lib.hu_abi_version() != 1:
        raise RuntimeError("libhip_util.so ABI version mismatch")
    _lib = lib
    return lib
