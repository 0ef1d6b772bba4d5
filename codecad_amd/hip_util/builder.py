"""Build libhip_util.so for gfx950 with hipcc (in-tree, next to this file).

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting
.so travels to the GPU box with the repository snapshot.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.normpath(os.path.join(HERE, "..", "csrc"))
INCLUDE = os.path.normpath(os.path.join(HERE, "..", "..", "include"))
LIB_PATH = os.path.join(HERE, "libhip_util.so")
SOURCES = ["hip_util.hip", "render.hip", "sort.hip", "exchange.hip", "mesh.hip"]


def headers():
    """Every header under csrc/ (a glob, like bench.csrc_hash(): a new header cannot be forgotten here)."""
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hpp"))


# Strict IEEE arithmetic is part of the contract (DESIGN.md "Canonical arithmetic"):
# no contraction, no fast-math, correctly rounded sqrt/divide.
HIPCC_FLAGS = [
    "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC",
    "-ffp-contract=off", "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt",
    "-Wall", "-Wno-unused-function",
]
# Which translation unit carries INTERPRETER_FLAGS, and why.  -structurizecfg-skip-uniform-regions is NOT harmless:
# it once gave a divergent loop with a second, uniform exit its exit-dependent value from a scalar branch
# (csrc/exchange.hip tells the story).  So it is confined to ONE unit, hip_util.hip, whose kernels are only
#   k_grid_eval / k_grid_eval_blocks / k_classify over InterpEval: the interpreter's wave-uniform dispatch loop around
#   branch-free ops (ops with divergent branches or loops are __noinline__ functions), plus straight-line index
#   arithmetic, stores and the ballot compaction -- no divergent loop with more than one exit;
# every other kernel -- ray caster, bitmap, 2D contouring, mass integrals, self-test (render.hip), the exchange step,
# the sort and marching cubes -- is built without it.  tests/test_hip_util_host.py checks both halves of that from the
# ISA: which kernels the flagged object holds, and that its loops have the shape described here.
FLAGGED_SOURCES = ("hip_util.hip",)
INTERPRETER_FLAGS = [
    # The tape dispatch loop is wave-uniform control flow (scalar branches).  By default the
    # AMDGPU backend still runs StructurizeCFG over it and turns the opcode switch into a
    # chain of flag-guarded blocks (~37 SALU + phi copies per tape instruction, measured
    # with rocprofv3: profiles/r01_*).  Skipping uniform regions keeps it a compare tree +
    # one branch back.  Every op inlined into the loop is written branch-free (selects) and
    # ops with divergent branches are __noinline__, so the loop region IS uniform.
    "-mllvm", "-structurizecfg-skip-uniform-regions",
    # Keep the grouped scalar loads of the tape records at the top of the dispatch loop (the
    # interpreter issues a whole fetch group, then waits once); MachineSink would push each
    # load down to its first use and re-expose the scalar-cache latency per instruction.
    "-mllvm", "-disable-machine-sink",
]


def find_hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    return None


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    built = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in SOURCES + headers()] + [os.path.join(INCLUDE, "hip_util.h"), __file__]
    return any(os.path.getmtime(d) > built for d in deps)


def build(force=False, verbose=False, extra_flags=(), out_path=None):
    """Compile if needed; returns the library path.  Raises RuntimeError on failure.
    `extra_flags`/`out_path` build an experimental variant next to the default library
    (selected at run time with CODECAD_AMD_LIB=<path>)."""
    if out_path is not None:
        return _compile(out_path, list(extra_flags), verbose)
    if not force and not is_stale():
        if not _pch_files():
            prepare_pch(verbose)
        return LIB_PATH
    path = _compile(LIB_PATH, [], verbose)
    prepare_pch(verbose)
    return path


PCH_DIR = os.path.join(os.path.dirname(LIB_PATH), "pch")


def _pch_files():
    try:
        return [f for f in os.listdir(PCH_DIR) if f.endswith(".pch")]
    except OSError:
        return []


def prepare_pch(verbose=False):
    """The precompiled header of the per-tape builds (hu_spec_pch_prepare, include/hip_util.h), next to the library:
    made in a process of its own -- host only, and with the hipRTC the compile servers will use (a process that has
    imported torch runs the hipRTC of the torch wheel, whose clang is not installed).  Best effort: returns the path or
    None (the per-tape builds then parse their headers as before); older headers in the directory are removed."""
    import sys
    code = ("import ctypes, sys\n"
            "lib = ctypes.CDLL(sys.argv[1])\n"
            "buf = ctypes.create_string_buffer(4096)\n"
            "lib.hu_spec_pch_prepare.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t]\n"
            "rc = lib.hu_spec_pch_prepare(sys.argv[2].encode(), sys.argv[3].encode(), buf, 4096)\n"
            "print(buf.value.decode() if rc == 0 else '')\n")
    try:
        os.makedirs(PCH_DIR, exist_ok=True)
        out = subprocess.run([sys.executable, "-c", code, LIB_PATH, CSRC, PCH_DIR], capture_output=True, text=True, timeout=300)
        paths = [p for p in out.stdout.splitlines() if p.endswith(".pch")] if out.returncode == 0 else []
    except (OSError, subprocess.SubprocessError):
        paths = []
    path = " ".join(paths)
    if paths:
        keep = tuple(os.path.basename(p)[:-len(".pch")] for p in paths)      # (one per optimisation level the builds use)
        for f in os.listdir(PCH_DIR):
            if not f.startswith(keep):
                full = os.path.join(PCH_DIR, f)
                if os.path.isdir(full):
                    shutil.rmtree(full, ignore_errors=True)
                else:
                    os.unlink(full)
    if verbose:
        print("precompiled header:", path or "none (no clang++ next to hipRTC: per-tape builds parse their headers)")
    return path or None


def _compile(LIB_PATH, extra_flags, verbose):
    hipcc = find_hipcc()
    if hipcc is None:
        raise RuntimeError("hipcc not found: cannot build libhip_util.so (set HIPCC or install ROCm)")
    tmp = LIB_PATH + ".tmp.%d" % os.getpid()
    objects, procs = [], []
    for s in SOURCES:      # one object per source, each with its own flags, compiled side by side
        obj = "%s.%s.o" % (tmp, s)
        flags = HIPCC_FLAGS + (INTERPRETER_FLAGS if s in FLAGGED_SOURCES else []) + extra_flags
        cmd = [hipcc] + flags + ["-I", INCLUDE, "-c", "-o", obj, os.path.join(CSRC, s)]
        objects.append(obj)
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
    errors = []
    for cmd, proc in procs:
        _, err = proc.communicate()
        if proc.returncode != 0:
            errors.append("hipcc failed:\n" + " ".join(cmd) + "\n" + err[-4000:])
    if not errors:
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objects + ["-lhiprtc"]
        proc = subprocess.run(cmd, capture_output=True, text=True)
        if proc.returncode != 0:
            errors.append("hipcc link failed:\n" + " ".join(cmd) + "\n" + proc.stderr[-4000:])
    for obj in objects:
        if os.path.exists(obj):
            os.unlink(obj)
    if errors:
        if os.path.exists(tmp):
            os.unlink(tmp)
        raise RuntimeError("\n".join(errors))
    os.replace(tmp, LIB_PATH)
    if verbose:
        print("built", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    build(force=True, verbose=True)
