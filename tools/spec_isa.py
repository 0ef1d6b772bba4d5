#!/usr/bin/env python3
"""Offline (no GPU): generate the specialised source of a tape, compile one kernel for gfx950 with
hipcc and print its instruction mix.  Usage: python tools/spec_isa.py [sponge4|<golden tape name>] [dense|scalar|blocks]
(dense / scalar: k_grid_eval with float4 / float output; blocks: k_grid_eval_blocks with float output)"""
import collections
import ctypes
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import codecad_amd as cc  # noqa: E402
from codecad_amd.hip_util import _lib  # noqa: E402


def source_of(tape):
    lib = _lib.load()
    tape = np.ascontiguousarray(tape, dtype=np.float32)
    needed = ctypes.c_size_t(0)
    p = tape.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
    assert lib.hu_tape_source(p, tape.size, None, 0, ctypes.byref(needed)) == 0, lib.hu_last_error()
    buf = ctypes.create_string_buffer(needed.value)
    assert lib.hu_tape_source(p, tape.size, buf, needed.value, ctypes.byref(needed)) == 0
    return buf.value.decode()


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "sponge4"
    layout = 1 if (len(sys.argv) > 2 and sys.argv[2] in ("scalar", "blocks")) else 0
    blocks = len(sys.argv) > 2 and sys.argv[2] == "blocks"
    if name.startswith("sponge"):
        tape = cc.nodes.make_program(cc.examples.sponge(int(name[6:])))
    else:
        import json
        shapes = json.load(open(os.path.join(ROOT, "tests/golden/ref_tapes.json")))["shapes"]
        tape = np.array({s["name"]: s for s in shapes}[name]["tape_u32"], dtype=np.uint32).view(np.float32)
    out = "/tmp/spec_isa"
    os.makedirs(out, exist_ok=True)
    ragged = "ragged" in sys.argv[3:]
    if ragged:
        inst = ("template __global__ void sdfk::k_grid_eval_ragged<sdfk::JitEval, %d, 2>(const sdfk::JitEval, float, float, float, float, uint32_t, "
                "sdfk::Dim, sdfk::Dim, uint32_t, uint32_t, uint32_t, void*, const uint32_t*);\n" % layout)
    elif blocks:
        inst = ("template __global__ void sdfk::k_grid_eval_blocks<sdfk::JitEval, 1, 2>(const sdfk::JitEval, const int4*, const uint32_t*, uint32_t, "
                "uint32_t, uint32_t, double, double, double, double, float, uint32_t, sdfk::Dim, sdfk::Dim, void*, const uint32_t*);\n")
    else:
        inst = ("template __global__ void sdfk::k_grid_eval<sdfk::JitEval, %d, 2>(const sdfk::JitEval, float, float, float, float, uint32_t, "
                "sdfk::Dim, sdfk::Dim, uint32_t, uint32_t, uint32_t, void*, const uint32_t*);\n" % layout)
    src = source_of(tape) + "\n" + inst
    open(out + "/spec.hip", "w").write(src)
    extra = [a for a in sys.argv[3:] if a != "ragged"]
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I", ROOT + "/codecad_amd/csrc",
           "--cuda-device-only", "-S", "-o", out + "/spec.s", out + "/spec.hip"] + extra
    subprocess.run(cmd, check=True)
    text = open(out + "/spec.s").read()
    body = text[text.index("k_grid_eval"):]
    ops = collections.Counter()
    for line in body.splitlines():
        m = re.match(r"\s+([vs]_[a-z0-9_]+|ds_[a-z0-9_]+|global_[a-z0-9_]+|buffer_[a-z0-9_]+|scratch_[a-z0-9_]+)", line)
        if m:
            ops[m.group(1)] += 1
    valu = sum(c for o, c in ops.items() if o.startswith("v_"))
    salu = sum(c for o, c in ops.items() if o.startswith("s_"))
    print("VALU %d  SALU %d  packed %d" % (valu, salu, sum(c for o, c in ops.items() if o.startswith("v_pk_"))))
    for o, c in ops.most_common(40):
        print("  %-28s %d" % (o, c))
    for key in ("NumVgprs", "NumSgprs", "ScratchSize", "Occupancy"):
        m = re.search(r"; %s: .*" % key, body)
        if m:
            print(m.group(0).strip())


if __name__ == "__main__":
    main()
