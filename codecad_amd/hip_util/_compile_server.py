"""hipRTC in a process of its own: `python _compile_server.py <path of libhip_util.so>`.

Reads one JSON request per line on stdin -- {"tape": base64 of the float32 tape, "include": csrc directory,
"dir": cache directory, "groups": hu_spec_group bits} --, builds that tape's kernel families with
hu_tape_compile_groups (host only: this process never touches a GPU) into the cache directory, and answers one JSON
line on stdout: {"rc": 0} or {"rc": code, "error": text}.  Ends when stdin closes.

Why a process and not just a thread (codecad_amd/hip_util/buffer.py _BackgroundCompiler): the HIP runtime parses code
objects with the same compiler-support library hipRTC compiles with, behind one lock -- a hipModuleLoadData, or the
first launch of a kernel, in a process whose other thread is in the middle of a hipRTC build waits for that build
(measured: a 2.9 s stall loading a finished family while the next build ran).  In a process of its own the build holds
no lock of the application's.  Deliberately free of imports from the package (no torch, no numpy): it starts in a
fraction of a second.
"""
import base64
import ctypes
import json
import sys


def main():
    lib = ctypes.CDLL(sys.argv[1])
    fp = ctypes.POINTER(ctypes.c_float)
    lib.hu_tape_compile_groups.argtypes = [fp, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint32,
                                           ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_int)]
    lib.hu_tape_compile_groups.restype = ctypes.c_int
    lib.hu_last_error.restype = ctypes.c_char_p
    # the protocol gets a descriptor of its own: whatever the library, hipRTC or comgr print on fd 1 goes to stderr and
    # cannot end up between the replies
    import os
    out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    for line in sys.stdin:
        line = line.strip()
        if not line:
            continue
        try:
            req = json.loads(line)
            raw = base64.b64decode(req["tape"])
            n = len(raw) // 4
            buf = (ctypes.c_float * n).from_buffer_copy(raw[:4 * n])
            size, hit = ctypes.c_size_t(0), ctypes.c_int(0)
            rc = lib.hu_tape_compile_groups(buf, n, req["include"].encode(), req["dir"].encode(), int(req["groups"]),
                                            ctypes.byref(size), ctypes.byref(hit))
            reply = {"rc": int(rc)}
            if rc != 0:
                msg = lib.hu_last_error()
                reply["error"] = msg.decode(errors="replace") if msg else "hipRTC failed"
        except Exception as e:   # (a malformed request must not end the server)
            reply = {"rc": -100, "error": "%s: %s" % (type(e).__name__, e)}
        out.write(json.dumps(reply) + "\n")
        out.flush()


if __name__ == "__main__":
    main()
