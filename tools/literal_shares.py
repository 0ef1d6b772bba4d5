#!/usr/bin/env python3
"""CPU only: the shares tests/test_literal_oracle.py allows -- points skipped as ill-conditioned, directions beyond
1e-5 -- measured over all golden tapes and the random trees of the GPU differential test, summed up and written to
profiles/<tag>_literal_shares.json.  Usage: python tools/literal_shares.py [tag]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import test_literal_oracle as t  # noqa: E402
from codecad_amd import nodes  # noqa: E402
import random_trees  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
out = {}
for group in ("golden", "random"):
    acc = {"points": 0, "skipped": 0, "directions_compared": 0, "directions_over_1e-5": 0, "max_dir": 0.0, "max_rel_w": 0.0, "tapes": 0,
           "worst_skipped_share_of_a_tape": 0.0}
    if group == "golden":
        cases = []
        for name in sorted(t.GOLDEN):
            ref = t.GOLDEN[name]
            rng = np.random.default_rng(7)
            a, b = np.array(ref["bbox_a"]), np.array(ref["bbox_b"])
            lo = np.where(np.isfinite(a), a, -3.0) - 1.0
            hi = np.where(np.isfinite(b), b, 3.0) + 1.0
            pts = (lo + rng.random((4000, 3)) * (hi - lo)).astype(np.float32)
            if ref["dimension"] == 2:
                pts[:, 2] = 0
            cases.append((ref["tape"], pts, float(np.max(hi - lo)), 0.01, slice(None)))
    else:
        cases = []
        for kind, seed in [(3, s) for s in range(60)] + [(2, s) for s in range(36)]:
            tape = nodes.make_program(t._tree(kind, seed))
            rng = np.random.default_rng(seed)
            pts = [(rng.random((3000, 3)) * 10 - 5).astype(np.float32)]
            for corner, step, dims in random_trees.grids():
                ix = np.stack(np.meshgrid(*[np.arange(d, dtype=np.float32) for d in dims], indexing="ij"), axis=-1).reshape(-1, 3)
                pts.append((corner.astype(np.float32) + step * ix).astype(np.float32))
            pts = np.concatenate(pts)
            if kind == 2:
                pts[:, 2] = 0
            cases.append((tape, pts, 10.0, 0.03, slice(0, 3000)))
    for tape, pts, scale, ill, directions in cases:
        s = t.compare(tape, pts, scale, max_ill_share=ill, directions=directions)
        for k in ("points", "skipped", "directions_compared", "directions_over_1e-5"):
            acc[k] += s[k]
        acc["max_dir"] = max(acc["max_dir"], s["max_dir"])
        acc["max_rel_w"] = max(acc["max_rel_w"], s["max_rel_w"])
        acc["worst_skipped_share_of_a_tape"] = max(acc["worst_skipped_share_of_a_tape"], s["skipped"] / max(s["points"], 1))
        acc["tapes"] += 1
    acc["skipped_share"] = acc["skipped"] / max(acc["points"], 1)
    acc["direction_outlier_share"] = acc["directions_over_1e-5"] / max(acc["directions_compared"], 1)
    out[group] = acc
path = os.path.join(ROOT, "profiles", "%s_literal_shares.json" % tag)
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out, indent=1))
