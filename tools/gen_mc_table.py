#!/usr/bin/env python3
"""Derive the marching-cubes triangle table from first principles and write it as C headers.

Conventions (Lorensen & Cline 1987, numbering as popularised by P. Bourke): cube corners
v0..v7 = (0,0,0) (1,0,0) (1,1,0) (0,1,0) (0,0,1) (1,0,1) (1,1,1) (0,1,1); edges e0..e11 =
v0v1 v1v2 v2v3 v3v0 v4v5 v5v6 v6v7 v7v4 v0v4 v1v5 v2v6 v3v7; bit m of the case index is set when
corner m is INSIDE (value <= iso).

Construction: on every cube face the iso-line segments are a function of that face's four corner
signs only -- with two crossings they are joined, with four (inside corners on a diagonal) each
INSIDE corner is cut off on its own -- so two cells sharing a face always agree and the surface
has no cracks.  Segments are oriented with the inside on their right seen from outside the cube;
following them gives closed loops; each loop is triangulated without any diagonal that lies in a
cube face (the neighbouring cell could put the same diagonal there: four triangles on one edge).  Triangles wind
anticlockwise seen from the outside (positive) region, i.e. normals point out of the solid.

Usage: python tools/gen_mc_table.py [--check]   (writes codecad_amd/csrc/mc_table.hpp and oracle/mc_table.h)
"""
import itertools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CORNERS = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)]
EDGES = [(0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4), (0, 4), (1, 5), (2, 6), (3, 7)]
EDGE_OF = {frozenset(e): i for i, e in enumerate(EDGES)}


def faces():
    """Each face as its 4 corner indices in anticlockwise order seen from OUTSIDE the cube."""
    out = []
    for axis in range(3):
        for side in (0, 1):
            u, v = [a for a in range(3) if a != axis]
            # (u, v, axis) right-handed?  outward normal is +axis for side 1, -axis for side 0
            right_handed = (u, v, axis) in ((0, 1, 2), (1, 2, 0), (2, 0, 1))
            ring = [(0, 0), (1, 0), (1, 1), (0, 1)]          # anticlockwise in the (u, v) plane
            if right_handed != (side == 1):
                ring = ring[::-1]
            idx = []
            for cu, cv in ring:
                c = [0, 0, 0]
                c[axis], c[u], c[v] = side, cu, cv
                idx.append(CORNERS.index(tuple(c)))
            out.append(idx)
    return out


FACES = faces()


def case_triangles(case):
    inside = [(case >> m) & 1 for m in range(8)]
    nxt = {}   # edge -> next edge along an oriented segment
    for ring in FACES:
        # walking the ring anticlockwise (seen from outside): a segment ENTERS the face region of an
        # inside corner... orient every segment so the inside corner(s) it cuts off lie on its left.
        for k in range(4):
            a, b, c = ring[k - 1], ring[k], ring[(k + 1) % 4]
            if not inside[b]:
                continue
            # maximal run of inside corners starting at b going anticlockwise, only if a is outside
            if inside[a]:
                continue
            run_end = k
            while inside[ring[(run_end + 1) % 4]] and (run_end + 1 - k) < 4:
                run_end += 1
            if run_end - k >= 3:
                continue   # whole face inside (cannot happen: a is outside)
            last = ring[run_end % 4]
            after = ring[(run_end + 1) % 4]
            e_in = EDGE_OF[frozenset((a, b))]         # crossing before the run
            e_out = EDGE_OF[frozenset((last, after))]  # crossing after the run
            # the run of inside corners lies between e_in and e_out going anticlockwise; travelling from
            # e_in to e_out keeps it on the right seen from outside the cube, which makes the triangles
            # anticlockwise seen from the outside REGION (checked against the gradient in self_check)
            assert e_in not in nxt
            nxt[e_in] = e_out
    loops, seen = [], set()
    for start in sorted(nxt):
        if start in seen:
            continue
        loop, e = [], start
        while e not in seen:
            seen.add(e)
            loop.append(e)
            e = nxt[e]
        assert e == start
        loops.append(loop)
    tris = []
    for loop in loops:
        assert len(loop) >= 3
        tris.extend(triangulate(tuple(loop)))
    return tris


def on_common_face(e1, e2):
    """Do two cube edges lie on one cube face?  A loop diagonal between such edges runs inside that
    face, where the neighbouring cell could put the same diagonal: four triangles on one edge."""
    for ring in FACES:
        r = set(ring)
        if set(EDGES[e1]) <= r and set(EDGES[e2]) <= r:
            return True
    return False


def triangulate(loop):
    """Triangulation of the loop polygon with the fewest diagonals lying in a cube face (none, for every
    case: asserted in build()); among those the first in a fixed recursive order, so the table is
    reproducible.  Orientation follows the loop."""
    from functools import lru_cache

    @lru_cache(maxsize=None)
    def best(i, j):
        # polygon loop[i..j] (indices into loop, i < j), edge (i, j) is already present
        if j - i < 2:
            return 0, ()
        out = None
        for k in range(i + 1, j):
            cost = 0
            if k - i >= 2 and on_common_face(loop[i], loop[k]):
                cost += 1
            if j - k >= 2 and on_common_face(loop[k], loop[j]):
                cost += 1
            c1, t1 = best(i, k)
            c2, t2 = best(k, j)
            cand = (cost + c1 + c2, ((loop[i], loop[k], loop[j]),) + t1 + t2)
            if out is None or cand[0] < out[0]:
                out = cand
        return out

    cost, tris = best(0, len(loop) - 1)
    triangulate.worst = max(getattr(triangulate, "worst", 0), cost)
    return list(tris)


def build():
    table = [case_triangles(c) for c in range(256)]
    assert table[0] == [] and table[255] == []
    assert triangulate.worst == 0, "some case needs a diagonal inside a cube face"
    return table


def self_check(table):
    """Orientation against the gradient of the trilinear interpolant, and crack-freeness on random fields."""
    import numpy as np
    rng = np.random.default_rng(1)

    def edge_point(e, vals):
        a, b = EDGES[e]
        t = (0.0 - vals[a]) / (vals[b] - vals[a])
        return np.array(CORNERS[a], float) + t * (np.array(CORNERS[b], float) - np.array(CORNERS[a], float))

    # orientation: the eight one-corner cases against the gradient (the corner is inside, the normal
    # must point away from it); consistency of all other cases follows from the directed-edge pairing
    # below, and the signed volume check pins the sign of whole closed surfaces
    for m in range(8):
        (tri,) = table[1 << m]
        vals = np.ones(8)
        vals[m] = -1.0
        p = [edge_point(e, vals) for e in tri]
        normal = np.cross(p[1] - p[0], p[2] - p[0])
        assert np.dot(normal, (p[0] + p[1] + p[2]) / 3 - np.array(CORNERS[m], float)) > 0, (m, tri)
    # random sign fields: every directed edge of the mesh must be matched by its reverse exactly once
    for trial in range(20):
        n = 7
        f = rng.uniform(-1, 1, (n, n, n))
        f[0, :, :] = f[-1, :, :] = f[:, 0, :] = f[:, -1, :] = f[:, :, 0] = f[:, :, -1] = 1.0   # closed surface
        directed = {}
        volume = 0.0
        for i, j, k in itertools.product(range(n - 1), repeat=3):
            vals = [f[i + x, j + y, k + z] for x, y, z in CORNERS]
            case = sum(1 << m for m in range(8) if vals[m] <= 0)
            for tri in table[case]:
                p = [edge_point(e, np.array(vals)) + np.array((i, j, k), float) for e in tri]
                volume += float(np.dot(p[0], np.cross(p[1], p[2]))) / 6
                keys = []
                for e in tri:
                    a, b = EDGES[e]
                    pa = (i + CORNERS[a][0], j + CORNERS[a][1], k + CORNERS[a][2])
                    pb = (i + CORNERS[b][0], j + CORNERS[b][1], k + CORNERS[b][2])
                    keys.append((min(pa, pb), max(pa, pb)))
                for u in range(3):
                    d = (keys[u], keys[(u + 1) % 3])
                    directed[d] = directed.get(d, 0) + 1
        for (a, b), cnt in directed.items():
            assert cnt == 1 and directed.get((b, a), 0) == 1, "crack or non-manifold edge"
        assert volume > 0, "closed surfaces around inside regions must have positive signed volume"
    return True


def render(table, cpp):
    width = 3 * max(len(t) for t in table) + 1
    rows = []
    for t in table:
        flat = [e for tri in t for e in tri]
        flat += [-1] * (width - len(flat))
        rows.append("    {" + ", ".join("%2d" % v for v in flat) + "},")
    counts = ["    " + ", ".join(str(len(table[c])) for c in range(r, r + 32)) + "," for r in range(0, 256, 32)]
    lines = ["// GENERATED by tools/gen_mc_table.py -- do not edit.  Marching-cubes case table derived from the",
             "// face-consistent construction described there: kMcTriangles[case] lists cube edges (three per",
             "// triangle, -1 terminated); kMcTriangleCount[case]; corner/edge numbering of Lorensen-Cline/Bourke,",
             "// bit m of `case` set when corner m is inside (value <= iso); triangles wind anticlockwise seen",
             "// from outside the solid.",
             "#ifndef MC_TABLE_H", "#define MC_TABLE_H",
             "#define MC_TABLE_WIDTH %d" % width]
    if cpp:   # initialiser macros, so that device code can define __constant__ copies
        lines.append("#define MC_TRIANGLES_INIT { \\")
        lines += [r + " \\" for r in rows]
        lines.append("}")
        lines.append("#define MC_TRIANGLE_COUNT_INIT { \\")
        lines += [c + " \\" for c in counts]
        lines.append("}")
    else:
        lines.append("static const signed char kMcTriangles[256][MC_TABLE_WIDTH] = {")
        lines += rows
        lines.append("};")
        lines.append("static const unsigned char kMcTriangleCount[256] = {")
        lines += counts
        lines.append("};")
    lines.append("#endif")
    return "\n".join(lines) + "\n"


def main():
    table = build()
    self_check(table)
    outputs = {os.path.join(ROOT, "codecad_amd", "csrc", "mc_table.hpp"): render(table, True),
               os.path.join(ROOT, "oracle", "mc_table.h"): render(table, False)}
    if "--check" in sys.argv:
        for path, text in outputs.items():
            assert open(path).read() == text, path + " is stale: run tools/gen_mc_table.py"
        print("tables up to date")
        return
    for path, text in outputs.items():
        open(path, "w").write(text)
    print("max triangles per case:", max(len(t) for t in table), " total:", sum(len(t) for t in table))


if __name__ == "__main__":
    main()
