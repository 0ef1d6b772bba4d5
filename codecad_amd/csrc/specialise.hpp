// codecad_amd/csrc/specialise.hpp -- host only: the HIP source of a tape's per-tape kernels.
//
// The analogue of the reference's generate_fixed_eval_source_code (nodes/codegen.py:137-204): the decoded program
// is unrolled into straight-line code over the SAME op library (interp.hpp) -- one exec_one call per record with the
// record as a literal and its opcode as a template argument -- and compiled with hipRTC (hip_util.hip).
//
// Two forms, same bytes out (tests/test_gpu_variants.py, tests/test_gpu_random_shapes.py):
//
//  * plain: the full program, record by record.  Every union / intersection / subtraction selects a whole
//    (direction, distance) value, every primitive computes its direction, every transformation_from rotates one.
//
//  * deferred directions (tapes without rounded blends, built from the ops listed in `deferrable`): the DISTANCE of
//    such a tape never depends on a direction, and its direction is the direction of ONE primitive -- the one whose
//    distance survived every min / max on the way to the root -- pushed through the transformations on that path.
//      phase 1  the distance-only program (what subdivision_step / mass_properties / grid_eval_pymcubes run anyway),
//               plus, at each select, the comparison the full op would have made (`a.w < b.w`, exactly as
//               rounded_union writes it: same operands, same ties), kept as a wavefront mask in scalar registers;
//      phase 2  for every (primitive, path to the root): its lanes = the AND of the choices along the path (scalar
//               instructions); if the wavefront has any such lane, a wave-uniform branch recomputes the primitive's
//               local coordinates, its direction, and applies the path's transformations; three selects per voxel
//               merge it into the result.  A wavefront pays for the primitives that win somewhere in it (compact
//               bricks, kernels.hpp k_grid_eval: 1.8 of 13 for sponge(4) at 512^3) instead of for all of them.
//    Every operation that produces an output bit is the one the plain form executes on the same inputs, so the
//    results are identical; only operations whose results were going to be discarded are gone.
#pragma once

#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <string>
#include <vector>

#include "tape.hpp"

namespace sdf {

struct SpecProgram {
    std::vector<Rec> full;      // full program (slots of the full numbering), with the _return pad records
    std::vector<Rec> dist;      // distance-only program, same record positions; empty: the tape has a rounded blend
    int n_slots = 0, n_point_slots = 0, n_result_slots = 0;
};

namespace spec_detail {

inline std::string rec_literal(const Rec& r, bool clear_fold, uint32_t hdr)
{
    std::ostringstream o;
    char buf[32];
    std::snprintf(buf, sizeof buf, "0x%08xu", hdr);
    o << "Rec{" << buf << ", {";
    for (int i = 0; i < SDF_REC_DWORDS - 1; ++i) {
        uint32_t bits;
        std::memcpy(&bits, &r.p[i], 4);
        if (clear_fold && i == kFoldParam) bits = 0;
        std::snprintf(buf, sizeof buf, "0x%08xu", bits);
        o << (i ? ", " : "") << "__builtin_bit_cast(float, " << buf << ")";
    }
    o << "}}";
    return o.str();
}

// a float constant of the generated source, bit for bit
inline std::string flit(float v)
{
    uint32_t bits;
    std::memcpy(&bits, &v, 4);
    char buf[48];
    std::snprintf(buf, sizeof buf, "__builtin_bit_cast(float, 0x%08xu)", bits);
    return buf;
}

inline uint32_t fold_of(const Rec& r)
{
    uint32_t f;
    std::memcpy(&f, &r.p[kFoldParam], 4);
    return f;
}

enum Kind { POINT, RESULT };
enum Role { LEAF, UNARY, WITH_POINT, SELECT, POINT_OP };

struct Node {
    Kind kind;
    Role role;
    int rec;        // index of the record that produces it
    uint32_t op;
    int a = -1;     // operand in `last` (POINT_OP / UNARY / WITH_POINT / SELECT: first operand; LEAF: its point)
    int b = -1;     // register operand (WITH_POINT: a point; SELECT: a result)
};

inline bool is_leaf(uint32_t op)
{
    switch (op) {
    case OP_RECTANGLE: case OP_CIRCLE: case OP_REGULAR_POLYGON2D: case OP_POLYGON2D: case OP_SPHERE: case OP_HALF_SPACE:
    case OP_INVOLUTE_GEAR:
        return true;
    default: return false;
    }
}
inline bool is_unary_result(uint32_t op)
{
    switch (op) {
    case OP_TRANSFORMATION_FROM: case OPX_FROM_SCALE: case OPX_FROM_AXIS_X: case OPX_FROM_AXIS_Y: case OPX_FROM_AXIS_Z:
    case OPX_FROM_MATRIX: case OP_OFFSET: case OP_SHELL:
        return true;
    default: return false;
    }
}
inline bool is_select(uint32_t op) { return op == OP_UNION || op == OP_INTERSECTION || op == OP_SUBTRACTION; }
// ops whose direction reads the distance that entered them: phase 1 keeps that distance for phase 2
inline bool reads_input_distance(uint32_t op) { return op == OP_SHELL || op == OP_EXTRUSION; }

// Symbolic execution of the full program: which value is where.  false: a shape this generator does not defer.
inline bool build_graph(const std::vector<Rec>& recs, std::vector<Node>& nodes, int& root, std::vector<int>* rec_node = nullptr)
{
    std::vector<int> slot(256, -1);
    int last = -1;
    root = -1;
    if (rec_node) rec_node->assign(recs.size(), -1);   // the node a record produces; for _store / _load: the node it moves
    for (int i = 0; i < (int)recs.size(); ++i) {
        const Rec& r = recs[i];
        const uint32_t op = r.hdr & 0xffu, reg = (r.hdr >> 8) & 0xffffu, fold = fold_of(r);
        if (reg >= 256u) return false;
        if (fold & kFoldLoad) last = slot[fold & 0xffu];
        Node n;
        n.rec = i;
        n.op = op;
        bool made = true;
        if (op == OP_RETURN) { root = last; break; }
        else if (op == OP_STORE) { slot[reg] = last; made = false; }
        else if (op == OP_LOAD) { last = slot[reg]; made = false; }
        else if (op == OPX_POINT || op == OP_INITIAL_TRANSFORMATION_TO || op == OPX_INIT_ROW_X) { n.kind = POINT; n.role = POINT_OP; }
        else if (op == OPX_INIT_ROWS_YZ) {   // reads the sample point and the x' its first half parked in `last`
            if (last < 0 || nodes[last].op != OPX_INIT_ROW_X) return false;
            n.kind = POINT; n.role = POINT_OP; n.a = last;
        }
        else if (produces_point(op)) {
            if (last < 0 || nodes[last].kind != POINT) return false;
            n.kind = POINT; n.role = POINT_OP; n.a = last;
        }
        else if (op == OP_MIRROR) {
            if (last < 0) return false;
            n.kind = nodes[last].kind; n.role = n.kind == POINT ? POINT_OP : UNARY; n.a = last;
        }
        else if (is_leaf(op)) {
            if (last < 0 || nodes[last].kind != POINT) return false;
            n.kind = RESULT; n.role = LEAF; n.a = last;
        }
        else if (is_unary_result(op)) {
            if (last < 0 || nodes[last].kind != RESULT) return false;
            n.kind = RESULT; n.role = UNARY; n.a = last;
        }
        else if (reads_point_operand(op)) {
            if (last < 0 || nodes[last].kind != RESULT || slot[reg] < 0 || nodes[slot[reg]].kind != POINT) return false;
            n.kind = RESULT; n.role = WITH_POINT; n.a = last; n.b = slot[reg];
        }
        else if (is_select(op)) {
            if (r.p[0] >= 0.0f) return false;   // a rounded blend: the distance depends on directions
            if (last < 0 || nodes[last].kind != RESULT || slot[reg] < 0 || nodes[slot[reg]].kind != RESULT) return false;
            n.kind = RESULT; n.role = SELECT; n.a = last; n.b = slot[reg];
        }
        else return false;
        if (made) {
            nodes.push_back(n);
            last = (int)nodes.size() - 1;
        }
        if (rec_node) (*rec_node)[i] = last;
        if (fold & kFoldStore) slot[(fold >> 16) & 0xffu] = last;
    }
    return root >= 0 && nodes[root].kind == RESULT;
}

struct Step { int node; int negate; };   // an op on the way up, or (node == -1) a bare negation

struct Path {
    int leaf;
    std::vector<std::pair<int, bool>> choices;   // (select's record, taken when the comparison was true)
    std::vector<Step> up;                        // from the leaf towards the root
};

inline bool collect_paths(const std::vector<Node>& nodes, int at, Path cur, std::vector<Path>& out, size_t limit)
{
    const Node& n = nodes[at];
    switch (n.role) {
    case LEAF:
        cur.leaf = at;
        out.push_back(cur);
        return out.size() <= limit;
    case UNARY:
    case WITH_POINT:
        cur.up.insert(cur.up.begin(), Step{at, 0});
        return collect_paths(nodes, n.a, cur, out, limit);
    case SELECT: {
        // rounded_union(r < 0): the direction of `a` where a.w < b.w, else of `b`;
        //   union(last, reg)         a = last,  b = reg
        //   intersection(last, reg)  -(a' or b') with a' = -last, b' = -reg: the two negations cancel exactly
        //   subtraction(last, reg)   -(a' or b) with a' = -last:  last's direction as it is, reg's negated
        Path pa = cur, pb = cur;
        pa.choices.push_back({n.rec, true});
        pb.choices.push_back({n.rec, false});
        if (n.op == OP_SUBTRACTION) pb.up.insert(pb.up.begin(), Step{-1, 1});
        return collect_paths(nodes, n.a, pa, out, limit) && collect_paths(nodes, n.b, pb, out, limit);
    }
    default: return false;
    }
}

}  // namespace spec_detail

// The straight-line full program (the only form of round 1; still the form of every tape the deferral does not cover).
inline void emit_plain(std::ostringstream& o, const SpecProgram& p)
{
    using namespace spec_detail;
    o << "constexpr int kHoisted = 0;\n"
      << "template <class T, int PRE, uint32_t AXIS = 4u> __device__ __forceinline__ sdf::V4<T> tape_eval(T px, T py, T pz, const float* __restrict__ extra, T*)\n{\n"
      << "    using namespace sdf;\n    RegsV<T, " << p.n_slots << "> regs;\n"
      << "    V4<T> last = v4<T>(bc<T>(0.0f), bc<T>(0.0f), bc<T>(0.0f), bc<T>(0.0f));\n";
    for (const Rec& r : p.full) {
        if ((r.hdr & 0xffu) == OP_RETURN) break;
        o << "    { const Rec r = " << rec_literal(r, false, r.hdr) << "; exec_one<T, false, decltype(regs), " << (r.hdr & 0xffu)
          << ">(r, last, extra, px, py, pz, regs); }\n";
    }
    o << "    return last;\n}\n"
      << "template <class T, int PRE, uint32_t AXIS = 4u> __device__ __forceinline__ T tape_dist(T px, T py, T pz, const float* __restrict__ extra, T*)\n"
      << "{ return tape_eval<T, 0>(px, py, pz, extra, nullptr).w; }\n";
}

// true: the deferred form was emitted; false: nothing was written (use emit_plain)
inline bool emit_deferred(std::ostringstream& o, const SpecProgram& p, size_t max_paths = 40, bool save_points = false)
{
    using namespace spec_detail;
    if (p.dist.empty() || p.dist.size() != p.full.size()) return false;
    std::vector<Node> nodes;
    std::vector<int> rec_node;
    int root;
    if (!build_graph(p.full, nodes, root, &rec_node)) return false;
    std::vector<Path> paths;
    if (!collect_paths(nodes, root, Path(), paths, max_paths)) return false;
    if (paths.size() < 2) return false;   // a single primitive: nothing to defer

    // ---- phase 1: the distance-only program; `keep` = what phase 2 wants from it
    std::vector<char> is_choice(p.full.size(), 0), keep_w(p.full.size(), 0);
    for (const Path& path : paths) {
        for (auto& c : path.choices) is_choice[c.first] = 1;
        for (const Step& s : path.up)
            if (s.node >= 0 && reads_input_distance(nodes[s.node].op)) keep_w[nodes[s.node].rec] = 1;
    }
    // save_points: keep each primitive's local coordinates from phase 1 instead of recomputing them in phase 2
    // (registers against instructions: measured per tape family)
    std::vector<char> keep_pt(p.full.size(), 0);
    if (save_points)
        for (const Path& path : paths) {
            keep_pt[nodes[nodes[path.leaf].a].rec] = 1;
            for (const Step& s : path.up)
                if (s.node >= 0 && nodes[s.node].role == WITH_POINT) keep_pt[nodes[nodes[s.node].b].rec] = 1;
        }
    // ---- what does not change along z.  The brick kernels walk a wavefront's bricks along z with x and y fixed
    // (kernels.hpp), and a primitive whose distance only reads x and y -- the bar of a cross that runs along z -- is the
    // same in all of them (the leaf-block kernel walks along x: the same for the bar along x, AXIS): its records are
    // hoisted out of that loop (PRE == 1: phase 1 once with z = 0, the hoisted
    // distances captured -- everything else is dead code there; PRE == 2: those records replaced by the captured
    // distance; PRE == 0: everything in place).  A hoisted run is a private chain `to ... primitive from ...` of
    // scalings, quarter turns (which permute the coordinates without touching them: interp.hpp axis_rotate with B == 0)
    // and a rectangle or a circle (which read |x|, |y| or x^2 + y^2: no sign of a zero can differ).
    std::vector<int> run_first, run_last, run_of(p.dist.size(), -1);
    std::vector<uint32_t> run_free;   // bit a: the run's distance does not read sample coordinate a (x, y, z): free along a walk in that direction
    {
        enum : uint8_t { X = 1, Y = 2, Z = 4 };
        struct Deps { uint8_t c[3]; };
        const Deps unknown{{X | Y | Z, X | Y | Z, X | Y | Z}};
        Deps last_d = unknown;
        std::vector<Deps> slot_d(256, unknown);
        const int n = (int)p.dist.size();
        std::vector<Deps> before(n, unknown);   // of `last` when the record's own operation starts (after its folded load)
        std::vector<uint8_t> w_deps(n, X | Y | Z);
        for (int i = 0; i < n; ++i) {
            const Rec& r = p.dist[i];
            const uint32_t op = r.hdr & 0xffu, fold = fold_of(r);
            if (op == OP_RETURN) break;
            if (fold & kFoldLoad) last_d = (fold & kFoldLoadResult) ? unknown : slot_d[fold & 0xffu];
            before[i] = last_d;
            Deps out = unknown;
            uint8_t w = X | Y | Z;
            const bool quarter = r.p[1] == 0.0f;
            switch (op) {
            case OPX_POINT: out = Deps{{X, Y, Z}}; break;
            case OPX_TO_SCALE: case OP_REPETITION: out = last_d; break;
            case OPX_TO_AXIS_X: if (quarter) out = Deps{{last_d.c[0], last_d.c[2], last_d.c[1]}}; break;
            case OPX_TO_AXIS_Y: if (quarter) out = Deps{{last_d.c[2], last_d.c[1], last_d.c[0]}}; break;
            case OPX_TO_AXIS_Z: if (quarter) out = Deps{{last_d.c[1], last_d.c[0], last_d.c[2]}}; break;
            case OP_RECTANGLE: case OP_CIRCLE: w = last_d.c[0] | last_d.c[1]; break;
            case OPX_FROM_SCALE: case OPX_FROM_AXIS_X: case OPX_FROM_AXIS_Y: case OPX_FROM_AXIS_Z:
                w = i > 0 ? w_deps[i - 1] : (X | Y | Z);   // (only meaningful inside a run, where the previous record made the distance)
                break;
            default: break;
            }
            w_deps[i] = w;
            last_d = out;
            if ((fold & kFoldStore) && !(fold & kFoldStoreResult)) slot_d[(fold >> 16) & 0xffu] = last_d;
            if (op == OP_STORE && !(r.hdr & kResultKind)) slot_d[(r.hdr >> 8) & 0xffu] = before[i];
        }
        auto in_run = [&](uint32_t op, const Rec& r) {
            if (op == OPX_TO_SCALE || op == OP_RECTANGLE || op == OP_CIRCLE || op == OPX_FROM_SCALE || op == OPX_FROM_AXIS_X ||
                op == OPX_FROM_AXIS_Y || op == OPX_FROM_AXIS_Z) return true;
            return (op == OPX_TO_AXIS_X || op == OPX_TO_AXIS_Y || op == OPX_TO_AXIS_Z) && r.p[1] == 0.0f;
        };
        for (int a = 0; a < n;) {
            const uint32_t op_a = p.dist[a].hdr & 0xffu;
            if (op_a == OP_RETURN) break;
            if (!in_run(op_a, p.dist[a]) || (fold_of(p.dist[a]) & kFoldLoadResult)) { ++a; continue; }
            int b = a, prims = 0, prim_at = -1;
            for (int i = a; i < n; ++i) {
                const Rec& r = p.dist[i];
                const uint32_t op = r.hdr & 0xffu, fold = fold_of(r);
                if (!in_run(op, r) || keep_w[i] || is_choice[i] || keep_pt[i]) break;
                if (i > a && (fold & kFoldLoad)) break;
                const bool is_prim = op == OP_RECTANGLE || op == OP_CIRCLE;
                const bool is_from = op == OPX_FROM_SCALE || op == OPX_FROM_AXIS_X || op == OPX_FROM_AXIS_Y || op == OPX_FROM_AXIS_Z;
                if (is_prim && prims) break;
                if (is_from && !prims) break;            // a direction's transformation ahead of any primitive: not this pattern
                if (!is_prim && !is_from && prims) break;   // a point operation after the primitive: the next leaf starts
                if (is_prim) { ++prims; prim_at = i; }
                b = i;
                if (fold & kFoldStore) break;            // the value leaves `last`: the run ends here
            }
            // a run must hold its primitive, keep every point it computes to itself, and end in a distance free of z
            bool ok = prims == 1 && b >= prim_at;
            for (int i = a; ok && i < b; ++i) ok = !(fold_of(p.dist[i]) & kFoldStore);
            if (ok && (fold_of(p.dist[b]) & kFoldStore)) ok = (fold_of(p.dist[b]) & kFoldStoreResult) != 0;
            // (walks are along z -- the dense kernel -- or along x -- the leaf-block kernel: kernels.hpp)
            const uint32_t free_along = (uint32_t)(~w_deps[b]) & (X | Z);
            if (ok) ok = free_along != 0u;
            if (ok) {
                for (int i = a; i <= b; ++i) run_of[i] = (int)run_first.size();
                run_first.push_back(a);
                run_last.push_back(b);
                run_free.push_back(free_along);
                a = b + 1;
            } else {
                ++a;
            }
        }
    }
    // ... and a rectangle that is not free of the walk's coordinate as a whole may still be in ONE of its two: the bars of
    // a cross that do not run along the walk.  Its `|x| - h` (or `|y| - h`) for that coordinate is hoisted the same way
    // (half[i]: per walk direction, which of the two, and under which number).
    struct Half { int which[3] = {-1, -1, -1}; int number[3] = {-1, -1, -1}; };   // index: 0 walk along x, 2 along z
    std::vector<Half> half(p.dist.size());
    int n_hoisted = (int)run_first.size();
    static const bool halves = [] { const char* e = getenv("HU_HOIST_HALVES"); return !(e && e[0] == '0'); }();
    if (halves) {
        // (the analysis above, once more, for what it did not keep: the components' dependencies where a rectangle starts)
        enum : uint8_t { X = 1, Y = 2, Z = 4 };
        struct Deps { uint8_t c[3]; };
        const Deps unknown{{X | Y | Z, X | Y | Z, X | Y | Z}};
        Deps last_d = unknown;
        std::vector<Deps> slot_d(256, unknown);
        for (int i = 0; i < (int)p.dist.size(); ++i) {
            const Rec& r = p.dist[i];
            const uint32_t op = r.hdr & 0xffu, fold = fold_of(r);
            if (op == OP_RETURN) break;
            if (fold & kFoldLoad) last_d = (fold & kFoldLoadResult) ? unknown : slot_d[fold & 0xffu];
            const Deps in = last_d;
            Deps out = unknown;
            const bool quarter = r.p[1] == 0.0f;
            switch (op) {
            case OPX_POINT: out = Deps{{X, Y, Z}}; break;
            case OPX_TO_SCALE: case OP_REPETITION: out = in; break;
            case OPX_TO_AXIS_X: if (quarter) out = Deps{{in.c[0], in.c[2], in.c[1]}}; break;
            case OPX_TO_AXIS_Y: if (quarter) out = Deps{{in.c[2], in.c[1], in.c[0]}}; break;
            case OPX_TO_AXIS_Z: if (quarter) out = Deps{{in.c[1], in.c[0], in.c[2]}}; break;
            case OP_RECTANGLE:
                // (only for the walk along x, the leaf-block kernel's: 0.52 -> 0.495 ms for the bench's blocks; in the dense
                // kernel's walk along z the eight extra registers cost more than the subtractions saved: 0.757 -> 0.782 ms)
                if (run_of[i] < 0 && !keep_w[i] && !keep_pt[i] && !is_choice[i])
                    for (int axis : {0}) {
                        const uint8_t bit = (uint8_t)(1u << axis);
                        const bool free0 = !(in.c[0] & bit), free1 = !(in.c[1] & bit);
                        if (free0 != free1) {   // (both: a whole run above; neither: nothing to hoist)
                            half[i].which[axis] = free0 ? 0 : 1;
                            half[i].number[axis] = n_hoisted++;
                        }
                    }
                break;
            default: break;
            }
            last_d = out;
            if ((fold & kFoldStore) && !(fold & kFoldStoreResult)) slot_d[(fold >> 16) & 0xffu] = last_d;
            if (op == OP_STORE && !(r.hdr & kResultKind)) slot_d[(r.hdr >> 8) & 0xffu] = in;
        }
    }
    std::ostringstream body;
    body << "    using namespace sdf;\n    using M = typename mask_of<T>::type;\n"
         << "    RegsDO<T, " << p.n_point_slots << ", " << p.n_result_slots << "> regs;\n"
         << "    V4<T> last = v4<T>(bc<T>(0.0f), bc<T>(0.0f), bc<T>(0.0f), bc<T>(0.0f));\n";
    for (int i = 0; i < (int)p.dist.size(); ++i) {   // what phase 2 reads is declared up front: records may sit inside branches
        if ((p.dist[i].hdr & 0xffu) == OP_RETURN) break;
        if (keep_w[i]) body << "    T w" << i << " = bc<T>(0.0f);\n";
        if (is_choice[i]) body << "    M c" << i << " = ~mask_of<T>::all();\n";
        if (keep_pt[i]) body << "    V4<T> pt" << i << " = last;\n";
    }
    for (int i = 0; i < (int)p.dist.size(); ++i) {
        const Rec& r = p.dist[i];
        const uint32_t op = r.hdr & 0xffu, slot = (r.hdr >> 8) & 0xffffu, fold = fold_of(r);
        if (op == OP_RETURN) break;
        const int run = run_of[i];
        // (AXIS: the direction of the walk, as a bit -- 1 x, 4 z; a run is hoisted in the instantiations whose walk it is free along)
        if (run >= 0 && run_first[run] == i)
            body << "    if constexpr (!(PRE == 2 && (" << run_free[run] << "u & AXIS))) {   // (hoisted out of walks along " << ((run_free[run] & 1u) ? "x " : "") << ((run_free[run] & 4u) ? "z" : "") << ")\n";
        if (fold & kFoldLoad) {
            if (fold & kFoldLoadResult) body << "    last.w = regs.load_res(" << (fold & 0xffu) << ");\n";
            else body << "    last = regs.load(" << (fold & 0xffu) << ");\n";
        }
        if (keep_w[i]) body << "    w" << i << " = last.w;\n";
        const std::string run_text = "{ const Rec r = " + rec_literal(r, true, r.hdr) + "; exec_one<T, true, decltype(regs), " + std::to_string(op) +
                                ">(r, last, extra, px, py, pz, regs); }";
        if (is_select(op)) {
            // the comparison of rounded_union(r < 0) for this op, on the operands it would have seen
            const char* a = op == OP_UNION ? "last.w" : "-last.w";
            const std::string b_raw = "regs.load_res(" + std::to_string(slot) + ")";
            const std::string b = op == OP_INTERSECTION ? "-" + b_raw : b_raw;
            const std::string choice = is_choice[i] ? "c" + std::to_string(i) + " = " : std::string();
            {
                if (is_choice[i]) body << "    " << choice << "lt(" << a << ", " << b << ");\n";
                body << "    " << run_text << "\n";
            }
        } else if (op == OP_RECTANGLE && (half[i].number[0] >= 0 || half[i].number[2] >= 0)) {
            // exec_one's distance-only rectangle, last.w = perp_w(|x| - hw, |y| - hh), with the term that the walk does
            // not change taken from (PRE == 2) or left in (PRE == 1) the hoisted values
            body << "    {\n        T ax, ay;\n";
            for (int c = 0; c < 2; ++c) {
                const char* name = c == 0 ? "ax" : "ay";
                const std::string plain = std::string(name) + " = abs_minus(last." + (c == 0 ? "x" : "y") + ", " + flit(r.p[c]) + ");";
                std::string cond;   // the instantiations in which THIS term is hoisted
                for (int axis : {0, 2})
                    if (half[i].which[axis] == c)
                        cond += (cond.empty() ? "" : " || ") + std::string("(AXIS == ") + std::to_string(1u << axis) + "u)";
                if (cond.empty()) { body << "        " << plain << "\n"; continue; }
                // (each term is hoisted along at most one of the two walks, so one number per term and instantiation)
                const int number = half[i].which[0] == c ? half[i].number[0] : half[i].number[2];
                const int number2 = (half[i].which[0] == c && half[i].which[2] == c) ? half[i].number[2] : number;
                body << "        if constexpr (PRE == 2 && (" << cond << ")) " << name << " = hoisted[AXIS == 1u ? " << number << " : " << number2 << "];\n"
                     << "        else {\n            " << plain << "\n"
                     << "            if constexpr (PRE == 1 && (" << cond << ")) hoisted[AXIS == 1u ? " << number << " : " << number2 << "] = " << name << ";\n        }\n";
            }
            body << "        last.w = perp_w<T>(ax, ay);\n    }\n";
        } else {
            body << "    " << run_text << "\n";
        }
        if (keep_pt[i]) body << "    pt" << i << " = last;\n";
        if (run >= 0 && run_last[run] == i)
            body << "    if constexpr (PRE == 1 && (" << run_free[run] << "u & AXIS)) hoisted[" << run << "] = last.w;\n    } else {\n    last.w = hoisted[" << run << "];\n    }\n";
        if (fold & kFoldStore) {
            if (fold & kFoldStoreResult) body << "    regs.store_res(" << ((fold >> 16) & 0xffu) << ", last.w);\n";
            else body << "    regs.store(" << ((fold >> 16) & 0xffu) << ", last);\n";
        }
    }
    o << "constexpr int kHoisted = " << n_hoisted << ";   // distances that do not change along z (emit_deferred)\n"
      << "template <class T, int PRE, uint32_t AXIS = 4u> __device__ __forceinline__ T tape_dist(T px, T py, T pz, const float* __restrict__ extra, T* hoisted)\n{\n";
    {   // the distance alone: phase 1 without the captures (they are dead there)
        std::string text = body.str();
        o << text << "    return last.w;\n}\n";
    }
    o << "// deferred directions: " << paths.size() << " (primitive, path) pairs" << "\n"
      << "template <class T, int PRE, uint32_t AXIS = 4u> __device__ __forceinline__ sdf::V4<T> tape_eval(T px, T py, T pz, const float* __restrict__ extra, T* hoisted)\n{\n"
      << body.str()
      << "    const T w_root = last.w;\n"
      << "    // ---- phase 2\n"
      << "    const T qx = opaque(px), qy = opaque(py), qz = opaque(pz);\n"
      << "    V4<T> dir = v4<T>(bc<T>(0.0f), bc<T>(0.0f), bc<T>(0.0f), bc<T>(0.0f));\n"
      << "    RegsOne<T> one;\n";
    const std::string zero4 = "v4<T>(bc<T>(0.0f), bc<T>(0.0f), bc<T>(0.0f), bc<T>(0.0f))";
    auto run_record = [&](const Rec& r, const std::string& value) {   // one record of the FULL program on `value`
        const uint32_t op = r.hdr & 0xffu;
        o << "{ const Rec r = " << rec_literal(r, true, op) << "; exec_one<T, false, RegsOne<T>, " << op << ">(r, " << value
          << ", extra, qx, qy, qz, one); }";
    };
    for (const Path& path : paths) {
        o << "    {   // the primitive of record " << nodes[path.leaf].rec << " along one path to the root\n        const M m = ";
        if (path.choices.empty()) o << "mask_of<T>::all()";
        for (size_t k = 0; k < path.choices.size(); ++k)
            o << (k ? " & " : "") << (path.choices[k].second ? "c" : "~c") << path.choices[k].first;
        o << ";\n        if (wave_any(m)) {\n#ifndef SDF_PHASE2_ALL_LANES\n            one.act = m;\n#endif\n";
        // the local coordinates of a point node: its chain of point ops from the sample point, emitted once per block
        std::vector<int> var(nodes.size(), -1);
        int next_var = 0;
        auto emit_point = [&](int node) -> std::string {
            if (keep_pt[nodes[node].rec]) return "pt" + std::to_string(nodes[node].rec);
            std::vector<int> chain;
            for (int at = node; at >= 0 && var[at] < 0; at = nodes[at].a) chain.push_back(at);
            for (auto it = chain.rbegin(); it != chain.rend(); ++it) {
                const Node& n = nodes[*it];
                var[*it] = next_var++;
                o << "            V4<T> p" << var[*it] << " = " << (n.a >= 0 ? "p" + std::to_string(var[n.a]) : zero4) << "; ";
                run_record(p.full[n.rec], "p" + std::to_string(var[*it]));
                o << "\n";
            }
            return "p" + std::to_string(var[node]);
        };
        const Node& leaf = nodes[path.leaf];
        const std::string at = emit_point(leaf.a);
        o << "            V4<T> d = " << at << "; ";
        run_record(p.full[leaf.rec], "d");
        o << "\n";
        for (const Step& st : path.up) {
            if (st.node < 0) {
                o << "            d = v4<T>(-d.x, -d.y, -d.z, d.w);\n";
                continue;
            }
            const Node& n = nodes[st.node];
            if (n.role == WITH_POINT) {
                const std::string operand = emit_point(n.b);   // (emits the chain's statements first)
                o << "            one.v = " << operand << ";\n";
            }
            if (reads_input_distance(n.op)) o << "            d.w = w" << n.rec << ";\n";
            o << "            ";
            run_record(p.full[n.rec], "d");
            o << "\n";
        }
        o << "            dir.x = sel(m, d.x, dir.x); dir.y = sel(m, d.y, dir.y); dir.z = sel(m, d.z, dir.z);\n"
          << "        }\n    }\n";
    }
    o << "    return v4<T>(dir.x, dir.y, dir.z, w_root);\n}\n";
    return true;
}

// The whole translation unit handed to hipRTC.  `deferred` (may be NULL) <- whether the deferred form was used.
inline std::string specialised_source(const SpecProgram& p, bool allow_deferred, bool* deferred = nullptr)
{
    std::ostringstream o;
    // HU_ABS_BUILTIN=1 (an experiment that lost): |x| - h written plainly in the brick kernels, so that the compiler hoists
    // the terms in x and y out of the loop over a wavefront's bricks (interp.hpp abs_minus) -- it does, and the hoisted
    // values cost 30 registers (85 -> 115, a wavefront less per SIMD): dense 0.83 -> 1.12 ms, leaf blocks 0.58 -> 0.92 ms
    static const bool abs_builtin = [] { const char* e = getenv("HU_ABS_BUILTIN"); return e && e[0] == '1'; }();
    std::ostringstream d;
    static const bool save_points = [] { const char* e = getenv("HU_PHASE2_SAVE_POINTS"); return e && e[0] == '1'; }();
    const bool ok = allow_deferred && emit_deferred(d, p, 40, save_points);
    o << (ok && abs_builtin ? "#define SDF_ABS_MINUS_BUILTIN 1\n" : "") << "#include \"kernels.hpp\"\nnamespace sdfk {\nusing sdf::Rec;\n";
    if (ok) o << d.str();
    else emit_plain(o, p);
    if (deferred) *deferred = ok;
    o << "struct JitEval {\n    static constexpr bool kBricks = " << (ok ? "true" : "false") << ";\n"
      << "    const float* extra;\n"
      << "    template <class T> __device__ __forceinline__ sdf::V4<T> operator()(T px, T py, T pz, void*) const\n"
      << "    { return tape_eval<T, 0>(px, py, pz, extra, nullptr); }\n"
      << "    template <class T> __device__ __forceinline__ T dist(T px, T py, T pz, void*) const\n"
      << "    { return tape_dist<T, 0>(px, py, pz, extra, nullptr); }\n"
      // what does not change along z, for kernels that walk bricks along z with x and y fixed (kernels.hpp)
      << "    template <class T> struct Hoisted { T v[kHoisted > 0 ? kHoisted : 1]; };\n"
      << "    template <class T> __device__ __forceinline__ Hoisted<T> hoist(T px, T py) const\n"
      << "    { Hoisted<T> h; if constexpr (kHoisted > 0) tape_dist<T, 1, 4u>(px, py, sdf::bc<T>(0.0f), extra, h.v); return h; }\n"
      << "    template <class T> __device__ __forceinline__ sdf::V4<T> eval_hoisted(T px, T py, T pz, Hoisted<T>& h) const\n"
      << "    { return tape_eval<T, (kHoisted > 0 ? 2 : 0), 4u>(px, py, pz, extra, h.v); }\n"
      << "    template <class T> __device__ __forceinline__ T dist_hoisted(T px, T py, T pz, Hoisted<T>& h) const\n"
      << "    { return tape_dist<T, (kHoisted > 0 ? 2 : 0), 4u>(px, py, pz, extra, h.v); }\n"
      // ... and the same for a walk along x with y and z fixed (k_grid_eval_blocks)
      << "    template <class T> __device__ __forceinline__ Hoisted<T> hoist_x(T py, T pz) const\n"
      << "    { Hoisted<T> h; if constexpr (kHoisted > 0) tape_dist<T, 1, 1u>(sdf::bc<T>(0.0f), py, pz, extra, h.v); return h; }\n"
      << "    template <class T> __device__ __forceinline__ sdf::V4<T> eval_hoisted_x(T px, T py, T pz, Hoisted<T>& h) const\n"
      << "    { return tape_eval<T, (kHoisted > 0 ? 2 : 0), 1u>(px, py, pz, extra, h.v); }\n"
      << "    template <class T> __device__ __forceinline__ T dist_hoisted_x(T px, T py, T pz, Hoisted<T>& h) const\n"
      << "    { return tape_dist<T, (kHoisted > 0 ? 2 : 0), 1u>(px, py, pz, extra, h.v); }\n"
      << "};\n}  // namespace sdfk\n";
    return o.str();
}

}  // namespace sdf
