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
inline bool build_graph(const std::vector<Rec>& recs, std::vector<Node>& nodes, int& root)
{
    std::vector<int> slot(256, -1);
    int last = -1;
    root = -1;
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
    o << "template <class T> __device__ __forceinline__ sdf::V4<T> tape_eval(T px, T py, T pz, const float* __restrict__ extra)\n{\n"
      << "    using namespace sdf;\n    RegsV<T, " << p.n_slots << "> regs;\n"
      << "    V4<T> last = v4<T>(bc<T>(0.0f), bc<T>(0.0f), bc<T>(0.0f), bc<T>(0.0f));\n";
    for (const Rec& r : p.full) {
        if ((r.hdr & 0xffu) == OP_RETURN) break;
        o << "    { const Rec r = " << rec_literal(r, false, r.hdr) << "; exec_one<T, false, decltype(regs), " << (r.hdr & 0xffu)
          << ">(r, last, extra, px, py, pz, regs); }\n";
    }
    o << "    return last;\n}\n"
      << "template <class T> __device__ __forceinline__ T tape_dist(T px, T py, T pz, const float* __restrict__ extra)\n"
      << "{ return tape_eval<T>(px, py, pz, extra).w; }\n";
}

// true: the deferred form was emitted; false: nothing was written (use emit_plain)
inline bool emit_deferred(std::ostringstream& o, const SpecProgram& p, size_t max_paths = 40, bool save_points = false)
{
    using namespace spec_detail;
    if (p.dist.empty() || p.dist.size() != p.full.size()) return false;
    std::vector<Node> nodes;
    int root;
    if (!build_graph(p.full, nodes, root)) return false;
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
    std::ostringstream body;
    body << "    using namespace sdf;\n    using M = typename mask_of<T>::type;\n"
         << "    RegsDO<T, " << p.n_point_slots << ", " << p.n_result_slots << "> regs;\n"
         << "    V4<T> last = v4<T>(bc<T>(0.0f), bc<T>(0.0f), bc<T>(0.0f), bc<T>(0.0f));\n";
    for (int i = 0; i < (int)p.dist.size(); ++i) {
        const Rec& r = p.dist[i];
        const uint32_t op = r.hdr & 0xffu, slot = (r.hdr >> 8) & 0xffffu, fold = fold_of(r);
        if (op == OP_RETURN) break;
        if (fold & kFoldLoad) {
            if (fold & kFoldLoadResult) body << "    last.w = regs.load_res(" << (fold & 0xffu) << ");\n";
            else body << "    last = regs.load(" << (fold & 0xffu) << ");\n";
        }
        if (keep_w[i]) body << "    const T w" << i << " = last.w;\n";
        if (is_choice[i]) {
            // the comparison of rounded_union(r < 0) for this op, on the operands it would have seen
            const char* a = op == OP_UNION ? "last.w" : "-last.w";
            std::string b = "regs.load_res(" + std::to_string(slot) + ")";
            if (op == OP_INTERSECTION) b = "-" + b;
            body << "    const M c" << i << " = lt(" << a << ", " << b << ");\n";
        }
        body << "    { const Rec r = " << rec_literal(r, true, r.hdr) << "; exec_one<T, true, decltype(regs), " << op
             << ">(r, last, extra, px, py, pz, regs); }\n";
        if (keep_pt[i]) body << "    const V4<T> pt" << i << " = last;\n";
        if (fold & kFoldStore) {
            if (fold & kFoldStoreResult) body << "    regs.store_res(" << ((fold >> 16) & 0xffu) << ", last.w);\n";
            else body << "    regs.store(" << ((fold >> 16) & 0xffu) << ", last);\n";
        }
    }
    o << "template <class T> __device__ __forceinline__ T tape_dist(T px, T py, T pz, const float* __restrict__ extra)\n{\n";
    {   // the distance alone: phase 1 without the captures (they are dead there)
        std::string text = body.str();
        o << text << "    return last.w;\n}\n";
    }
    o << "// deferred directions: " << paths.size() << " (primitive, path) pairs\n"
      << "template <class T> __device__ __forceinline__ sdf::V4<T> tape_eval(T px, T py, T pz, const float* __restrict__ extra)\n{\n"
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
    o << "#include \"kernels.hpp\"\nnamespace sdfk {\nusing sdf::Rec;\n";
    std::ostringstream d;
    static const bool save_points = [] { const char* e = getenv("HU_PHASE2_SAVE_POINTS"); return e && e[0] == '1'; }();
    const bool ok = allow_deferred && emit_deferred(d, p, 40, save_points);
    if (ok) o << d.str();
    else emit_plain(o, p);
    if (deferred) *deferred = ok;
    o << "struct JitEval {\n    static constexpr bool kBricks = " << (ok ? "true" : "false") << ";\n    const float* extra;\n"
      << "    template <class T> __device__ __forceinline__ sdf::V4<T> operator()(T px, T py, T pz, void*) const\n"
      << "    { return tape_eval<T>(px, py, pz, extra); }\n"
      << "    template <class T> __device__ __forceinline__ T dist(T px, T py, T pz, void*) const\n"
      << "    { return tape_dist<T>(px, py, pz, extra); }\n};\n}  // namespace sdfk\n";
    return o.str();
}

}  // namespace sdf
