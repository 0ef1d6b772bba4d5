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

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <numeric>
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

namespace spec_detail {

// ---- phase 1 as a little compiler ---------------------------------------------------------------------------------
// The distance-only program is executed SYMBOLICALLY, component by component: every coordinate of every point and
// every distance becomes one statement `const auto t<id> = <expression of earlier values>;`, with the sample
// coordinates it depends on noted beside it.  Four things fall out of that which record-by-record code (one exec_one
// call per record on a four-component value) could not give:
//   * width: `auto` takes the narrowest type -- float where the two voxels of a lane cannot differ (in the brick kernels
//     they differ in x only), f2 elsewhere (interp.hpp "values of mixed width");
//   * tables: a statement that reads one sample coordinate, or two, is evaluated once per sample (pair of samples) of a
//     workgroup's box into a table in LDS and read from there by the walks ("AXIS TABLES", "PAIR TABLES" below);
//   * hoisting: a statement that does not read the coordinate a wavefront walks along and is no table column is computed
//     once per walk (`pre`) and handed on (`h.t<id>`);
//   * sharing: equal expressions are one statement (the bars of a cross read the same |z| - h).
// The arithmetic is exec_one's, operation for operation (the statement texts below restate its distance-only cases).
struct Stmt {
    std::string text;        // "$0", "$1": the operands
    std::vector<int> ops;
    uint8_t deps = 0;        // DX | DY | DZ: the sample coordinates the value depends on
    bool mask = false;       // a predicate (wavefront masks live in scalar registers: never handed from `pre`)
    // |value| <= g * B + o when every sample coordinate is within [-B, B] (coordinate_limit below); !bounded: unknown
    double g = 0.0, o = 0.0;
    bool bounded = true;
    float abs_h = -1.0f;     // the value is |x| - abs_h
    // What box pruning knows about a DISTANCE statement ("BOX PRUNING" below): how to bound it over a box from its value at
    // the box's centre (IV_LEAF: a function of the sample point with Lipschitz constant iv_l) or from the bounds of its
    // operands iv_a, iv_b (the rest); IV_NONE: not a distance (a coordinate, a mask, ...)
    uint8_t iv = 0;
    double iv_l = 0.0;
    float iv_c = 0.0f;
    int iv_a = -1, iv_b = -1;
    std::vector<int> iv_conds;   // IV_LEAF: the bound holds in a box only where these conditions do (Phase1::conds: repetitions)
};
enum : uint8_t { DX = 1, DY = 2, DZ = 4 };
enum : uint8_t { IV_NONE = 0, IV_LEAF, IV_UNKNOWN, IV_SCALE, IV_OFFSET, IV_SHELL, IV_PERP, IV_MIN, IV_MAX, IV_MAXNEG };

struct Emitter {
    std::vector<Stmt> st;
    std::vector<int> neg_of;                 // value id -> the value it negates, or -1
    std::vector<std::pair<std::string, int>> seen;
    int add(const std::string& text, const std::vector<int>& ops, uint8_t deps = 0, bool mask = false)
    {
        std::string key = text;
        for (int o : ops) key += "|" + std::to_string(o);
        for (auto& k : seen) if (k.first == key) return k.second;
        Stmt s;
        s.text = text; s.ops = ops; s.deps = deps; s.mask = mask;
        for (int o : ops) s.deps |= st[o].deps;
        st.push_back(s);
        neg_of.push_back(-1);
        seen.emplace_back(key, (int)st.size() - 1);
        return (int)st.size() - 1;
    }
    int neg(int v)
    {
        if (neg_of[v] >= 0) return neg_of[v];
        const int r = add("-($0)", {v});
        neg_of[r] = v;
        return r;
    }
    int abs_minus(int v, float h)            // |-x| - h is |x| - h, bit for bit
    {
        if (neg_of[v] >= 0) v = neg_of[v];
        return add("abs_minus($0, " + flit(h) + ")", {v});
    }
};

inline std::string render(const Stmt& s, const std::function<std::string(int)>& name)
{
    std::string out;
    for (size_t i = 0; i < s.text.size(); ++i) {
        if (s.text[i] == '$' && i + 1 < s.text.size() && s.text[i + 1] >= '0' && s.text[i + 1] <= '9') {
            out += name(s.ops[(size_t)(s.text[i + 1] - '0')]);
            ++i;
        } else out += s.text[i];
    }
    return out;
}

struct Phase1 {
    Emitter e;
    std::vector<std::pair<int, int>> perp;    // the operands of every perp_w_x statement (coordinate_limit)
    std::vector<int> dist_of;                 // node -> value id of its distance (RESULT nodes)
    std::vector<int> choice_of_rec;           // record -> value id of its choice mask, or -1
    std::vector<int> keep_w_of_rec;           // record -> value id of the distance that entered it, or -1
    std::vector<std::pair<int, int>> choice_of_select;   // (a select's value id, the value id of its choice mask)
    // A repetition's remainder jumps at its cell boundaries, but in a box that stays inside ONE cell it is a shift: what is
    // computed from it is then Lipschitz as if the repetition were not there.  cond: "coordinate statement u (Lipschitz constant
    // lip) stays within one period (1 / inv) over the box" -- decided per box by the mask function.
    struct Cond { int u; double lip; float inv; };
    std::vector<Cond> conds;
    int root = -1;
    int px = -1, py = -1, pz = -1;
    int n_phase1 = 0;                         // statements [0, n_phase1) are phase 1's; the rest are directions (phase 2)
};

// false: an op this generator does not restate (the plain form is used instead)
inline bool symbolic_phase1(const SpecProgram& p, const std::vector<Node>& nodes, int root, const std::vector<char>& is_choice,
                            const std::vector<char>& keep_w, Phase1& out, std::vector<std::array<int, 3>>* points = nullptr)
{
    Emitter& e = out.e;
    // lip: a Lipschitz constant of the map sample point -> this point (both in the Euclidean norm), < 0: none is known (the
    // map jumps: repetitions, twists); rowx: the first row of a general matrix, parked until its other two rows arrive
    struct Pt { int c[3] = {-1, -1, -1}; int w = -1; double lip = 1.0; float rowx[3] = {0.0f, 0.0f, 0.0f}; std::vector<int> conds; };
    std::vector<Pt> pt(nodes.size());
    out.dist_of.assign(nodes.size(), -1);
    out.choice_of_rec.assign(p.full.size(), -1);
    out.keep_w_of_rec.assign(p.full.size(), -1);
    out.px = e.add("px", {}, DX);
    out.py = e.add("py", {}, DY);
    out.pz = e.add("pz", {}, DZ);
    const int zero = e.add("0.0f", {});
    // magnitude bounds (Stmt::g, o): set right after a statement is made; anything not set below is "unknown"
    auto bound = [&](int id, double g, double o, bool ok = true) {
        Stmt& s = e.st[id];
        s.g = g; s.o = o; s.bounded = ok && std::isfinite(g) && std::isfinite(o);
        return id;
    };
    auto G = [&](int v) { return e.st[v].g; };
    auto O = [&](int v) { return e.st[v].o; };
    auto OK = [&](int v) { return e.st[v].bounded; };
    auto A = [](float v) { return std::fabs((double)v); };
    for (int v : {out.px, out.py, out.pz}) bound(v, 1.0, 0.0);
    bound(zero, 0.0, 0.0);
    auto fma_c = [&](int v, float a, float b) { return bound(e.add("fma_x($0, " + flit(a) + ", " + flit(b) + ")", {v}), A(a) * G(v), A(a) * O(v) + A(b), OK(v)); };
    auto fma_v = [&](int v, float a, int acc) {
        return bound(e.add("fma_x($0, " + flit(a) + ", $1)", {v, acc}), A(a) * G(v) + G(acc), A(a) * O(v) + O(acc), OK(v) && OK(acc));
    };
    auto neg = [&](int v) { return bound(e.neg(v), G(v), O(v), OK(v)); };
    auto abs_minus = [&](int v, float h) {
        const int r = bound(e.abs_minus(v, h), G(v), O(v) + A(h), OK(v));
        e.st[r].abs_h = h;
        return r;
    };
    auto perp = [&](int a, int b) {
        const int r = bound(e.add("perp_w_x($0, $1, flags)", {a, b}), G(a) + G(b), O(a) + O(b), OK(a) && OK(b));
        out.perp.emplace_back(a, b);
        return r;
    };
    auto row = [&](const int (&c)[3], const float* q) {   // fma_x(x, q0, fma_x(y, q1, fma_x(z, q2, q3)))
        return bound(e.add("fma_x($0, " + flit(q[0]) + ", fma_x($1, " + flit(q[1]) + ", fma_x($2, " + flit(q[2]) + ", " + flit(q[3]) + ")))", {c[0], c[1], c[2]}),
                     A(q[0]) * G(c[0]) + A(q[1]) * G(c[1]) + A(q[2]) * G(c[2]), A(q[0]) * O(c[0]) + A(q[1]) * O(c[1]) + A(q[2]) * O(c[2]) + A(q[3]),
                     OK(c[0]) && OK(c[1]) && OK(c[2]));
    };
    auto unknown = [&](int id) { return bound(id, 0.0, 0.0, false); };
    // interval annotations (box pruning): a leaf with the Lipschitz constant of its point, an op on bounded operands
    auto iv_leaf = [&](int id, double lip, const std::vector<int>& conds) {
        Stmt& s = e.st[id];
        if (lip >= 0.0 && std::isfinite(lip) && s.bounded) { s.iv = IV_LEAF; s.iv_l = lip; s.iv_conds = conds; }
        else s.iv = IV_UNKNOWN;
        return id;
    };
    auto iv_op = [&](int id, uint8_t kind, int a, int b = -1, float c = 0.0f) {
        Stmt& s = e.st[id];
        if (a >= 0 && e.st[a].iv == IV_NONE) e.st[a].iv = IV_UNKNOWN;
        if (b >= 0 && e.st[b].iv == IV_NONE) e.st[b].iv = IV_UNKNOWN;
        s.iv = kind; s.iv_a = a; s.iv_b = b; s.iv_c = c;
        return id;
    };
    // the largest singular value of the 3 x 3 matrix with these rows (power iteration on M^T M; a hair above, never below)
    auto matrix_norm = [](const float* r0, const float* r1, const float* r2) {
        double m[3][3] = {{r0[0], r0[1], r0[2]}, {r1[0], r1[1], r1[2]}, {r2[0], r2[1], r2[2]}}, a[3][3];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { a[i][j] = 0.0; for (int k = 0; k < 3; ++k) a[i][j] += m[k][i] * m[k][j]; }
        double v[3] = {1.0, 0.7, 0.4}, lambda = 0.0;
        for (int it = 0; it < 200; ++it) {
            double w[3] = {0.0, 0.0, 0.0};
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) w[i] += a[i][j] * v[j];
            const double n = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
            if (!(n > 0.0)) return 0.0;
            lambda = n;
            for (int i = 0; i < 3; ++i) v[i] = w[i] / n;
        }
        const double frob = std::sqrt(a[0][0] + a[1][1] + a[2][2]);      // (an upper bound whatever the iteration did)
        return std::min(frob, std::sqrt(lambda) * (1.0 + 1e-6));
    };
    // a record of the library run on widened values (ops that are rare in CAD tapes and branchy inside: polygons, gears,
    // twists, circular repetitions): last = ($0, $1, $2, $3), its register operand = ($4, $5, $6, $7)
    auto run_record = [&](const Rec& r, const int (&last)[4], const int (&operand)[4]) {
        const uint32_t op = r.hdr & 0xffu;
        return unknown(e.add("run_record<" + std::to_string(op) + ">(" + rec_literal(r, true, op) + ", extra, v4x($0, $1, $2, $3), v4x($4, $5, $6, $7))",
                             {last[0], last[1], last[2], last[3], operand[0], operand[1], operand[2], operand[3]}));
    };
    for (int n = 0; n < (int)nodes.size(); ++n) {
        const Node& nd = nodes[n];
        const Rec& r = p.full[nd.rec];
        const float* q = r.p;
        const uint32_t op = nd.op;
        const int none4[4] = {zero, zero, zero, zero};
        if (nd.kind == POINT) {
            Pt in;
            if (nd.a >= 0) in = pt[nd.a];
            Pt o;
            o.conds = in.conds;      // (what holds for the point an op starts from holds for what it makes of it; INIT ops start afresh)
            switch (op) {
            case OPX_POINT: o.c[0] = out.px; o.c[1] = out.py; o.c[2] = out.pz; o.lip = 1.0; break;
            case OPX_TO_SCALE:
                for (int c = 0; c < 3; ++c) o.c[c] = fma_c(in.c[c], q[0], q[4 + c]);
                o.lip = in.lip < 0.0 ? -1.0 : in.lip * A(q[0]);
                break;
            case OPX_TO_AXIS_X: case OPX_TO_AXIS_Y: case OPX_TO_AXIS_Z: {
                // interp.hpp axis_rotate: (along, u, v) = the axis and the other two in cyclic order
                const int ax = op == OPX_TO_AXIS_X ? 0 : op == OPX_TO_AXIS_Y ? 1 : 2, u = (ax + 1) % 3, v = (ax + 2) % 3;
                o.c[ax] = fma_c(in.c[ax], q[0], q[4 + ax]);
                int ru = fma_c(neg(in.c[v]), q[2], q[4 + u]);
                int rv = fma_c(in.c[u], q[2], q[4 + v]);
                if (q[1] != 0.0f) {
                    ru = fma_v(in.c[u], q[1], ru);
                    rv = fma_v(in.c[v], q[1], rv);
                }
                o.c[u] = ru;
                o.c[v] = rv;
                o.lip = in.lip < 0.0 ? -1.0 : in.lip * std::max(A(q[0]), std::hypot((double)q[1], (double)q[2]));
                break;
            }
            case OPX_TO_ROW_X: case OPX_INIT_ROW_X: {
                Pt src = in;
                if (op == OPX_INIT_ROW_X) { src = Pt(); src.c[0] = out.px; src.c[1] = out.py; src.c[2] = out.pz; }
                o = src;
                o.w = row(src.c, q);
                for (int c = 0; c < 3; ++c) o.rowx[c] = q[c];
                break;
            }
            case OPX_TO_ROWS_YZ: case OPX_INIT_ROWS_YZ: {
                if (in.w < 0) return false;
                Pt src = in;
                if (op == OPX_INIT_ROWS_YZ) { src = Pt(); src.c[0] = out.px; src.c[1] = out.py; src.c[2] = out.pz; }
                o.c[0] = in.w;
                o.c[1] = row(src.c, q);
                o.c[2] = row(src.c, q + 4);
                o.lip = src.lip < 0.0 ? -1.0 : src.lip * matrix_norm(in.rowx, q, q + 4);
                o.conds = src.conds;
                break;
            }
            case OP_REPETITION:
                for (int c = 0; c < 3; ++c)   // remainder_t: inv == 0 (an infinite spacing) returns the coordinate itself
                    o.c[c] = q[3 + c] == 0.0f ? in.c[c] : bound(e.add("remainder_t($0, " + flit(q[c]) + ", " + flit(q[3 + c]) + ")", {in.c[c]}),
                                                                G(in.c[c]), O(in.c[c]), OK(in.c[c]));   // (|x - n y| <= |x|: safe for huge x too)
                // (a remainder jumps at the cell boundaries: what is computed from it is Lipschitz only in boxes that stay
                // inside one cell -- Phase1::conds)
                o.lip = in.lip;
                for (int c = 0; c < 3; ++c)
                    if (q[3 + c] != 0.0f && in.lip >= 0.0 && OK(in.c[c])) {
                        out.conds.push_back(Phase1::Cond{in.c[c], in.lip, q[3 + c]});
                        o.conds.push_back((int)out.conds.size() - 1);
                    } else if (q[3 + c] != 0.0f) o.lip = -1.0;
                break;
            case OP_MIRROR: o = in; o.c[0] = neg(in.c[0]); break;
            case OP_SYMMETRICAL_TO: o = in; o.c[0] = bound(e.add("abs_($0)", {in.c[0]}), G(in.c[0]), O(in.c[0]), OK(in.c[0])); break;
            case OP_REVOLUTION_TO:
                o.c[0] = bound(e.add("len2_x($0, $1)", {in.c[0], in.c[2]}), G(in.c[0]) + G(in.c[2]), O(in.c[0]) + O(in.c[2]), OK(in.c[0]) && OK(in.c[2]));
                o.c[1] = in.c[1];
                o.c[2] = zero;
                o.lip = in.lip;
                break;
            case OP_CIRCULAR_REPETITION_TO: case OP_TWIST_REVOLUTION_TO: {
                const int last[4] = {in.c[0], in.c[1], in.c[2], zero};
                const int v = run_record(r, last, none4);
                o.c[0] = unknown(e.add("$0.x", {v})); o.c[1] = unknown(e.add("$0.y", {v})); o.c[2] = unknown(e.add("$0.z", {v}));
                o.lip = -1.0;
                break;
            }
            default: return false;
            }
            pt[n] = o;
            if (points) (*points)[n] = {{o.c[0], o.c[1], o.c[2]}};
            continue;
        }
        // ---- results: the distance alone
        int w = -1;
        const int a = nd.a;
        switch (nd.role) {
        case LEAF: {
            const Pt& c = pt[a];
            // (these four are exact distances in their local coordinates: 1-Lipschitz there, so Lipschitz with the point's constant)
            if (op == OP_RECTANGLE) w = iv_leaf(perp(abs_minus(c.c[0], q[0]), abs_minus(c.c[1], q[1])), c.lip, c.conds);
            else if (op == OP_CIRCLE)
                w = iv_leaf(bound(e.add("len2_x($0, $1) - " + flit(q[0]), {c.c[0], c.c[1]}), G(c.c[0]) + G(c.c[1]), O(c.c[0]) + O(c.c[1]) + A(q[0]), OK(c.c[0]) && OK(c.c[1])), c.lip, c.conds);
            else if (op == OP_SPHERE)
                w = iv_leaf(bound(e.add("len3_x($0, $1, $2) - " + flit(q[0]), {c.c[0], c.c[1], c.c[2]}), G(c.c[0]) + G(c.c[1]) + G(c.c[2]),
                                  O(c.c[0]) + O(c.c[1]) + O(c.c[2]) + A(q[0]), OK(c.c[0]) && OK(c.c[1]) && OK(c.c[2])), c.lip, c.conds);
            else if (op == OP_HALF_SPACE) w = iv_leaf(neg(c.c[1]), c.lip, c.conds);
            else {   // polygons, the gear: no bound is claimed for them (the gear's distance jumps between teeth)
                // (2D primitives: they read x and y only -- handing them a zero for z keeps the statement a function of the
                // two coordinates its frame's x and y depend on, so it can be a column of a pair table)
                const int last[4] = {c.c[0], c.c[1], zero, zero};
                w = unknown(e.add("$0.w", {run_record(r, last, none4)}));
                e.st[w].iv = IV_UNKNOWN;
            }
            break;
        }
        case UNARY: {
            const int in = out.dist_of[a];
            if (keep_w[nd.rec]) out.keep_w_of_rec[nd.rec] = in;
            switch (op) {
            case OP_TRANSFORMATION_FROM: case OPX_FROM_SCALE: case OPX_FROM_AXIS_X: case OPX_FROM_AXIS_Y: case OPX_FROM_AXIS_Z:
                w = iv_op(bound(e.add("$0 * " + flit(q[5]), {in}), A(q[5]) * G(in), A(q[5]) * O(in), OK(in)), IV_SCALE, in, -1, q[5]);
                break;
            case OPX_FROM_MATRIX: w = iv_op(bound(e.add("$0 * " + flit(q[9]), {in}), A(q[9]) * G(in), A(q[9]) * O(in), OK(in)), IV_SCALE, in, -1, q[9]); break;
            case OP_OFFSET: w = iv_op(bound(e.add("$0 - " + flit(q[0]), {in}), G(in), O(in) + A(q[0]), OK(in)), IV_OFFSET, in, -1, q[0]); break;
            case OP_SHELL: w = iv_op(bound(e.add("sel(ge($0, 0.0f), $0, -($0)) - " + flit(q[0]), {in}), G(in), O(in) + A(q[0]), OK(in)), IV_SHELL, in, -1, q[0]); break;
            case OP_MIRROR: w = in; break;      // (flips the direction's x: the distance stays)
            default: return false;
            }
            break;
        }
        case WITH_POINT: {
            const int in = out.dist_of[a];
            const Pt& c = pt[nd.b];
            if (keep_w[nd.rec]) out.keep_w_of_rec[nd.rec] = in;
            if (op == OP_EXTRUSION) {
                // perp(|z| - h, w): non-decreasing in both operands; the slab's distance is Lipschitz with the point's constant
                const int slab = iv_leaf(abs_minus(c.c[2], q[0]), c.lip, c.conds);
                w = iv_op(perp(slab, in), IV_PERP, slab, in);
            }
            else if (op == OP_SYMMETRICAL_FROM || op == OP_CIRCULAR_REPETITION_FROM || op == OP_REVOLUTION_FROM) w = in;   // directions only
            else if (op == OP_TWIST_REVOLUTION_FROM) {
                const int last[4] = {zero, zero, zero, in}, operand[4] = {c.c[0], c.c[1], c.c[2], zero};
                w = unknown(e.add("$0.w", {run_record(r, last, operand)}));
                e.st[w].iv = IV_UNKNOWN;
            } else return false;
            break;
        }
        case SELECT: {
            const int x = out.dist_of[a], y = out.dist_of[nd.b];
            if (is_choice[nd.rec]) {
                // the comparison of rounded_union(r < 0) for this op, on the operands it would have seen (union: a, b;
                // intersection: -a, -b; subtraction: -a, b)
                const int ca = op == OP_UNION ? x : neg(x), cb = op == OP_INTERSECTION ? neg(y) : y;
                out.choice_of_rec[nd.rec] = e.add("lt_x($0, $1)", {ca, cb}, 0, true);
            }
            w = iv_op(bound(e.add(std::string(op == OP_UNION ? "min_x" : op == OP_INTERSECTION ? "max_x" : "max_neg_x") + "($0, $1)", {x, y}),
                            std::max(G(x), G(y)), std::max(O(x), O(y)), OK(x) && OK(y)),
                      op == OP_UNION ? IV_MIN : op == OP_INTERSECTION ? IV_MAX : IV_MAXNEG, x, y);
            if (out.choice_of_rec[nd.rec] >= 0) out.choice_of_select.emplace_back(w, out.choice_of_rec[nd.rec]);
            break;
        }
        default: return false;
        }
        out.dist_of[n] = w;
    }
    out.root = out.dist_of[root];
    return out.root >= 0;
}

// The largest |sample coordinate| B up to which every perp_w_x of the tape stays inside the fast range of sqrt_cr
// (interp.hpp kFlagInRange) in the lanes where its result is used -- both operands positive there:
//   below: one operand is |x| - h with h >= 2^-25; a positive difference of two binary32 numbers is at least half an ulp
//          of the smaller, so that operand is >= 2^-50 and the sum of squares >= 2^-100;
//   above: both operands stay below 2^49 (the bounds g * B + o of symbolic_phase1), so the sum stays below 2^100.
// 0: no such B (an op outside the analysis feeds a rectangle or an extrusion); +inf: the tape has no perp_w_x at all.
inline double coordinate_limit(const Phase1& ph)
{
    double limit = HUGE_VAL;
    const double top = std::ldexp(1.0, 49);
    for (const auto& ab : ph.perp) {
        const Stmt& a = ph.e.st[ab.first];
        const Stmt& b = ph.e.st[ab.second];
        if (!(a.abs_h >= 0x1p-25f) && !(b.abs_h >= 0x1p-25f)) return 0.0;
        for (const Stmt* s : {&a, &b}) {
            if (!s->bounded || !(s->o < top)) return 0.0;
            if (s->g > 0.0) limit = std::min(limit, (top - s->o) / s->g);
        }
    }
    return limit;
}

// One variant of phase 1 as text.  walk = 0: everything in place; walk = DX (the walks of a box, kernels.hpp box_eval):
// the statements that do not read the walk's coordinate go to `pre` (returned in a struct, one member per value the walk-dependent part reads).
struct Variant {
    std::string pre;      // body of the hoisting function ("" when nothing is hoisted)
    std::string main;     // body of the evaluation up to the root distance
    int n_hoisted = 0;
    std::vector<char> handed;     // per statement: is it a member of the struct `pre` returns
    std::vector<char> tab_read;   // per statement: the body or `pre` reads it from its axis' table (axis tables, below)
    std::vector<char> tab_main;   // ... the body does, in every brick
};
// Roughly the instructions a statement's whole tree costs (shared subexpressions count once per use: an upper bound);
// 0 for names (sample coordinates, constants), masks and the directions of phase 2.
inline std::vector<int> statement_costs(const Phase1& ph)
{
    const std::vector<Stmt>& st = ph.e.st;
    std::vector<int> cost(st.size(), 0);
    for (int i = 0; i < ph.n_phase1; ++i) {
        if (st[i].ops.empty() || st[i].mask) continue;
        const std::string& t = st[i].text;
        int c = t.compare(0, 11, "remainder_t") == 0 ? 3 : t.compare(0, 8, "perp_w_x") == 0 ? 8 : t.compare(0, 4, "len2") == 0 ? 10 :
                t.compare(0, 4, "len3") == 0 ? 12 : t.compare(0, 10, "run_record") == 0 ? 100 : t.compare(0, 2, "$0") == 0 && t.size() == 4 ? 0 : 1;
        for (int o : st[i].ops) c += cost[o];
        cost[i] = c;
    }
    return cost;
}
// Which statements are handed from `pre` when the walk is along `walk`: those that do not read its coordinate -- unless
// they are cheaper to compute again in every brick than to keep in a register for the whole walk (a value costs one or
// two VGPRs for the walk's duration, and registers decide how many wavefronts a SIMD holds): `min_cost` = the number
// of instructions (roughly) a value must save per brick to be kept.
inline std::vector<char> hoistable_set(const Phase1& ph, uint8_t walk, int min_cost)
{
    const std::vector<Stmt>& st = ph.e.st;
    const int n = (int)st.size();
    std::vector<char> out(n, 0);
    if (walk == 0) return out;
    const std::vector<int> cost = statement_costs(ph);
    for (int i = 0; i < ph.n_phase1; ++i) {   // (a direction is computed where its path's mask is known: never in `pre`)
        if (st[i].ops.empty() || st[i].mask || (st[i].deps & walk)) continue;
        out[i] = cost[i] >= min_cost;
    }
    return out;
}
// AXIS TABLES (round 3).  A workgroup of the grid kernels covers a BOX of up to 16 x 16 x 16 voxels (kernels.hpp
// box_eval), and a statement that reads ONE sample coordinate takes as many distinct values in it as that axis has
// samples: 16, not 4096.  Such statements are evaluated once per sample of their axis into a table in LDS (`tape_tab_x_*`,
// called by the kernel before its walks) and the walks read them there (one ds_read where a chain of remainder / fma /
// |x| - h was recomputed in every brick by every lane).  The arithmetic is the same statement on the same coordinate
// value, so the bits are the same.  Candidates: phase 1's single-axis values worth at least `min_cost` instructions; which
// of them become table columns is decided by the code that reads them (render_variant, the direction blocks, the
// builders of the pair tables).
// PAIR TABLES: the same for statements that read TWO coordinates -- a bar of a cross, any extruded profile: its distance
// takes nx * ny distinct values in a box of nx * ny * nz voxels.  In a 16^3 box such a statement is
// evaluated 256 times into a 2D table, from the single-axis columns, instead of 4096 times by the walks.
// -> per statement: 0 not a candidate, 1 a single-axis column, 2 a pair column (`min_pair` = 0: no pair tables)
inline std::vector<char> table_candidates(const Phase1& ph, int min_cost, int min_pair = 0)
{
    const std::vector<Stmt>& st = ph.e.st;
    std::vector<char> out(st.size(), 0);
    const std::vector<int> cost = statement_costs(ph);
    for (int i = 0; i < ph.n_phase1; ++i) {
        if (st[i].ops.empty() || st[i].mask) continue;
        const uint8_t d = st[i].deps;
        if ((d == DX || d == DY || d == DZ) && cost[i] >= min_cost) out[i] = 1;
        else if ((d == (DX | DY) || d == (DX | DZ) || d == (DY | DZ)) && min_pair > 0 && cost[i] >= min_pair) out[i] = 2;
    }
    return out;
}
// the table a statement's column lives in, by the coordinates it reads: x, y, z, xy, xz, yz
inline int table_slot(uint8_t deps)
{
    switch (deps) { case DX: return 0; case DY: return 1; case DZ: return 2; case DX | DY: return 3; case DX | DZ: return 4; default: return 5; }
}
constexpr const char* kTableName[6] = {"X", "Y", "Z", "XY", "XZ", "YZ"};

// the expression that reads column `column` of a statement's table (interp.hpp AxisTabs / BoxTabs); `single`: one entry
// (the table builders: a lane fills one entry) where the walks read the pair of a lane's two voxels
inline std::string table_load(const Stmt& s, int column, bool single = false)
{
    return std::string("tb.template ") + kTableName[table_slot(s.deps)] + (single && (s.deps & DX) ? "1" : "") + "<" + std::to_string(column) + ">()";
}

// BOX PRUNING (round 4).  A CAD assembly is a union of many parts, and a 16^3 box of samples is near two or three of them:
// for most boxes most operands of most min / max cannot win ANYWHERE in the box.  Which ones is decided once per box, before
// the box's kernel runs (kernels.hpp k_box_masks, one box per lane): the distances are evaluated at the box's centre and
// bounded over the box --
//   a primitive that is an exact distance in its local frame (circle, rectangle, sphere, half-space, the slab of an
//   extrusion) is Lipschitz in the sample point with its frame's constant: centre value +- (constant * the box's radius
//   over the coordinates it reads + a margin for rounding);
//   scalings, offsets, shells, extrusions (perp is non-decreasing in both operands), min and max carry bounds through;
//   what has no bound (gears, polygons, anything behind a repetition or a twist) is (-inf, +inf) -- and is still skipped
//   where the min / max structure above it decides without it: a gear lives between its root circle and its tip circle.
// An operand of min whose lower bound lies above the other's upper bound loses in every sample of the box: the hardware
// minimum then returns the other operand's bits, so dropping the loser -- with everything only it needed -- changes no
// bit of the result (and its choice mask for the direction phase is constant).  The box's mask has one bit per prunable
// operand ("scope": alive or not); the generated code tests them with scalar branches.  Tapes in which nothing can be
// bounded (the sponge: everything sits behind a repetition) get no bits and exactly the code they had.
struct PruneInfo {
    int n_bits = 0;
    std::vector<int> scope_of;                 // per statement: its scope (0 = always evaluated)
    std::vector<int> parent;                   // per scope
    std::vector<int> bit;                      // per scope: the mask bit that says it is alive (scope 0: -1)
    struct Sel { int a_scope = 0, b_scope = 0; };
    std::vector<Sel> sel;                      // per statement: the scopes of a guarded select's operands (0: that side is never pruned)
    std::vector<int> select_of_choice;         // per statement: the select whose choice mask it is, or -1
    std::vector<char> need_iv;                 // the statements the mask function bounds
    int words() const { return (n_bits + 31) / 32; }
    bool guarded(int i) const { return i < (int)sel.size() && (sel[i].a_scope || sel[i].b_scope); }
    std::string alive(int scope) const { return "pr.template alive<" + std::to_string(bit[scope]) + ">()"; }
    std::vector<int> path(int scope) const
    {
        std::vector<int> p;
        for (int s = scope; s > 0; s = parent[s]) p.push_back(s);
        std::reverse(p.begin(), p.end());
        return p;
    }
};
constexpr int kMaxPruneBits = 512;

inline PruneInfo analyse_pruning(const Phase1& ph, int min_cost)
{
    const std::vector<Stmt>& st = ph.e.st;
    const int n = ph.n_phase1;
    PruneInfo pi;
    pi.scope_of.assign(st.size(), 0);
    pi.sel.assign(st.size(), PruneInfo::Sel());
    pi.select_of_choice.assign(st.size(), -1);
    pi.need_iv.assign(st.size(), 0);
    pi.parent.push_back(-1);
    pi.bit.push_back(-1);
    if (min_cost <= 0 || ph.root < 0) return pi;
    // which bounds can be finite at all
    std::vector<char> lo(st.size(), 0), hi(st.size(), 0);
    for (int i = 0; i < n; ++i) {
        const Stmt& s = st[i];
        const int a = s.iv_a, b = s.iv_b;
        switch (s.iv) {
        case IV_LEAF: lo[i] = hi[i] = 1; break;
        case IV_SCALE: if (s.iv_c > 0.0f) { lo[i] = lo[a]; hi[i] = hi[a]; } else if (s.iv_c < 0.0f) { lo[i] = hi[a]; hi[i] = lo[a]; } else lo[i] = hi[i] = 1; break;
        case IV_OFFSET: lo[i] = lo[a]; hi[i] = hi[a]; break;
        case IV_SHELL: lo[i] = 1; hi[i] = lo[a] && hi[a]; break;
        case IV_PERP: lo[i] = lo[a] || lo[b]; hi[i] = hi[a] && hi[b]; break;
        case IV_MIN: lo[i] = lo[a] && lo[b]; hi[i] = hi[a] || hi[b]; break;
        case IV_MAX: lo[i] = lo[a] || lo[b]; hi[i] = hi[a] && hi[b]; break;
        case IV_MAXNEG: lo[i] = lo[a] || hi[b]; hi[i] = hi[a] && lo[b]; break;
        default: break;
        }
    }
    auto is_select = [&](int i) { return st[i].iv == IV_MIN || st[i].iv == IV_MAX || st[i].iv == IV_MAXNEG; };
    for (const auto& sc : ph.choice_of_select) pi.select_of_choice[sc.second] = sc.first;
    std::vector<int> choice_of(st.size(), -1);
    for (const auto& sc : ph.choice_of_select) choice_of[sc.first] = sc.second;
    // the statements the root distance is computed from, and who reads whom
    std::vector<char> in_graph(st.size(), 0);
    std::vector<std::vector<int>> users(st.size());
    {
        std::vector<int> stack{ph.root};
        for (const auto& sc : ph.choice_of_select) stack.push_back(sc.second);   // (the comparisons of the direction phase read operands too)
        while (!stack.empty()) {
            const int i = stack.back();
            stack.pop_back();
            if (in_graph[i]) continue;
            in_graph[i] = 1;
            for (int o : st[i].ops) { users[o].push_back(i); stack.push_back(o); }
        }
    }
    const std::vector<int> cost = statement_costs(ph);
    auto depth_of = [&](int s) { int d = 0; for (; s > 0; s = pi.parent[s]) ++d; return d; };
    auto lca = [&](int x, int y) {
        int dx = depth_of(x), dy = depth_of(y);
        while (dx > dy) { x = pi.parent[x]; --dx; }
        while (dy > dx) { y = pi.parent[y]; --dy; }
        while (x != y) { x = pi.parent[x]; y = pi.parent[y]; }
        return x;
    };
    // users have larger ids than what they read -- except that a select's comparison was made just before the select
    // itself: a comparison lives where its select lives and reads each operand in that operand's scope
    std::vector<char> placed(st.size(), 0);
    auto place = [&](int i) {
        if (!in_graph[i] || placed[i]) return;
        placed[i] = 1;
        int scope = -1;
        if (i == ph.root) scope = 0;
        if (pi.select_of_choice[i] >= 0) scope = pi.scope_of[pi.select_of_choice[i]];
        for (int u : users[i]) {
            int at = pi.scope_of[u];
            const int sel = is_select(u) ? u : pi.select_of_choice[u];     // (the select u is, or compares for)
            if (sel >= 0 && pi.guarded(sel) && st[u].ops.size() == 2) {
                // a select's first operand is its `a`, and so is its comparison's (a itself or -a); the second its `b`
                const bool on_a = st[u].ops[0] == i, on_b = st[u].ops[1] == i;
                const int base = pi.scope_of[sel];
                if (on_a && !on_b) at = pi.sel[sel].a_scope ? pi.sel[sel].a_scope : base;
                else if (on_b && !on_a) at = pi.sel[sel].b_scope ? pi.sel[sel].b_scope : base;
                else at = base;
            }
            scope = scope < 0 ? at : lca(scope, at);
        }
        pi.scope_of[i] = scope < 0 ? 0 : scope;
        if (is_select(i)) {
            const int a = st[i].iv_a, b = st[i].iv_b;
            bool can_a = false, can_b = false;     // "a can be pruned", "b can be pruned"
            if (st[i].iv == IV_MIN) { can_b = hi[a] && lo[b]; can_a = hi[b] && lo[a]; }
            else if (st[i].iv == IV_MAX) { can_b = lo[a] && hi[b]; can_a = lo[b] && hi[a]; }
            else { can_b = lo[a] && lo[b]; can_a = hi[b] && hi[a]; }
            if (a == b) can_a = can_b = false;
            auto open = [&](int operand) {
                if (cost[operand] < min_cost || pi.n_bits >= kMaxPruneBits) return 0;
                pi.parent.push_back(pi.scope_of[i]);
                pi.bit.push_back(pi.n_bits++);
                return (int)pi.parent.size() - 1;
            };
            if (can_a) pi.sel[i].a_scope = open(a);
            if (can_b) pi.sel[i].b_scope = open(b);
        }
    };
    for (int i = (int)st.size() - 1; i >= 0; --i) {
        if (i < n && is_select(i) && choice_of[i] >= 0) {
            place(i);                 // the select first, then its comparison (whose id is smaller but not adjacent: negations lie between)
            place(choice_of[i]);
        }
        place(i);
    }
    if (pi.n_bits == 0) return pi;
    // what the mask function has to bound: the operands of the guarded selects, down to the leaves
    std::vector<int> stack;
    for (int i = 0; i < n; ++i) if (in_graph[i] && pi.guarded(i)) { stack.push_back(st[i].iv_a); stack.push_back(st[i].iv_b); }
    int leaves = 0, conditional = 0;
    while (!stack.empty()) {
        const int i = stack.back();
        stack.pop_back();
        if (i < 0 || pi.need_iv[i]) continue;
        pi.need_iv[i] = 1;
        if (st[i].iv == IV_LEAF) { ++leaves; conditional += st[i].iv_conds.empty() ? 0 : 1; }
        if (st[i].iv != IV_LEAF && st[i].iv != IV_UNKNOWN) { stack.push_back(st[i].iv_a); stack.push_back(st[i].iv_b); }
    }
    // A tape that is MOSTLY repetitions (the sponge: twelve of its thirteen primitives sit behind one) is left alone: in a box
    // its primitives are table reads, three vector instructions each, and the scalar tests cost more than what they skip
    // (measured: the bench's leaf blocks 0.137 -> 0.187 ms, C5 6.5 -> 8.5 ms with the sponge's 24 scopes guarded in its
    // distance walks, 0.41 -> 0.66 ms for its float4 grid guarded throughout).  Bounds behind a repetition serve assemblies
    // that contain one (a bolt circle), not fractals.  HU_PRUNE_COND_SHARE: the share of conditional leaves from which on a
    // tape gets no scopes (percent, default 50).
    static const int cond_share = [] { const char* e = std::getenv("HU_PRUNE_COND_SHARE"); return e && *e ? std::atoi(e) : 50; }();
    if (leaves > 0 && conditional * 100 > leaves * cond_share) {
        PruneInfo none;
        none.scope_of.assign(st.size(), 0);
        none.sel.assign(st.size(), PruneInfo::Sel());
        none.select_of_choice = pi.select_of_choice;
        none.need_iv.assign(st.size(), 0);
        none.parent.push_back(-1);
        none.bit.push_back(-1);
        return none;
    }
    return pi;
}

// Statements as guarded assignments (box pruning): every value is declared first (its type taken from its expression, which
// nothing evaluates), then assigned in statement order inside the `if`s of its scope's path; a scope is entered again
// when statements of another scope lie between (rare: a subtree's statements are contiguous in tape order).
struct ScopedItem {
    int id;
    std::string decl;      // the expression the type is taken from
    std::string assign;    // complete statement(s) that give t<id> its value, ending in ";"
};
inline std::string emit_scoped(const PruneInfo& pi, const std::vector<ScopedItem>& items, const std::string& pad = "    ")
{
    std::ostringstream o;
    // (declared, NOT initialised: an initial value would be live from here to the assignment in its scope -- every value of the
    // tape at once, 256 registers and one wavefront per SIMD for planetary's float4 kernel; a value is only ever read under
    // the condition it was assigned under)
    for (const ScopedItem& it : items) o << pad << "decltype(" << it.decl << ") t" << it.id << ";\n";
    std::vector<int> open;
    auto indent = [&]() { return pad + std::string(4 * open.size(), ' '); };
    for (const ScopedItem& it : items) {
        const std::vector<int> want = pi.path(pi.scope_of[it.id]);
        size_t keep = 0;
        while (keep < open.size() && keep < want.size() && open[keep] == want[keep]) ++keep;
        while (open.size() > keep) { open.pop_back(); o << indent() << "}\n"; }
        while (open.size() < want.size()) {
            o << indent() << "if (" << pi.alive(want[open.size()]) << ") {\n";
            open.push_back(want[open.size()]);
        }
        o << indent() << it.assign << "\n";
    }
    while (!open.empty()) { open.pop_back(); o << indent() << "}\n"; }
    return o.str();
}
// the assignment of a guarded select / of its comparison: the operation where both operands are alive, else the survivor
inline std::string guarded_select(const PruneInfo& pi, const Stmt& s, int id, const std::function<std::string(int)>& name)
{
    const PruneInfo::Sel& g = pi.sel[id];
    const std::string t = "t" + std::to_string(id), a = name(s.ops[0]), b = name(s.ops[1]);
    const std::string as = "as<decltype(" + t + ")>(";
    const std::string only_a = t + " = " + as + a + ");", only_b = t + " = " + as + (s.iv == IV_MAXNEG ? "-(" + b + ")" : b) + ");";
    const std::string both = t + " = " + render(s, name) + ";";
    if (g.a_scope && g.b_scope)
        return "if (" + pi.alive(g.a_scope) + " && " + pi.alive(g.b_scope) + ") " + both + " else if (" + pi.alive(g.a_scope) + ") " + only_a + " else " + only_b;
    if (g.b_scope) return "if (" + pi.alive(g.b_scope) + ") " + both + " else " + only_a;
    return "if (" + pi.alive(g.a_scope) + ") " + both + " else " + only_b;
}
inline std::string guarded_choice(const PruneInfo& pi, const Stmt& s, int id, int select, const std::function<std::string(int)>& name)
{
    const PruneInfo::Sel& g = pi.sel[select];
    const std::string t = "t" + std::to_string(id), both = t + " = " + render(s, name) + ";";
    const std::string constant = t + " = mask_const<decltype(" + t + ")>(";       // true: the first operand is the one that is left
    if (g.a_scope && g.b_scope)
        return "if (" + pi.alive(g.a_scope) + " && " + pi.alive(g.b_scope) + ") " + both + " else " + constant + pi.alive(g.a_scope) + ");";
    if (g.b_scope) return "if (" + pi.alive(g.b_scope) + ") " + both + " else " + constant + "true);";
    return "if (" + pi.alive(g.a_scope) + ") " + both + " else " + constant + "false);";
}

// `tabc`: the table candidates the walk may read (empty: none); `tab_index` (may be NULL while the columns are still
// being counted): statement -> its column; `held` (may be empty): the columns that do not change along the walk and are
// read ONCE, in `pre`, and kept in registers (a table read is not free: the data returning from LDS takes the register
// file's write port for about one vector instruction per dword -- measured: 220 vector instructions + 34 dwords read per
// brick ran like 264 --, so a column the walk reads in every brick is worth a register or two once registers are there).
inline Variant render_variant(const Phase1& ph, const std::vector<char>& hoistable, const std::vector<int>& roots,
                              const std::vector<char>& tabc = std::vector<char>(), const std::vector<int>* tab_index = nullptr,
                              const std::vector<char>& held = std::vector<char>(), const PruneInfo* pi = nullptr)
{
    const bool scoped = pi && pi->n_bits > 0;       // box pruning: guarded assignments (emit_scoped)
    const std::vector<Stmt>& st = ph.e.st;
    const int n = (int)st.size();
    auto invariant = [&](int i) { return hoistable[i] != 0; };
    auto tabled = [&](int i) { return i < (int)tabc.size() && tabc[i] != 0; };
    auto is_held = [&](int i) { return i < (int)held.size() && held[i] != 0; };
    auto column = [&](int i) { return table_load(st[i], tab_index ? (*tab_index)[i] : 0); };
    // (statements without operands -- the sample coordinates, constants -- are names, not work: never handed on)
    std::vector<char> in_main(n, 0), frontier(n, 0), in_pre(n, 0), tab_read(n, 0), pre_load(n, 0);
    std::vector<int> stack(roots.begin(), roots.end());
    while (!stack.empty()) {
        const int i = stack.back();
        stack.pop_back();
        if (i < 0) continue;
        if (is_held(i)) { frontier[i] = 1; continue; }
        if (tabled(i)) { tab_read[i] = 1; continue; }      // (a table column costs no register for the walk: before `pre`)
        if (invariant(i)) { frontier[i] = 1; continue; }
        if (in_main[i]) continue;
        in_main[i] = 1;
        for (int o : st[i].ops) stack.push_back(o);
    }
    for (int i = 0; i < n; ++i) if (frontier[i]) stack.push_back(i);
    while (!stack.empty()) {
        const int i = stack.back();
        stack.pop_back();
        if (in_pre[i] || pre_load[i]) continue;
        if (tabled(i)) { pre_load[i] = 1; continue; }      // (`pre` runs after the tables are built: it reads them too)
        in_pre[i] = 1;
        for (int o : st[i].ops) stack.push_back(o);
    }
    Variant v;
    auto plain = [&](int i) { return st[i].ops.empty() ? st[i].text : "t" + std::to_string(i); };
    std::ostringstream pre, main;
    // one statement as a scoped item: a guarded select or its comparison assigns by cases, anything else plainly
    auto scoped_item = [&](int i, const std::string& rhs, const std::function<std::string(int)>& name) {
        const std::string t = "t" + std::to_string(i);
        if (pi->guarded(i) && rhs.compare(0, 3, "tb.") != 0) return ScopedItem{i, rhs, guarded_select(*pi, st[i], i, name)};
        const int sel = pi->select_of_choice[i];
        if (sel >= 0 && pi->guarded(sel)) return ScopedItem{i, rhs, guarded_choice(*pi, st[i], i, sel, name)};
        return ScopedItem{i, rhs, t + " = " + rhs + ";"};
    };
    std::vector<ScopedItem> pre_items;
    for (int i = 0; i < n; ++i) {
        if (st[i].ops.empty()) continue;
        if (scoped) {
            if (pre_load[i]) pre_items.push_back(ScopedItem{i, column(i), "t" + std::to_string(i) + " = " + column(i) + ";"});
            else if (in_pre[i]) pre_items.push_back(scoped_item(i, render(st[i], plain), plain));
            continue;
        }
        if (pre_load[i]) pre << "    const auto t" << i << " = " << column(i) << ";\n";
        else if (in_pre[i]) pre << "    const auto t" << i << " = " << render(st[i], plain) << ";\n";
    }
    if (scoped) pre << emit_scoped(*pi, pre_items);
    std::ostringstream members, values;
    for (int i = 0; i < n; ++i)
        if (frontier[i]) {
            members << " plain_t<decltype(t" << i << ")> t" << i << ";";       // (not const: the walks of a cut box assign the struct)
            values << (v.n_hoisted ? ", " : "") << "t" << i;
            ++v.n_hoisted;
        }
    if (v.n_hoisted) {
        pre << "    struct Hoisted {" << members.str() << " };\n    return Hoisted{" << values.str() << "};\n";
        v.pre = pre.str();
    }
    auto in_walk = [&](int i) { return st[i].ops.empty() ? st[i].text : (frontier[i] ? "h.t" : "t") + std::to_string(i); };
    // min(min(a, b), c) -> min3(a, b, c) (and max) where this body is the inner result's only reader -- the distances of a
    // union of several shapes when no comparison of the direction phase looks at the partial minimum (interp.hpp min3_)
    std::vector<int> readers(n, 0), fused_inner(n, -1);
    static const bool fuse = [] { const char* e = std::getenv("HU_MINMAX3"); return !(e && e[0] == '0'); }();
    for (int i = 0; i < n; ++i) if (in_main[i]) for (int o : st[i].ops) ++readers[o];
    for (int r : roots) if (r >= 0) ++readers[r];
    std::vector<char> dead(n, 0);
    for (int i = 0; fuse && i < n; ++i) {
        if (!in_main[i] || st[i].ops.size() != 2) continue;
        if (st[i].text != "min_x($0, $1)" && st[i].text != "max_x($0, $1)") continue;
        const int j = st[i].ops[0];
        if (!in_main[j] || readers[j] != 1 || st[j].text != st[i].text || fused_inner[j] >= 0) continue;
        if (scoped && (pi->guarded(i) || pi->guarded(j) || pi->scope_of[i] != pi->scope_of[j])) continue;   // (each keeps its own cases)
        fused_inner[i] = j;
        dead[j] = 1;
    }
    std::vector<ScopedItem> main_items;
    for (int i = 0; i < n; ++i) {
        if (st[i].ops.empty()) continue;
        std::string rhs;
        if (tab_read[i]) rhs = column(i);
        else if (in_main[i] && !dead[i]) {
            if (fused_inner[i] >= 0) {
                const Stmt& in = st[fused_inner[i]];
                rhs = std::string(st[i].text[1] == 'i' ? "min3_x(" : "max3_x(") + in_walk(in.ops[0]) + ", " + in_walk(in.ops[1]) + ", " + in_walk(st[i].ops[1]) + ")";
            } else rhs = render(st[i], in_walk);
        } else continue;
        if (!scoped) main << "    const auto t" << i << " = " << rhs << ";\n";
        else if (tab_read[i] || fused_inner[i] >= 0) main_items.push_back(ScopedItem{i, rhs, "t" + std::to_string(i) + " = " + rhs + ";"});
        else main_items.push_back(scoped_item(i, rhs, in_walk));
    }
    if (scoped) main << emit_scoped(*pi, main_items);
    v.main = main.str();
    v.handed = frontier;
    v.tab_read = tab_read;
    v.tab_main = tab_read;
    for (int i = 0; i < n; ++i) if (pre_load[i]) v.tab_read[i] = 1;   // (what needs a column, whoever reads it)
    return v;
}

// The function that fills one table for one entry: the statements of its columns and what they are computed from, each
// column stored at out[column * S].  A single-axis table computes everything from its coordinate; a pair table reads the
// single-axis columns (`tabc`, may be empty; `reads`, may be NULL, collects them) and computes the rest.
inline std::string render_table_builder(const Phase1& ph, uint8_t deps, const std::vector<int>& tab_index, const std::vector<char>& tabc,
                                        const std::vector<char>& used, std::vector<char>* reads = nullptr, const PruneInfo* pi = nullptr)
{
    const std::vector<Stmt>& st = ph.e.st;
    const int n = (int)st.size();
    const bool pair = (deps & (deps - 1)) != 0;
    std::vector<char> in(n, 0), loads(n, 0);
    std::vector<int> stack;
    for (int i = 0; i < n; ++i) if (used[i] && st[i].deps == deps && !st[i].ops.empty()) stack.push_back(i);
    while (!stack.empty()) {
        const int i = stack.back();
        stack.pop_back();
        if (in[i] || loads[i]) continue;
        if (pair && i < (int)tabc.size() && tabc[i] == 1) { loads[i] = 1; if (reads) (*reads)[i] = 1; continue; }
        in[i] = 1;
        for (int o : st[i].ops) stack.push_back(o);
    }
    auto plain = [&](int i) { return st[i].ops.empty() ? st[i].text : "t" + std::to_string(i); };
    std::ostringstream o;
    // box pruning guards the entries of the PAIR tables (256 evaluations per column and box; an axis table's 16 are not worth a branch)
    const bool scoped = pi && pi->n_bits > 0 && pair;
    std::vector<ScopedItem> items;
    for (int i = 0; i < n; ++i) {
        if (st[i].ops.empty()) continue;
        const std::string t = "t" + std::to_string(i);
        if (loads[i]) {
            const std::string rhs = table_load(st[i], tab_index[i], true);
            if (scoped) items.push_back(ScopedItem{i, rhs, t + " = " + rhs + ";"});
            else o << "    const auto " << t << " = " << rhs << ";\n";
        }
        if (!in[i]) continue;
        const std::string rhs = render(st[i], plain);
        const std::string store = used[i] && st[i].deps == deps ? "out[" + std::to_string(tab_index[i]) + " * S] = " + t + ";" : std::string();
        if (scoped) {
            const int sel = pi->select_of_choice[i];
            const std::string assign = pi->guarded(i) ? guarded_select(*pi, st[i], i, plain)
                                       : sel >= 0 && pi->guarded(sel) ? guarded_choice(*pi, st[i], i, sel, plain) : t + " = " + rhs + ";";
            items.push_back(ScopedItem{i, rhs, assign + (store.empty() ? "" : " " + store)});
        } else {
            o << "    const auto " << t << " = " << rhs << ";\n";
            if (!store.empty()) o << "    " << store << "\n";
        }
    }
    if (scoped) o << emit_scoped(*pi, items);
    return o.str();
}

// The mask function of a tape (box pruning): one box per call -- its centre (px, py, pz), its half extents (hx, hy, hz: the
// samples lie within them of the centre) -> out.w[]: bit k set = scope k is alive.  Everything is evaluated in place, in
// float; the bounds (interp.hpp Iv) are widened outwards at every step, and a leaf's radius carries a margin of 64 ulps of
// the largest magnitude its arithmetic can see (g * B + o, B = the box's largest |coordinate|), so that rounding in the
// kernels' evaluation of the same statements cannot carry a value out of its bounds.
inline std::string render_prune_function(const Phase1& ph, const PruneInfo& pi)
{
    const std::vector<Stmt>& st = ph.e.st;
    const int n = ph.n_phase1;
    std::ostringstream o;
    o << "template <class PR> __device__ __forceinline__ void tape_prune(float px, float py, float pz, float hx, float hy, float hz, "
         "const float* __restrict__ extra, PR& out)\n{\n    using namespace sdf;\n";
    if (pi.n_bits == 0) { o << "}\n"; return o.str(); }
    o << "    const uint32_t flags = 0u;\n";
    // the centre values of the bounded leaves, in place
    std::vector<char> in(st.size(), 0), cond_used(ph.conds.size(), 0);
    std::vector<int> stack;
    for (int i = 0; i < n; ++i)
        if (pi.need_iv[i] && st[i].iv == IV_LEAF) {
            stack.push_back(i);
            for (int c : st[i].iv_conds) { cond_used[c] = 1; stack.push_back(ph.conds[c].u); }
        }
    while (!stack.empty()) {
        const int i = stack.back();
        stack.pop_back();
        if (in[i]) continue;
        in[i] = 1;
        for (int op : st[i].ops) stack.push_back(op);
    }
    auto plain = [&](int i) { return st[i].ops.empty() ? st[i].text : "t" + std::to_string(i); };
    for (int i = 0; i < n; ++i)
        if (in[i] && !st[i].ops.empty()) o << "    const auto t" << i << " = " << render(st[i], plain) << ";\n";
    // radii over the coordinates a leaf reads (index = DX | DY | DZ), a hair large; the box's largest |coordinate|
    o << "    const float kUp = 1.0009765625f;\n"
      << "    const float rad[8] = {0.0f, hx * kUp, hy * kUp, iv_hypot(hx, hy) * kUp, hz * kUp, iv_hypot(hx, hz) * kUp, iv_hypot(hy, hz) * kUp, "
         "iv_hypot(iv_hypot(hx, hy), hz) * kUp};\n"
      << "    const float B = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(px) + hx, __builtin_fabsf(py) + hy), __builtin_fabsf(pz) + hz);\n";
    auto up = [](double v) { return flit(std::nextafter((float)(v * (1.0 + 1e-6)), HUGE_VALF)); };     // a non-negative constant, rounded up
    auto iv = [](int i) { return "i" + std::to_string(i); };
    // the conditions of the repetitions: does the coordinate that enters the remainder stay inside one cell over the box?
    for (size_t c = 0; c < ph.conds.size(); ++c) {
        if (!cond_used[c]) continue;
        const Phase1::Cond& k = ph.conds[c];
        const Stmt& u = st[k.u];
        o << "    const bool same" << c << " = iv_same_cell(" << plain(k.u) << ", " << up(k.lip) << " * rad[" << (int)u.deps << "] + (" << up(u.g)
          << " * B + " << up(u.o) << ") * kIvMargin, " << flit(k.inv) << ");\n";
    }
    for (int i = 0; i < n; ++i) {
        if (!pi.need_iv[i]) continue;
        const Stmt& s = st[i];
        o << "    const Iv " << iv(i) << " = ";
        switch (s.iv) {
        case IV_LEAF: {
            std::string ok;
            for (int c : s.iv_conds) ok += (ok.empty() ? "" : " && ") + ("same" + std::to_string(c));
            if (!ok.empty()) o << "!(" << ok << ") ? iv_unknown() : ";
            o << "iv_leaf(" << plain(i) << ", " << up(s.iv_l) << " * rad[" << (int)s.deps << "], (" << up(s.g) << " * B + " << up(s.o) << ") * kIvMargin)";
            break;
        }
        case IV_SCALE: o << (s.iv_c == 0.0f ? std::string("iv_zero()") : "iv_scale(" + iv(s.iv_a) + ", " + flit(s.iv_c) + ")"); break;
        case IV_OFFSET: o << "iv_offset(" << iv(s.iv_a) << ", " << flit(s.iv_c) << ")"; break;
        case IV_SHELL: o << "iv_shell(" << iv(s.iv_a) << ", " << flit(s.iv_c) << ")"; break;
        case IV_PERP: o << "iv_perp(" << iv(s.iv_a) << ", " << iv(s.iv_b) << ")"; break;
        case IV_MIN: o << "iv_min(" << iv(s.iv_a) << ", " << iv(s.iv_b) << ")"; break;
        case IV_MAX: o << "iv_max(" << iv(s.iv_a) << ", " << iv(s.iv_b) << ")"; break;
        case IV_MAXNEG: o << "iv_max(" << iv(s.iv_a) << ", iv_neg(" << iv(s.iv_b) << "))"; break;
        default: o << "iv_unknown()"; break;
        }
        o << ";\n";
    }
    // the decisions: an operand that loses everywhere in the box is dead
    const int words = pi.words();
    for (int w = 0; w < words; ++w) o << "    uint32_t w" << w << " = 0xffffffffu;\n";
    auto kill = [&](int scope, const std::string& cond) {
        if (!scope) return;
        const int b = pi.bit[scope];
        o << "    if (" << cond << ") w" << (b >> 5) << " &= ~" << (1u << (b & 31)) << "u;\n";
    };
    for (int i = 0; i < n; ++i) {
        if (!pi.guarded(i)) continue;
        const Stmt& s = st[i];
        const std::string a = iv(s.iv_a), b = iv(s.iv_b);
        if (s.iv == IV_MIN) { kill(pi.sel[i].b_scope, a + ".hi < " + b + ".lo"); kill(pi.sel[i].a_scope, b + ".hi < " + a + ".lo"); }
        else if (s.iv == IV_MAX) { kill(pi.sel[i].b_scope, a + ".lo > " + b + ".hi"); kill(pi.sel[i].a_scope, b + ".lo > " + a + ".hi"); }
        else { kill(pi.sel[i].b_scope, a + ".lo > -" + b + ".lo"); kill(pi.sel[i].a_scope, "-" + b + ".hi > " + a + ".hi"); }
    }
    for (int w = 0; w < words; ++w) o << "    out.w[" << w << "] = w" << w << ";\n";
    o << "}\n";
    return o.str();
}

}  // namespace spec_detail

// true: the deferred form was emitted; false: nothing was written (use emit_plain)
// What the host needs to know about the generated code (hip_util.hip SpecKernels).
struct SpecMeta {
    bool deferred = false;
    double coord_limit = 0.0;    // the largest |sample coordinate| for which a launch may set sdf::kFlagInRange (0: never)
    int tabs[6] = {0, 0, 0, 0, 0, 0};   // columns of a box's tables: x, y, z, xy, xz, yz
    int prune_words = 0;         // 32-bit words of a box's pruning mask (0: the tape has nothing to prune)
    int prune_bits = 0;
    bool prune_all = false;      // the float4 walks, `pre` and the table builders are guarded too (else: the distance walks only)
    bool plain_in_place = false; // the in-place functions (single points: ragged grids, small levels, the ray caster) are the plain form
};
// Up to this many (primitive, path) pairs the in-place functions defer directions too; beyond it they are the plain
// record-by-record form (its second phase in place -- every value two voxels wide, no tables, no pruning -- held 250
// registers for planetary's 80 pairs and took as long to compile as everything else together)
constexpr size_t kInPlaceDeferredPaths = 40;
constexpr int kMaxTableColumns = 48;   // per axis (a column of a 16^3 box is 64 B of LDS)
constexpr int kMaxPairColumns = 16;    // per pair of axes (a column of a 16^3 box is 1 KiB) ...
constexpr int kMaxPairTotal = 24;      // ... and in all: the tables of a box stay below 28 KiB, five workgroups to a CU

inline bool emit_deferred(std::ostringstream& o, const SpecProgram& p, size_t max_paths = 400, SpecMeta* meta = nullptr)
{
    using namespace spec_detail;
    if (p.dist.empty() || p.dist.size() != p.full.size()) return false;
    std::vector<Node> nodes;
    int root;
    if (!build_graph(p.full, nodes, root)) return false;
    std::vector<Path> paths;
    if (!collect_paths(nodes, root, Path(), paths, max_paths)) return false;
    if (paths.size() < 2) return false;   // a single primitive: nothing to defer

    // ---- phase 1: distances; what phase 2 wants from it: the comparison at every select on a path, the distance that
    // entered an op whose direction reads it
    std::vector<char> is_choice(p.full.size(), 0), keep_w(p.full.size(), 0);
    for (const Path& path : paths) {
        for (auto& c : path.choices) is_choice[c.first] = 1;
        for (const Step& s : path.up)
            if (s.node >= 0 && reads_input_distance(nodes[s.node].op)) keep_w[nodes[s.node].rec] = 1;
    }
    Phase1 ph;
    std::vector<std::array<int, 3>> pt(nodes.size(), {{-1, -1, -1}});   // local coordinates by (point) node
    if (!symbolic_phase1(p, nodes, root, is_choice, keep_w, ph, &pt)) return false;
    if (meta) meta->coord_limit = coordinate_limit(ph);
    ph.n_phase1 = (int)ph.e.st.size();
    // box pruning: which operands of which selects can be decided per box (HU_PRUNE=0: none; HU_PRUNE_MIN: what an operand
    // must cost, in instructions, to be worth a scalar branch)
    static const int prune_min = [] {
        const char* off = std::getenv("HU_PRUNE");
        if (off && off[0] == '0') return 0;
        const char* e = std::getenv("HU_PRUNE_MIN");
        return e && *e ? std::atoi(e) : 6;
    }();
    const PruneInfo prune = analyse_pruning(ph, prune_min);
    // Where the guards go.  An assembly (many scopes: planetary 131) gets them everywhere: walks of both kinds, `pre`, the pair
    // tables' builders.  A tape with a few (the sponge: 24, all behind repetitions) gets them in the DISTANCE walks only -- leaf
    // blocks, classification, distance grids, where the vector ALU is the bound --: its float4 code, which sits on the store
    // roof, stays exactly as it was (and its launches skip the mask kernel).  HU_PRUNE_EVAL_MIN: the number of scopes from
    // which a tape counts as an assembly.
    static const int prune_eval_min = [] { const char* e = std::getenv("HU_PRUNE_EVAL_MIN"); return e && *e ? std::atoi(e) : 64; }();
    const bool prune_all = prune.n_bits >= prune_eval_min;
    if (meta) { meta->prune_bits = prune.n_bits; meta->prune_words = prune.words(); meta->prune_all = prune_all; }
    std::vector<int> dist_roots{ph.root}, eval_roots{ph.root};
    for (int v : ph.choice_of_rec) if (v >= 0) eval_roots.push_back(v);
    for (int v : ph.keep_w_of_rec) if (v >= 0) eval_roots.push_back(v);

    // ---- phase 2: the directions of the primitives that win somewhere in the wavefront.  For every (primitive, path to
    // the root) a block under a wave-uniform branch: the primitive's direction from its local coordinates, pushed through
    // the ops on the path -- restated component by component like phase 1 (interp.hpp "directions of mixed width"), in the
    // SAME statement list, so that a coordinate phase 1 already names is found again: one that is handed from `pre` is
    // simply read there; anything else is computed again inside the block from opaque copies of the sample point
    // (reusing phase 1's registers across the whole second phase would cost more than recomputing: interp.hpp opaque).
    Emitter& e = ph.e;
    const int zero = e.add("0.0f", {});
    struct PathCode { std::string mask; int d[3]; };
    std::vector<PathCode> codes;
    {
        auto full_record = [&](const Rec& r, const int (&last)[4], const int (&operand)[4]) {
            const uint32_t op = r.hdr & 0xffu;
            return e.add("run_record_full<" + std::to_string(op) + ">(" + rec_literal(r, true, op) + ", extra, v4x($0, $1, $2, $3), v4x($4, $5, $6, $7), m)",
                         {last[0], last[1], last[2], last[3], operand[0], operand[1], operand[2], operand[3]});
        };
        auto mulc = [&](int v, float c) { return e.add("$0 * " + flit(c), {v}); };
        for (const Path& path : paths) {
            PathCode code;
            for (size_t k = 0; k < path.choices.size(); ++k)
                code.mask += std::string(k ? " & " : "") + (path.choices[k].second ? "" : "~") + "as_mask(@" + std::to_string(ph.choice_of_rec[path.choices[k].first]) + "@, T())";
            if (path.choices.empty()) code.mask = "mask_of<T>::all()";
            const Node& leaf = nodes[path.leaf];
            const Rec& lr = p.full[leaf.rec];
            const std::array<int, 3>& c = pt[leaf.a];
            int d[3] = {zero, zero, zero};
            if (leaf.op == OP_RECTANGLE) {
                const int v = e.add("rect_dir_x($0, $1, " + flit(lr.p[0]) + ", " + flit(lr.p[1]) + ", m, flags)", {c[0], c[1]});
                d[0] = e.add("$0.x", {v}); d[1] = e.add("$0.y", {v});
            } else if (leaf.op == OP_CIRCLE) {
                const int v = e.add("circle_dir_x($0, $1, m)", {c[0], c[1]});
                d[0] = e.add("$0.x", {v}); d[1] = e.add("$0.y", {v});
            } else if (leaf.op == OP_SPHERE) {
                const int v = e.add("sphere_dir_x($0, $1, $2, m)", {c[0], c[1], c[2]});
                d[0] = e.add("$0.x", {v}); d[1] = e.add("$0.y", {v}); d[2] = e.add("$0.z", {v});
            } else if (leaf.op == OP_HALF_SPACE) {
                d[1] = e.add("-1.0f", {});
            } else {
                const int last[4] = {c[0], c[1], zero, zero}, none4[4] = {zero, zero, zero, zero};      // (2D primitives: x and y only)
                const int v = full_record(lr, last, none4);
                d[0] = e.add("$0.x", {v}); d[1] = e.add("$0.y", {v}); d[2] = e.add("$0.z", {v});
            }
            for (const Step& stp : path.up) {
                if (stp.node < 0) {
                    for (int k = 0; k < 3; ++k) d[k] = e.neg(d[k]);
                    continue;
                }
                const Node& n = nodes[stp.node];
                const Rec& r = p.full[n.rec];
                const float* q = r.p;
                const int w_in = reads_input_distance(n.op) ? ph.keep_w_of_rec[n.rec] : zero;
                switch (n.op) {
                case OPX_FROM_SCALE:
                    for (int k = 0; k < 3; ++k) d[k] = mulc(d[k], q[0]);
                    break;
                case OPX_FROM_AXIS_X: case OPX_FROM_AXIS_Y: case OPX_FROM_AXIS_Z: {
                    // interp.hpp axis_rotate_dir: (along, u, v) = the axis and the other two in cyclic order
                    const int ax = n.op == OPX_FROM_AXIS_X ? 0 : n.op == OPX_FROM_AXIS_Y ? 1 : 2, u = (ax + 1) % 3, v = (ax + 2) % 3;
                    const int along = d[ax], du = d[u], dv = d[v];
                    d[ax] = mulc(along, q[0]);
                    int ru = mulc(e.neg(dv), q[2]), rv = mulc(du, q[2]);
                    if (q[1] != 0.0f) {
                        ru = e.add("fma_x($0, " + flit(q[1]) + ", $1)", {du, ru});
                        rv = e.add("fma_x($0, " + flit(q[1]) + ", $1)", {dv, rv});
                    }
                    d[u] = ru;
                    d[v] = rv;
                    break;
                }
                case OPX_FROM_MATRIX: {
                    const int x = d[0], y = d[1], z = d[2];
                    for (int k = 0; k < 3; ++k)
                        d[k] = e.add("fma_x($0, " + flit(q[3 * k]) + ", fma_x($1, " + flit(q[3 * k + 1]) + ", $2 * " + flit(q[3 * k + 2]) + "))", {x, y, z});
                    break;
                }
                case OP_OFFSET: break;
                case OP_SHELL:
                    for (int k = 0; k < 3; ++k) d[k] = e.add("shell_dir_x($0, $1)", {d[k], w_in});
                    break;
                case OP_MIRROR: d[0] = e.neg(d[0]); break;
                case OP_EXTRUSION: {
                    const int v = e.add("extrusion_dir_x($0, $1, $2, $3, $4, " + flit(q[0]) + ", m, flags)", {d[0], d[1], d[2], w_in, pt[n.b][2]});
                    d[0] = e.add("$0.x", {v}); d[1] = e.add("$0.y", {v}); d[2] = e.add("$0.z", {v});
                    break;
                }
                case OP_SYMMETRICAL_FROM: d[0] = e.add("symm_dir_x($0, $1)", {d[0], pt[n.b][0]}); break;
                default: {   // transformation_from with a general quaternion, circular repetition, the revolutions
                    const int last[4] = {d[0], d[1], d[2], w_in};
                    const int operand[4] = {n.role == WITH_POINT ? pt[n.b][0] : zero, n.role == WITH_POINT ? pt[n.b][1] : zero,
                                            n.role == WITH_POINT ? pt[n.b][2] : zero, zero};
                    const int v = full_record(r, last, operand);
                    d[0] = e.add("$0.x", {v}); d[1] = e.add("$0.y", {v}); d[2] = e.add("$0.z", {v});
                    break;
                }
                }
            }
            code.d[0] = d[0]; code.d[1] = d[1]; code.d[2] = d[2];
            codes.push_back(code);
        }
    }
    // phase 1's values that phase 2 reads where they are (not computed again): the comparisons, the kept distances
    // A kept distance stays in its register(s) from phase 1 to the block that reads it -- for an assembly of extruded
    // profiles that is two registers per extrusion across the whole second phase (planetary: 35 of them) -- unless the block
    // can have it again cheaply: from a table column, or by computing a short subtree again (a circle's distance: ten
    // instructions) that holds nothing decided per box.
    std::vector<char> lives(e.st.size(), 0), kept_w(e.st.size(), 0);
    for (int v : ph.choice_of_rec) if (v >= 0) lives[v] = 1;
    for (int v : ph.keep_w_of_rec) if (v >= 0) kept_w[v] = 1;
    const std::vector<int> p1_cost = statement_costs(ph);
    static const int recompute_limit = [] { const char* e = std::getenv("HU_KEEP_RECOMPUTE"); return e && *e ? std::atoi(e) : 40; }();
    // `tabc` / `tab_index` / `tab_used`: the axis tables (render_variant): a block reads a table column where it would
    // have recomputed the statement; `tab_used` (may be NULL) collects which candidates the blocks read
    auto phase2_for = [&](const std::vector<char>& hoistable, const std::vector<char>& tabc, const std::vector<int>* tab_index,
                          std::vector<char>* tab_used) {
        const std::vector<Stmt>& st = e.st;
        auto outer = [&](int id) { return st[id].ops.empty() ? st[id].text : (id < (int)hoistable.size() && hoistable[id] ? "h.t" : "t") + std::to_string(id); };
        auto tabled = [&](int id) { return id < (int)tabc.size() && tabc[id] != 0; };
        auto handed = [&](int id) { return id < (int)hoistable.size() && hoistable[id] != 0; };
        // can a block compute statement `id` again: a short subtree, down to table columns / handed values, without a select
        // that box pruning guards (its operands may be dead), a comparison, or a library record
        std::function<bool(int)> again = [&](int id) -> bool {
            if (e.st[id].ops.empty() || tabled(id) || handed(id)) return true;
            if (e.st[id].mask || prune.guarded(id) || e.st[id].text.compare(0, 10, "run_record") == 0) return false;
            for (int op : e.st[id].ops) if (!again(op)) return false;
            return true;
        };
        std::vector<char> live_here = lives;
        for (int i = 0; i < (int)kept_w.size(); ++i)
            if (kept_w[i] && !handed(i) && !tabled(i) && !(p1_cost[i] <= recompute_limit && again(i))) live_here[i] = 1;
        std::ostringstream o2;
        o2 << "    // ---- phase 2\n"
           << "    const auto qx = opaque(px); const auto qy = opaque(py); const auto qz = opaque(pz);\n";
        // (table columns are read again where a block wants them: without this the compiler keeps phase 1's copies alive)
        if (!tabc.empty()) o2 << "    asm volatile(\"\" ::: \"memory\");\n";
        o2 << "    V4<T> dir = v4<T>(bc<T>(0.0f), bc<T>(0.0f), bc<T>(0.0f), bc<T>(0.0f));\n";
        // the body of path k's block (`m` = its lanes): the closure of the direction's statements, up to values that live
        // outside it, then three selects per voxel
        auto block_body = [&](size_t k, const std::string& pad) {
            const PathCode& code = codes[k];
            std::ostringstream ob;
            std::vector<char> inside(st.size(), 0), loads(st.size(), 0);
            std::vector<int> stack{code.d[0], code.d[1], code.d[2]};
            while (!stack.empty()) {
                const int i = stack.back();
                stack.pop_back();
                if (inside[i] || loads[i] || st[i].ops.empty() || live_here[i] || (i < (int)hoistable.size() && hoistable[i])) continue;
                if (tabled(i)) { loads[i] = 1; if (tab_used) (*tab_used)[i] = 1; continue; }
                inside[i] = 1;
                for (int op : st[i].ops) stack.push_back(op);
            }
            auto name = [&](int id) -> std::string {
                if (st[id].ops.empty()) return st[id].text == "px" ? "qx" : st[id].text == "py" ? "qy" : st[id].text == "pz" ? "qz" : st[id].text;
                return inside[id] || loads[id] ? "u" + std::to_string(id) : outer(id);
            };
            for (int i = 0; i < (int)st.size(); ++i) {
                if (loads[i]) ob << pad << "const auto u" << i << " = " << table_load(st[i], tab_index ? (*tab_index)[i] : 0) << ";\n";
                else if (inside[i]) ob << pad << "const auto u" << i << " = " << render(st[i], name) << ";\n";
            }
            ob << pad << "dir.x = sel(m, as<T>(" << name(code.d[0]) << "), dir.x); dir.y = sel(m, as<T>(" << name(code.d[1])
               << "), dir.y); dir.z = sel(m, as<T>(" << name(code.d[2]) << "), dir.z);\n";
            return ob.str();
        };
        static const bool tree = [] { const char* e = std::getenv("HU_PHASE2_TREE"); return !(e && e[0] == '0'); }();
        if (tree) {
            // The blocks hang in the tree of the selects: a select splits its lanes between its operands (one scalar
            // and / and-not per voxel of the lane), and an operand nobody chose is left with everything below it -- where
            // the flat form formed every path's product of choices and tested it, 13 paths of up to six factors for
            // sponge(4) (150 scalar instructions per brick against 95 vector ones).  collect_paths met the leaves in
            // this order.
            size_t next = 0;
            std::function<void(int, const std::string&, int)> walk = [&](int at, const std::string& lanes, int depth) {
                const Node& n = nodes[at];
                const std::string pad(4 + 4 * (size_t)depth, ' ');
                if (n.role == LEAF) {
                    o2 << pad << "{   // the primitive of record " << n.rec << "\n" << pad << "    const M m = " << (lanes.empty() ? "mask_of<T>::all()" : lanes) << ";\n"
                       << block_body(next++, pad + "    ") << pad << "}\n";
                } else if (n.role == SELECT) {
                    const std::string c = "as_mask(" + outer(ph.choice_of_rec[n.rec]) + ", T())", id = std::to_string(n.rec);
                    const std::string ma = "m" + id + "a", mb = "m" + id + "b";
                    o2 << pad << "const M " << ma << " = " << (lanes.empty() ? "" : lanes + " & ") << c << ", " << mb << " = " << (lanes.empty() ? "" : lanes + " & ") << "~" << c << ";\n"
                       << pad << "if (wave_any(" << ma << ")) {\n";
                    walk(n.a, ma, depth + 1);
                    o2 << pad << "}\n" << pad << "if (wave_any(" << mb << ")) {\n";
                    walk(n.b, mb, depth + 1);
                    o2 << pad << "}\n";
                } else walk(n.a, lanes, depth);
            };
            walk(root, "", 0);
        } else
        for (size_t k = 0; k < paths.size(); ++k) {
            const PathCode& code = codes[k];
            std::string mask;
            for (size_t i = 0; i < code.mask.size(); ++i) {
                if (code.mask[i] != '@') { mask += code.mask[i]; continue; }
                const size_t end = code.mask.find('@', i + 1);
                mask += outer(std::atoi(code.mask.substr(i + 1, end - i - 1).c_str()));
                i = end;
            }
            o2 << "    {   // the primitive of record " << nodes[paths[k].leaf].rec << " along one path to the root\n        const M m = " << mask
               << ";\n        if (wave_any(m)) {\n" << block_body(k, "            ") << "        }\n    }\n";
        }
        return o2.str();
    };

    const char* head = "    using namespace sdf;\n    using T = wider_t<wider_t<PX, PY>, PZ>;\n    using M = typename mask_of<T>::type;\n";
    // two forms: in place (the classification kernels, the ray caster, ragged grids), and for the walks along x of a box
    // (kernels.hpp box_eval: the dense grids and the leaf blocks)
    struct Form { const char* suffix; uint8_t walk; };
    const Form forms[2] = {{"", 0}, {"_x", DX}};
    // what a value that the walk does not change must save per brick to be computed once per walk and kept in a register
    // (hoistable_set; with the tables few statements are left to it).  Measured on sponge(4), MI355X, before the tables:
    // walks of four bricks along x 0.363 -> 0.356 ms at 8.
    auto knob = [](const char* name, int fallback) { const char* e = std::getenv(name); return e && *e ? std::atoi(e) : fallback; };
    const int tab_min = knob("HU_TAB_MIN", 2);      // what a single-axis value must cost to become a table column (0: no tables)
    const bool plain_in_place = paths.size() > kInPlaceDeferredPaths;
    if (meta) meta->plain_in_place = plain_in_place;
    for (const Form& f : forms) {
        if (f.walk == 0 && plain_in_place) continue;      // (specialised_source emits the plain form under these names)
        const std::vector<char> hoistable = hoistable_set(ph, f.walk, knob("HU_HOIST_MIN_X", 8));
        // ---- axis and pair tables: the candidates this form's walk-dependent code and its direction blocks read become
        // columns; the pair tables are filled from single-axis columns
        std::vector<char> tabc;
        std::vector<int> tab_index(ph.e.st.size(), -1);
        int n_tab[6] = {0, 0, 0, 0, 0, 0};
        std::vector<char> held, used;
        if (f.walk != 0 && tab_min > 0) {
            const std::vector<int> tab_cost = statement_costs(ph);
            tabc = table_candidates(ph, tab_min, knob("HU_TAB_PAIR_MIN", 3));
            for (;;) {
                // the columns the DISTANCES read in every brick and the walk does not change are kept in registers instead
                // (`held`), lowest statements first, while the budget lasts: a column with x in it costs two registers, others one
                const Variant probe = render_variant(ph, hoistable, dist_roots, tabc);
                held.assign(ph.e.st.size(), 0);
                // (measured, sponge(4), MI355X, single-axis tables only: leaf blocks 0.327 / 0.325 / 0.319 / 0.325 ms holding
                // 0 / 6 / 12 / 24 registers' worth; walks of sixteen bricks along z, round 3's first form of the dense kernel,
                // 0.497 / 0.508 / 0.532 / 0.553 ms at 0 / 6 / 9 / 16 -- registers were dearer there than reads)
                // (a tape with box pruning holds none: most of its columns are dead in any one box, and the registers decide how many
                // wavefronts hide its scalar branches -- planetary's distance kernels: 122 -> 78 registers, four -> six wavefronts per SIMD)
                int budget = knob("HU_TAB_HOLD_X", prune_all ? 0 : 12);
                for (int i = 0; i < (int)probe.tab_main.size(); ++i) {
                    if (!probe.tab_main[i] || !tabc[i] || (ph.e.st[i].deps & f.walk)) continue;
                    const int regs = (ph.e.st[i].deps & DX) ? 2 : 1;
                    if (budget >= regs) { held[i] = 1; budget -= regs; }
                }
                const Variant all = render_variant(ph, hoistable, eval_roots, tabc, nullptr, held);
                used = all.tab_read;
                for (int i = 0; i < (int)held.size(); ++i) if (held[i]) used[i] = 1;
                (void)phase2_for(all.handed, tabc, nullptr, &used);
                // what the builders of the pair tables read of the single-axis tables
                for (uint8_t pair : {(uint8_t)(DX | DY), (uint8_t)(DX | DZ), (uint8_t)(DY | DZ)})
                    (void)render_table_builder(ph, pair, tab_index, tabc, std::vector<char>(used), &used);
                std::fill(tab_index.begin(), tab_index.end(), -1);
                std::fill(n_tab, n_tab + 6, 0);
                bool over = false;
                // (the dearest statements first: when a tape has more two-coordinate statements than pair columns -- an assembly
                // of extruded profiles --, the columns go to the gears and polygons, not to the circles)
                std::vector<int> order;
                for (int i = 0; i < (int)used.size(); ++i) if (used[i]) order.push_back(i);
                int wanted[6] = {0, 0, 0, 0, 0, 0};
                for (int i : order) ++wanted[table_slot(ph.e.st[i].deps)];
                const bool crowded = wanted[0] > kMaxTableColumns || wanted[1] > kMaxTableColumns || wanted[2] > kMaxTableColumns || wanted[3] > kMaxPairColumns ||
                                     wanted[4] > kMaxPairColumns || wanted[5] > kMaxPairColumns || wanted[3] + wanted[4] + wanted[5] > kMaxPairTotal;
                if (crowded) std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return tab_cost[x] > tab_cost[y]; });
                for (int i : order) {
                    const int slot = table_slot(ph.e.st[i].deps);
                    const bool fits = slot < 3 ? n_tab[slot] < kMaxTableColumns
                                               : n_tab[slot] < kMaxPairColumns && n_tab[3] + n_tab[4] + n_tab[5] < kMaxPairTotal;
                    if (fits) tab_index[i] = n_tab[slot]++;
                    else { over = true; used[i] = 0; }
                }
                if (!over) break;
                // what did not fit is computed by the walk itself, which may then reach further candidates: only the
                // columns that were given out stay candidates, and the walk is looked at again
                for (int i = 0; i < (int)tabc.size(); ++i) if (tab_index[i] < 0) tabc[i] = 0;
            }
            if (std::accumulate(n_tab, n_tab + 6, 0) == 0) { tabc.clear(); held.clear(); }
        }
        if (meta && f.walk != 0) for (int a = 0; a < 6; ++a) meta->tabs[a] = n_tab[a];
        // (the in-place form evaluates single points: nothing to decide per box)
        const PruneInfo* pi_dist = f.walk != 0 ? &prune : nullptr;
        const PruneInfo* pi = f.walk != 0 && prune_all ? &prune : nullptr;
        const Variant vd = render_variant(ph, hoistable, dist_roots, tabc, &tab_index, held, pi_dist),
                      ve = render_variant(ph, hoistable, eval_roots, tabc, &tab_index, held, pi);
        // the values handed from `pre`: the union of what the distance and the evaluation read (one struct for both)
        const Variant& pre_of = ve;   // (eval's roots include dist's root: its frontier covers it)
        const bool hoists = f.walk != 0 && pre_of.n_hoisted > 0;
        if (f.walk != 0) {
            o << "// hoisted out of walks along x: " << pre_of.n_hoisted << " values; table columns: "
              << n_tab[0] << " / " << n_tab[1] << " / " << n_tab[2] << " (x / y / z), " << n_tab[3] << " / " << n_tab[4] << " / " << n_tab[5] << " (xy / xz / yz)\n"
              << "template <class PX, class PY, class PZ, class TB, class PR> __device__ __forceinline__ auto tape_pre" << f.suffix
              << "(PX px, PY py, PZ pz, const float* __restrict__ extra, uint32_t flags, const TB& tb, const PR& pr)\n{\n    using namespace sdf;\n";
            if (hoists) o << pre_of.pre;
            else o << "    struct Hoisted {};\n    return Hoisted{};\n";
            o << "}\n";
            const char* slot_name[6] = {"x", "y", "z", "xy", "xz", "yz"};
            const uint8_t slot_deps[6] = {DX, DY, DZ, DX | DY, DX | DZ, DY | DZ};
            for (int a = 0; a < 3; ++a)
                o << "template <int S, class L> __device__ __forceinline__ void tape_tab" << f.suffix << "_" << slot_name[a] << "(float p"
                  << slot_name[a] << ", const float* __restrict__ extra, uint32_t flags, L out)\n{\n    using namespace sdf;\n"
                  << (n_tab[a] ? render_table_builder(ph, slot_deps[a], tab_index, tabc, used) : std::string()) << "}\n";
            // (a pair table's entry: the two coordinates, and the single-axis tables positioned at it)
            for (int a = 3; a < 6; ++a)
                o << "template <int S, class TB, class PR, class L> __device__ __forceinline__ void tape_tab" << f.suffix << "_" << slot_name[a]
                  << "(float px, float py, float pz, const float* __restrict__ extra, uint32_t flags, const TB& tb, const PR& pr, L out)\n{\n    using namespace sdf;\n"
                  << (n_tab[a] ? render_table_builder(ph, slot_deps[a], tab_index, tabc, used, nullptr, pi) : std::string()) << "}\n";
            o << render_prune_function(ph, prune);
        }
        const std::string h_param = f.walk != 0 ? ", const H& h, const TB& tb, const PR& pr" : "";
        const std::string h_tmpl = f.walk != 0 ? ", class H, class TB, class PR" : "";
        const std::string root_name = ph.e.st[ph.root].ops.empty() ? ph.e.st[ph.root].text : ((ve.handed[ph.root] ? "h.t" : "t") + std::to_string(ph.root));
        o << "template <class PX, class PY, class PZ" << h_tmpl << "> __device__ __forceinline__ auto tape_dist" << f.suffix
          << "(PX px, PY py, PZ pz, const float* __restrict__ extra, uint32_t flags" << h_param << ")\n{\n" << head
          << vd.main << "    return as<T>(" << root_name << ");\n}\n";
        o << "template <class PX, class PY, class PZ" << h_tmpl << "> __device__ __forceinline__ auto tape_eval" << f.suffix
          << "(PX px, PY py, PZ pz, const float* __restrict__ extra, uint32_t flags" << h_param << ")\n{\n" << head
          << ve.main << phase2_for(ve.handed, tabc, &tab_index, nullptr)
          << "    return v4<T>(dir.x, dir.y, dir.z, as<T>(" << root_name << "));\n}\n";
    }
    o << "// deferred directions: " << paths.size() << " (primitive, path) pairs; " << ph.e.st.size() << " statements in phase 1; box pruning: "
      << prune.n_bits << " scopes\n";
    return true;
}

// The whole translation unit handed to hipRTC.  `meta` (may be NULL) <- what the host needs to know about it.
inline std::string specialised_source(const SpecProgram& p, bool allow_deferred, SpecMeta* meta = nullptr)
{
    std::ostringstream o, d;
    SpecMeta m;
    // (round 3's limit was 40: beyond it the second phase held too many registers.  With box pruning, walk coordinates the
    // compiler cannot hoist from and kept distances read again from the tables, planetary's 80 pairs run its float4 grid
    // in 1.0 ms where the plain form takes 4.7)
    static const size_t max_paths = [] { const char* e = std::getenv("HU_MAX_PATHS"); return e && *e ? (size_t)std::atoi(e) : (size_t)400; }();
    const bool ok = allow_deferred && emit_deferred(d, p, max_paths, &m);
    if (!ok) m = SpecMeta();
    m.deferred = ok;
    if (meta) *meta = m;
    o << "#include \"kernels.hpp\"\nnamespace sdfk {\nusing sdf::Rec;\n";
    if (ok) o << d.str();
    if (!ok || m.plain_in_place) emit_plain(o, p);
    o << "struct JitEval {\n    static constexpr bool kBricks = " << (ok ? "true" : "false") << ";\n"
      << "    const float* extra;\n"
      << "    uint32_t flags;   // sdf::kFlagInRange: the launch's coordinates cannot leave the fast range of sqrt_cr\n";
    if (ok) {
        if (m.plain_in_place)
            o << "    template <class T> __device__ __forceinline__ sdf::V4<T> operator()(T px, T py, T pz, void*) const\n"
              << "    { return tape_eval<T>(px, py, pz, extra); }\n"
              << "    template <class T> __device__ __forceinline__ T dist(T px, T py, T pz, void*) const\n"
              << "    { return tape_dist<T>(px, py, pz, extra); }\n";
        else
            o << "    template <class T> __device__ __forceinline__ sdf::V4<T> operator()(T px, T py, T pz, void*) const\n"
              << "    { return tape_eval(px, py, pz, extra, flags); }\n"
              << "    template <class T> __device__ __forceinline__ T dist(T px, T py, T pz, void*) const\n"
              << "    { return tape_dist(px, py, pz, extra, flags); }\n";
        o
          // the axis tables of the two walks: columns per axis, and the functions that fill one entry of each table
          << "    static constexpr int kTabXX = " << m.tabs[0] << ", kTabXY = " << m.tabs[1] << ", kTabXZ = " << m.tabs[2]
          << ", kPairXY = " << m.tabs[3] << ", kPairXZ = " << m.tabs[4] << ", kPairYZ = " << m.tabs[5] << ";\n"
          // box pruning: words of a box's mask, and the function that decides it (kernels.hpp k_box_masks)
          << "    static constexpr int kPruneWords = " << m.prune_words << ";\n"
          << "    static constexpr bool kPruneAll = " << (m.prune_all ? "true" : "false") << ";\n"
          << "    template <class PR> __device__ __forceinline__ void prune(float cx, float cy, float cz, float hx, float hy, float hz, PR& out) const\n"
          << "    { tape_prune(cx, cy, cz, hx, hy, hz, extra, out); }\n";
        for (const char* axis : {"x", "y", "z"})
            o << "    template <int S, class L> __device__ __forceinline__ void tab_x_" << axis << "(float p, L out) const\n"
              << "    { tape_tab_x_" << axis << "<S>(p, extra, flags, out); }\n";
        // one entry (a, b) of a pair table
        o << "    template <int S, class TB, class PR, class L> __device__ __forceinline__ void tab_x_xy(float a, float b, const TB& tb, const PR& pr, L out) const\n"
          << "    { tape_tab_x_xy<S>(a, b, 0.0f, extra, flags, tb, pr, out); }\n"
          << "    template <int S, class TB, class PR, class L> __device__ __forceinline__ void tab_x_xz(float a, float b, const TB& tb, const PR& pr, L out) const\n"
          << "    { tape_tab_x_xz<S>(a, 0.0f, b, extra, flags, tb, pr, out); }\n"
          << "    template <int S, class TB, class PR, class L> __device__ __forceinline__ void tab_x_yz(float a, float b, const TB& tb, const PR& pr, L out) const\n"
          << "    { tape_tab_x_yz<S>(0.0f, a, b, extra, flags, tb, pr, out); }\n"
          // what does not change along x, for the walks of a box with y and z fixed (kernels.hpp box_eval)
          << "    template <class PY, class PZ, class TB, class PR> __device__ __forceinline__ auto hoist_x(PY py, PZ pz, const TB& tb, const PR& pr) const\n"
          << "    { return tape_pre_x(0.0f, py, pz, extra, flags, tb, pr); }\n"
          << "    template <class PX, class PY, class PZ, class H, class TB, class PR> __device__ __forceinline__ auto eval_hoisted_x(PX px, PY py, PZ pz, const H& h, const TB& tb, const PR& pr) const\n"
          << "    { return tape_eval_x(px, py, pz, extra, flags, h, tb, pr); }\n"
          << "    template <class PX, class PY, class PZ, class H, class TB, class PR> __device__ __forceinline__ auto dist_hoisted_x(PX px, PY py, PZ pz, const H& h, const TB& tb, const PR& pr) const\n"
          << "    { return tape_dist_x(px, py, pz, extra, flags, h, tb, pr); }\n";
    } else {
        o << "    template <class T> __device__ __forceinline__ sdf::V4<T> operator()(T px, T py, T pz, void*) const\n"
          << "    { return tape_eval<T>(px, py, pz, extra); }\n"
          << "    template <class T> __device__ __forceinline__ T dist(T px, T py, T pz, void*) const\n"
          << "    { return tape_dist<T>(px, py, pz, extra); }\n";
    }
    o << "};\n}  // namespace sdfk\n";
    return o.str();
}

}  // namespace sdf
