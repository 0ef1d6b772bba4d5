// codecad_amd/csrc/cull.hpp
//
// Per-brick culling for the tape interpreter (host side: what the decoder works out once per tape).
//
// A CSG tape is mostly unions / intersections / subtractions of primitives, and inside a small brick of voxels most
// of their operands cannot win: a union's operand whose distance at the brick's centre exceeds the other's by more
// than the two can change across the brick is never the minimum there.  "Can change" needs no interval arithmetic:
// every value of a tape built from exact distance functions and rigid / uniformly scaling transformations is
// Lipschitz-continuous in the sample point, with a constant L this file derives per value (a primitive under frames
// scaled by J: L = J; a transformation_from multiplies by its distance scale; a select takes the larger; anything
// this file does not know -- twists, circular repetitions, gears, rounded blends -- gets L = infinity and is never
// culled).  A repetition folds space into one cell and is a translation only INSIDE a cell: the centre pass checks,
// per brick, that the brick keeps clear of the cell's faces (its extents in the repetition's frame: the record
// carries J in p[6], its number in p[7], "axes as the sample frame's" in p[8]), and a select under a repetition whose check failed is left alone there.
// So the dense grid kernels first run the ordinary distance-only program ONCE per brick, at its centre
// (kernels.hpp k_brick_keep), and note per plain select which operand is out: |a - b| > (La + Lb) * radius.
// The evaluation proper (k_grid_eval_culled) then skips, with one scalar test per record, every record whose value
// only feeds operands that are out -- the interpreter pays per record dispatched, so a skipped record is a real saving
// (the same idea in per-tape code lost to its straight-line code: DESIGN.md section 5).
//
// What the decoder adds to a program, all of it ignored by the plain interpreter:
//   * plain selects (r < 0) are numbered in program order, k = 0..15: bit 2k of a brick's `keep` word = "operand a may
//     win", bit 2k+1 = "operand b (the register operand) may win".  k is written into the record's fold word
//     (bits 10-14; a fused leaf's second select: bits 26-30; 31 = not culled) for the centre pass;
//   * per record two masks (CullInfo::masks): the record RUNS iff (keep & run) == run.  A value's mask is the AND over
//     its consumers of (consumer's mask | the select bit through which it is consumed): common to all, so whenever a
//     consumer runs, so does the producer; a record's mask is the AND over the values it defines.  A record that does
//     not run while the select that takes its value does -- (keep & live) == live, live = the mask without that
//     select's bit -- is replaced by its STAND-IN (CullInfo::stand_ins): "the constant +-infinity, stored where the
//     record would have stored its value", so that the select, which runs unchanged, takes the other operand.  Nothing
//     inside a record knows about culling: the interpreter is so sensitive to what surrounds its dispatch that tests
//     inside selects and fused leaves cost more than the skipped records saved.  The price: a
//     primitive fused into a leaf WITH its selects runs whenever the leaf's value is wanted;
//   * per select the sum La + Lb (CullInfo::lipschitz) and the repetitions it lies under.
// The reference has no counterpart (it evaluates every instruction for every sample: nodes/codegen.py:5-63).
#pragma once

#include <cmath>
#include <cstdint>
#include <limits>
#include <map>
#include <vector>

#include "tape.hpp"

namespace sdf {

constexpr uint32_t kSelNone = 31u;
constexpr int kSelShift1 = 10, kSelShift2 = 26;   // fold word: select numbers (tape_format.hpp: bits 0-9 and 16-25 are the folded moves)
constexpr int kMaxCullSelects = 16;

struct CullInfo {
    bool enabled = false;
    int n_selects = 0;                 // plain selects of the program (numbered ones: the first kMaxCullSelects)
    std::vector<uint32_t> masks;       // [record][2]: run, live (no stand-in: live == run)
    std::vector<Rec> stand_ins;        // per record: what runs in its place when only `live` is satisfied
    uint32_t n_records = 0;            // records up to and including _return
    float lipschitz[kMaxCullSelects];  // La + Lb per select; infinity: never culled
    uint32_t repetitions[kMaxCullSelects];   // per select: the repetitions (by number) its operands lie under
};

namespace cull_detail {

struct Value {
    bool point;
    double l;               // a point: scale J of its frame against the sample frame; a result: Lipschitz constant
    int in[2] = {-1, -1};   // operands (a select: a, b)
    int sel = -1;           // plain select number
    int kind = 0;           // a select: 1 union, 2 intersection, 3 subtraction
    bool in_leaf = false;   // a select fused into a leaf: its operand a is the leaf's own value and not culled
    int consumers = 0;
    uint32_t live = 0;      // the mask without the bit of the select that takes the value
    float stand_in = 0.0f;  // +-infinity: what that select must see when the value is out (0: no such select)
    uint32_t reps = 0;      // the repetitions (by number, at most 32) the value lies under
    bool aligned = false;   // a point whose frame has the sample frame's axes (only scalings, translations, repetitions so far)
    uint32_t need = 0;
    bool used = false;
};

inline double norm3(const float* p) { return std::sqrt((double)p[0] * p[0] + (double)p[1] * p[1] + (double)p[2] * p[2]); }

}  // namespace cull_detail

// Analyses one program of the interpreter (fused records; `typed`: the distance-only program, whose point and result
// slots are numbered separately) and writes the select numbers into its fold words.
inline void analyse_culling(std::vector<Rec>& prog, bool typed, CullInfo& out)
{
    using namespace cull_detail;
    using namespace fuse_detail;
    const double inf = std::numeric_limits<double>::infinity();
    std::vector<Value> vals;
    std::map<uint32_t, int> slots;
    auto key = [&](bool result_ns, uint32_t index) { return ((typed && result_ns) ? 0x10000u : 0u) | index; };
    auto slot_value = [&](bool result_ns, uint32_t index) {
        auto it = slots.find(key(result_ns, index));
        return it == slots.end() ? -1 : it->second;
    };
    auto l_of = [&](int v) { return v < 0 ? inf : vals[v].l; };
    auto make = [&](bool point, double l, int a, int b = -1, int sel = -1) {
        Value v;
        v.point = point;
        v.l = l;
        v.in[0] = a;
        v.in[1] = b;
        v.sel = sel;
        v.reps = (a >= 0 ? vals[a].reps : 0u) | (b >= 0 ? vals[b].reps : 0u);
        vals.push_back(v);
        return (int)vals.size() - 1;
    };
    const size_t n = prog.size();
    std::vector<std::vector<int>> defined(n);   // values a record defines
    std::vector<int> after(n, -1);              // the value in `last` after the record
    int cur = -1, root = -1, n_sel = 0, n_rep = 0;
    size_t end = n;
    for (size_t i = 0; i < n; ++i) {
        Rec& r = prog[i];
        const uint32_t op = op_of(r), reg = slot_of(r), fold = fold_word(r);
        const float* p = r.p;
        if (fold & kFoldLoad) cur = slot_value((fold & kFoldLoadResult) != 0, fold & 0xffu);
        auto def = [&](int v) { defined[i].push_back(v); return v; };
        uint32_t sel1 = kSelNone, sel2 = kSelNone;
        auto plain_select = [&](int a, int b, int kind, bool in_leaf) {
            const int k = n_sel++;
            const int v = def(make(false, std::fmax(l_of(a), l_of(b)), a, b, k));
            vals[v].kind = kind;
            vals[v].in_leaf = in_leaf;
            return std::make_pair(v, (uint32_t)(k < kMaxCullSelects ? k : (int)kSelNone));
        };
        if (op == OP_RETURN) { root = cur; end = i; break; }
        switch (op) {
        case OP_STORE: slots[key((r.hdr & kResultKind) != 0, reg)] = cur; break;
        case OP_LOAD: cur = slot_value((r.hdr & kResultKind) != 0, reg); break;
        case OPX_POINT:
            cur = def(make(true, 1.0, -1));
            vals[cur].aligned = true;
            break;
        case OPX_INIT_ROW_X: cur = def(make(true, 1.0, -1)); break;
        case OPX_INIT_ROWS_YZ: case OPX_TO_ROWS_YZ: cur = def(make(true, l_of(cur) * norm3(p), cur)); break;   // rotation times a uniform scale: every row has its norm
        case OPX_TO_ROW_X: cur = def(make(true, l_of(cur), cur)); break;   // x' parked in w: the frame changes with the second record
        case OPX_TO_SCALE: case OPX_TO_AXIS_X: case OPX_TO_AXIS_Y: case OPX_TO_AXIS_Z: {   // p[0] = |Q|^2: the scale along the axis and in the plane
            const bool aligned = op == OPX_TO_SCALE && cur >= 0 && vals[cur].aligned;
            cur = def(make(true, l_of(cur) * std::fabs((double)p[0]), cur));
            vals[cur].aligned = aligned;
            break;
        }
        case OP_REPETITION: {   // a translation inside a cell; the centre pass checks that the brick stays inside one
            const int id = n_rep++;
            const double j = l_of(cur);
            const bool aligned = cur >= 0 && vals[cur].aligned;
            cur = def(make(true, id < 32 ? j : inf, cur));
            vals[cur].aligned = aligned;
            r.p[8] = aligned ? 1.0f : 0.0f;   // the brick's extents along the frame's axes are its own, scaled: a box test instead of a ball
            if (id < 32) vals[cur].reps |= 1u << id;
            r.p[6] = std::isfinite(j) ? (float)j : std::numeric_limits<float>::infinity();
            const uint32_t number = id < 32 ? (uint32_t)id : 0u;
            std::memcpy(&r.p[7], &number, 4);
            break;
        }
        case OP_MIRROR: case OP_SYMMETRICAL_TO: case OP_REVOLUTION_TO:   // isometries / 1-Lipschitz maps of the point
            cur = def(make(cur >= 0 ? vals[cur].point : true, l_of(cur), cur));
            break;
        case OP_RECTANGLE: case OP_CIRCLE: case OP_SPHERE: case OP_HALF_SPACE: case OP_REGULAR_POLYGON2D: case OP_POLYGON2D:
            cur = def(make(false, l_of(cur), cur));   // exact distances in a frame scaled by J
            break;
        case OP_TRANSFORMATION_FROM: case OPX_FROM_SCALE: case OPX_FROM_AXIS_X: case OPX_FROM_AXIS_Y: case OPX_FROM_AXIS_Z:
            cur = def(make(false, l_of(cur) * std::fabs((double)p[5]), cur));
            break;
        case OPX_FROM_MATRIX: cur = def(make(false, l_of(cur) * std::fabs((double)p[9]), cur)); break;
        case OP_OFFSET: case OP_SHELL: cur = def(make(false, l_of(cur), cur)); break;
        case OP_EXTRUSION: {   // sqrt(max(d, 0)^2 + max(|z| - h, 0)^2) or the larger of the two: Lipschitz with the larger constant
            const int pt = slot_value(false, reg);
            cur = def(make(false, std::fmax(l_of(cur), l_of(pt)), cur, pt));
            break;
        }
        case OP_REVOLUTION_FROM: case OP_SYMMETRICAL_FROM:   // the distance passes through; the point only turns the direction
            cur = def(make(false, l_of(cur), cur, slot_value(false, reg)));
            break;
        case OP_UNION: case OP_INTERSECTION: case OP_SUBTRACTION: {
            const int b = slot_value(true, reg);
            if (p[0] < 0.0f) {
                auto vs = plain_select(cur, b, op == OP_UNION ? 1 : op == OP_INTERSECTION ? 2 : 3, false);
                cur = vs.first;
                sel1 = vs.second;
            } else {
                cur = def(make(false, inf, cur, b));   // a rounded blend: not a select, and no bound claimed
            }
            break;
        }
        case OPX_LEAF: {
            uint32_t c;
            std::memcpy(&c, &p[kLeafControl], 4);
            const int in = (c & kLeafSample) ? -1 : cur;
            const double j0 = (c & kLeafSample) ? 1.0 : l_of(cur);
            int pt = in;
            if ((c >> kLeafToShift) & 7u) pt = def(make(true, j0 * std::fabs((double)p[kLeafTo]), in));
            if (c & kLeafMidStore) slots[key(false, reg & 0xffu)] = pt;
            const bool from_first = ((c >> kLeafFromShift) & 7u) && !(c & kLeafFromLast);
            const double lp = (c & kLeafSample) && !((c >> kLeafToShift) & 7u) ? 1.0 : l_of(pt);
            int v = def(make(false, lp * (from_first ? std::fabs((double)p[kLeafScale]) : 1.0), pt));
            for (int s = 0; s < 2; ++s) {
                const uint32_t cb = c >> (s == 0 ? kLeafComb1Shift : kLeafComb2Shift);
                if ((cb & 3u) == 0u) continue;
                auto vs = plain_select(v, slot_value(true, (cb >> 2) & 0xffu), (int)(cb & 3u), true);
                v = vs.first;
                (s == 0 ? sel1 : sel2) = vs.second;
            }
            if (c & kLeafFromLast) v = def(make(false, l_of(v) * std::fabs((double)p[kLeafScale]), v));
            cur = v;
            break;
        }
        default:   // repetitions, twists, the gear, general quaternions ...: a value nothing is claimed about
            if (rec_arity(op) == 2) cur = def(make(produces_point(op), inf, cur, slot_value(!reads_point_operand(op), reg)));
            else cur = def(make(produces_point(op), inf, cur));
            break;
        }
        fold_word(r) = (fold & ~((31u << kSelShift1) | (31u << kSelShift2))) | (sel1 << kSelShift1) | (sel2 << kSelShift2);
        if (fold & kFoldStore) slots[key((fold & kFoldStoreResult) != 0, (fold >> 16) & 0xffu)] = cur;
        after[i] = cur;
    }
    // who needs what: consumers come after producers, so one pass from the back.  A select's bit goes to an operand
    // only if the select is that value's only consumer (then "the value is out" has one meaning) and the operand is
    // not a fused leaf's own value.
    for (const Value& v : vals)
        for (int side = 0; side < 2; ++side)
            if (v.in[side] >= 0) ++vals[v.in[side]].consumers;
    if (root >= 0) { vals[root].used = true; ++vals[root].consumers; }
    for (int id = (int)vals.size() - 1; id >= 0; --id) {
        const Value v = vals[id];
        if (!v.used) continue;
        for (int side = 0; side < 2; ++side) {
            const int u = v.in[side];
            if (u < 0) continue;
            const bool culled_through = v.sel >= 0 && v.sel < kMaxCullSelects && vals[u].consumers == 1 && !(v.in_leaf && side == 0) &&
                                        !vals[u].point;
            const uint32_t through = v.need | (culled_through ? 1u << (2 * v.sel + side) : 0u);
            if (!vals[u].used) { vals[u].used = true; vals[u].need = through; }
            else vals[u].need &= through;
            if (culled_through) {
                vals[u].live = v.need;
                // what the select must see: union takes the minimum of (a, b), intersection of (-a, -b), subtraction of (-a, b)
                const bool minus = v.kind == 2 || (v.kind == 3 && side == 0);
                vals[u].stand_in = minus ? -std::numeric_limits<float>::infinity() : std::numeric_limits<float>::infinity();
            }
        }
    }
    out.n_selects = n_sel;
    out.masks.assign(2 * n, 0u);
    out.stand_ins.assign(n, Rec{});
    out.n_records = (uint32_t)(end < n ? end + 1 : n);
    for (size_t i = 0; i < end; ++i) {
        uint32_t run = 0xffffffffu;
        bool any = false;
        for (int v : defined[i]) {
            run &= vals[v].used ? vals[v].need : 0u;   // a value nobody reads: nothing is claimed, the record just runs
            any = true;
        }
        const int moved = after[i];   // a _store / _load moves a value some record defined earlier: it runs while that value, or its stand-in, is wanted
        if (!any) run = (moved >= 0 && vals[moved].used) ? (vals[moved].stand_in != 0.0f ? vals[moved].live : vals[moved].need) : 0u;
        uint32_t live = run;
        // the record's final value has a stand-in, and nothing else the record defines is wanted more widely
        if (any && moved >= 0 && vals[moved].used && vals[moved].stand_in != 0.0f && run == vals[moved].need) {
            live = vals[moved].live;
            Rec& alt = out.stand_ins[i];
            alt.hdr = OPX_CONST | (prog[i].hdr & kResultKind);
            alt.p[0] = vals[moved].stand_in;
            fold_word(alt) = fold_word(prog[i]) & (kFoldStore | kFoldStoreResult | 0xff0000u);
        }
        out.masks[2 * i] = run;
        out.masks[2 * i + 1] = live;
    }
    for (int k = 0; k < kMaxCullSelects; ++k) {
        out.lipschitz[k] = std::numeric_limits<float>::infinity();
        out.repetitions[k] = 0u;
    }
    bool any_finite = false;
    for (const Value& v : vals)
        if (v.sel >= 0 && v.sel < kMaxCullSelects) {
            const double sum = l_of(v.in[0]) + l_of(v.in[1]);
            // (rounded up: the threshold must not fall below the exact product)
            out.lipschitz[v.sel] = std::isfinite(sum) ? std::nextafter((float)sum, std::numeric_limits<float>::infinity()) : std::numeric_limits<float>::infinity();
            out.repetitions[v.sel] = v.reps;
            any_finite = any_finite || std::isfinite(sum);
        }
    out.enabled = any_finite;
}

}  // namespace sdf
