// codecad_amd/csrc/tape.hpp
//
// The float32 instruction tape (reference nodes/program.py:55-76, opcode table
// nodes/node.py:12-56) and its pre-decoded device form.
//
// Reference tape: a flat float32 array; each instruction is one float
// `opcode*512 + secondaryRegister` followed by its parameters; the interpreter
// (reference nodes/codegen.py:5-63) re-derives opcode/register with an integer divide and
// modulo per instruction per work-item and reads parameters one float at a time.
//
// Device form: the tape is a straight line (no branches), identical for every voxel, so it
// is decoded ONCE on the host at upload into fixed 48-byte records the wave fetches with
// scalar loads (s_load_dwordx8 + x4, prefetched one record ahead), with every
// tape-constant subexpression folded in (|q|^2, 1/|q|^2, w^2-|v|^2, 1/spacing, gear
// constants ...) using the same IEEE operations the per-voxel code would use.
#pragma once

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "tape_format.hpp"

namespace sdf {

// (parameter count, arity) per opcode -- reference nodes/node.py:15-53
struct OpInfo { const char* name; int params; int arity; };
inline const OpInfo& op_info(uint32_t op)
{
    static const OpInfo table[OP_COUNT] = {
        {"_return", 0, 1}, {"_store", 0, 1}, {"_load", 0, 1}, {"rectangle", 2, 1},
        {"circle", 1, 1}, {"regular_polygon2d", 2, 1}, {"polygon2d", kVariableParams, 1},
        {"sphere", 1, 1}, {"half_space", 0, 1}, {"revolution_to", 0, 1},
        {"twist_revolution_to", 2, 1}, {"initial_transformation_to", 7, 0},
        {"transformation_to", 7, 1}, {"transformation_from", 4, 1}, {"mirror", 0, 1},
        {"symmetrical_to", 0, 1}, {"offset", 1, 1}, {"shell", 1, 1}, {"repetition", 3, 1},
        {"circular_repetition_to", 1, 1}, {"circular_repetition_from", 1, 2},
        {"involute_gear", 2, 1}, {"extrusion", 1, 2}, {"revolution_from", 0, 2},
        {"twist_revolution_from", 3, 2}, {"symmetrical_from", 0, 2}, {"union", 1, 2},
        {"intersection", 1, 2}, {"subtraction", 1, 2}};
    return table[op];
}

inline int rec_arity(uint32_t op) { return op < OP_COUNT ? op_info(op).arity : 1; }

inline float bits_f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

struct DecodedTape {
    std::vector<Rec> recs;        // full program: every slot is a float4 (direction + distance / point)
    std::vector<Rec> recs_do;     // distance-only program: point slots (float4) and result slots (float)
    std::vector<Rec> fused;       // the interpreter's form of `recs`: transformed primitives as single records (fuse_leaves)
    std::vector<Rec> fused_do;    // ... and of `recs_do`
    std::vector<float> extra;     // polygon2d vertex data
    int n_instructions = 0;       // instructions of the TAPE (the decoder may add internal records)
    int n_regs = 0;               // highest register index of the TAPE + 1 (what the reference allocates)
    int n_slots = 0;              // float4 slots the full program needs
    int n_point_slots = 0;        // distance-only program: float4 slots
    int n_result_slots = 0;       // distance-only program: float slots
    bool direction_feeds_distance = false;  // any rounded union (r >= 0): no distance-only program
};

// Does the op leave a POINT (xyz meaningful) or a RESULT (direction + distance) in lastValue?
inline bool produces_point(uint32_t op)
{
    switch (op) {
    case OP_INITIAL_TRANSFORMATION_TO: case OP_TRANSFORMATION_TO: case OP_SYMMETRICAL_TO: case OP_REPETITION:
    case OP_CIRCULAR_REPETITION_TO: case OP_REVOLUTION_TO: case OP_TWIST_REVOLUTION_TO:
    case OPX_POINT: case OPX_TO_SCALE: case OPX_TO_AXIS_X: case OPX_TO_AXIS_Y: case OPX_TO_AXIS_Z:
    case OPX_TO_ROW_X: case OPX_TO_ROWS_YZ: case OPX_INIT_ROW_X: case OPX_INIT_ROWS_YZ:
        return true;
    default: return false;
    }
}
// Binary ops whose register operand is a point (the sample coordinates), not a result.
inline bool reads_point_operand(uint32_t op)
{
    return op == OP_EXTRUSION || op == OP_REVOLUTION_FROM || op == OP_TWIST_REVOLUTION_FROM ||
           op == OP_SYMMETRICAL_FROM || op == OP_CIRCULAR_REPETITION_FROM;
}

// Register renaming.  The tape's register numbers come from the reference's scheduler, which
// leaks registers (the planetary assembly uses 40 while at most 10 values are ever live), and
// the interpreter keeps registers in LDS, where their number sets the occupancy.  Each _store
// starts a live range that ends at the last read before the next _store to the same register;
// ranges are packed greedily into slots.  For the distance-only program ranges are typed:
// a range holding a RESULT only needs its distance (4 bytes instead of 16).
// Returns false when the tape reads a register before writing it or mixes kinds in a way the
// distance-only program cannot represent (then only the full program is used).
inline bool allocate_slots(DecodedTape& d, std::string& err)
{
    struct Range { int first, last; bool result; int slot_full, slot_do; };
    const int n = (int)d.recs.size();
    std::vector<Range> ranges;
    std::vector<int> open_range(kRefRegisterCount, -1);  // register -> index into ranges
    std::vector<int> range_of(n, -1);                    // instruction -> range it writes/reads
    bool last_is_result = false;
    bool do_ok = !d.direction_feeds_distance;
    for (int i = 0; i < n; ++i) {
        const uint32_t op = d.recs[i].hdr & 0xffu, reg = d.recs[i].hdr >> 8;
        const int arity = rec_arity(op);
        if (op == OP_STORE) {
            ranges.push_back({i, i, last_is_result, -1, -1});
            open_range[reg] = (int)ranges.size() - 1;
            range_of[i] = open_range[reg];
        } else if (op == OP_LOAD || arity == 2) {
            const int r = open_range[reg];
            if (r < 0) { err = std::string(op < OP_COUNT ? op_info(op).name : "op") + " reads register " + std::to_string(reg) + " before any _store"; return false; }
            ranges[r].last = i;
            range_of[i] = r;
            if (op == OP_LOAD) last_is_result = ranges[r].result;
            else {
                if (ranges[r].result == reads_point_operand(op)) do_ok = false;  // kind mismatch
                last_is_result = true;
            }
        } else if (op == OP_MIRROR || op == OP_RETURN) {
            // keeps the kind
        } else {
            last_is_result = !produces_point(op);
        }
        if (op == OP_RETURN) break;
    }
    // greedy interval packing in order of start (ranges are already sorted by `first`)
    auto pack = [&](bool typed, bool want_result, int Range::*slot) {
        std::vector<int> busy_until;  // per slot
        int used = 0;
        for (auto& r : ranges) {
            if (typed && r.result != want_result) continue;
            int s = 0;
            while (s < (int)busy_until.size() && busy_until[s] >= r.first) ++s;
            if (s == (int)busy_until.size()) busy_until.push_back(-1);
            busy_until[s] = r.last;
            r.*slot = s;
            used = (int)busy_until.size();
        }
        return used;
    };
    d.n_slots = pack(false, false, &Range::slot_full);
    d.recs_do.clear();
    if (do_ok) {
        d.n_point_slots = pack(true, false, &Range::slot_do);
        d.n_result_slots = pack(true, true, &Range::slot_do);
        d.recs_do = d.recs;
    }
    for (int i = 0; i < n; ++i) {
        if (range_of[i] < 0) continue;
        const Range& r = ranges[range_of[i]];
        const uint32_t op = d.recs[i].hdr & 0xffu;
        d.recs[i].hdr = op | ((uint32_t)r.slot_full << 8);
        if (do_ok) d.recs_do[i].hdr = op | ((uint32_t)r.slot_do << 8) | (r.result ? kResultKind : 0u);
    }
    return true;
}

// quaternion helpers used for the folded constants; same op order as the kernels/oracle
inline float q_k(const float* q) { return fma_(q[3], q[3], -fma_(q[2], q[2], fma_(q[1], q[1], q[0] * q[0]))); }
// Rotation about one coordinate axis (quaternion with a single non-zero vector component `q`, scalar
// part `w`, not normalised: |Q|^2 is the transform's scale), written as the 2x2 rotation-and-scale it is:
//   along the axis  A = w^2 + q^2,   in the plane  B = w^2 - q^2 (0 for a quarter turn),  C = 2 q w.
// Folded in double (every product is exact there) and rounded once; `div` = 1 for transformation_to,
// |Q|^2 for transformation_from (which returns unit directions).  The oracle folds the same way.
inline void axis_constants(float q, float w, double div, float& A, float& B, float& C)
{
    const double qq = (double)q * (double)q, ww = (double)w * (double)w;
    A = (float)((ww + qq) / div);
    B = (float)((ww - qq) / div);
    C = (float)((2.0 * ((double)q * (double)w)) / div);
}
// A general quaternion Q = (x, y, z, w), not normalised (|Q|^2 is the transform's scale), as its matrix
//   R = (w^2 - |v|^2) I + 2 v v^T + 2 w [v]x,      m[3*row + column],
// folded in double in exactly this order and rounded once; `div` as in axis_constants.  The oracle folds
// the same way (sdf_oracle.c matrix_constants).
inline void matrix_constants(const float* q, double div, float m[9])
{
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double xx = x * x, yy = y * y, zz = z * z, ww = w * w;
    const double xy = x * y, xz = x * z, yz = y * z, wx = w * x, wy = w * y, wz = w * z;
    const double k = ww - ((xx + yy) + zz);
    const double e[9] = {k + 2.0 * xx,     2.0 * (xy - wz),  2.0 * (xz + wy),
                         2.0 * (xy + wz),  k + 2.0 * yy,     2.0 * (yz - wx),
                         2.0 * (xz - wy),  2.0 * (yz + wx),  k + 2.0 * zz};
    for (int i = 0; i < 9; ++i) m[i] = (float)(e[i] / div);
}
inline float q_scale(const float* q) { return fma_(q[3], q[3], fma_(q[2], q[2], fma_(q[1], q[1], q[0] * q[0]))); }

// Fold _load into the record that follows it and _store into the record before it (tape_format.hpp
// kFold*).  Order of effects is unchanged: the load happens first, the store last.
inline void fold_moves(std::vector<Rec>& prog)
{
#ifndef SDF_FOLD_MOVES
#define SDF_FOLD_MOVES 1
#endif
    if (!SDF_FOLD_MOVES) return;
    auto fold_of = [](Rec& r) -> uint32_t& { return reinterpret_cast<uint32_t&>(r.p[kFoldParam]); };
    std::vector<Rec> out;
    out.reserve(prog.size());
    uint32_t pending_load = 0;  // fold word bits for the next real record
    bool last_foldable = false; // the previous emitted record can still take a store
    for (size_t i = 0; i < prog.size(); ++i) {
        Rec r = prog[i];
        const uint32_t op = r.hdr & 0xffu, slot = (r.hdr >> 8) & 0xffffu;
        const bool result = (r.hdr & kResultKind) != 0;
        // never into _return: the per-tape code generator stops AT the return record and would drop the load
        const uint32_t next_op = i + 1 < prog.size() ? (prog[i + 1].hdr & 0xffu) : (uint32_t)OP_RETURN;
        const bool next_is_real = next_op != OP_STORE && next_op != OP_LOAD && next_op != OP_RETURN;
        if (op == OP_LOAD && slot < 256u && next_is_real && !pending_load) {
            pending_load = kFoldLoad | (result ? kFoldLoadResult : 0u) | slot;
            last_foldable = false;
            continue;
        }
        if (op == OP_STORE && slot < 256u && last_foldable && !(fold_of(out.back()) & kFoldStore)) {
            fold_of(out.back()) |= kFoldStore | (result ? kFoldStoreResult : 0u) | (slot << 16);
            continue;
        }
        if (op != OP_STORE && op != OP_LOAD) {
            fold_of(r) = pending_load;
            pending_load = 0;
        }
        out.push_back(r);
        last_foldable = op != OP_STORE && op != OP_LOAD && op != OP_RETURN;
        if (op == OP_RETURN) break;
    }
    prog.swap(out);
}

// ---------------------------------------------------------------------------------------------------------------
// Superinstructions for the interpreter (tape_format.hpp OPX_LEAF).  The interpreter is bound by its SCALAR work:
// fetch, decode, a compare tree and ~10 branches per record (rocprofv3: 1136 scalar instructions and 599 branches
// per wavefront for sponge(4)'s 56 records, the one scalar unit of a CU saturated before its four vector units).
// CAD tapes are made of one pattern -- a primitive under a transformation, combined into an accumulator -- and
// the decoder recognises it: [point] to primitive [extrusion] [from] [select] [select] becomes ONE record whose
// parts run back to back, chosen by a few uniform bit tests.  Only what is provably the same computation is fused:
// the parts must be adjacent, `last` must flow straight through them (a folded load only on the first, a folded
// store only on the last; the store of the transformed point that an extrusion reads back is kept, or dropped when
// nothing else reads it), selects must be plain (r < 0).  Per-tape code is generated from the unfused records.
// ---------------------------------------------------------------------------------------------------------------
#ifndef SDF_FUSE_LEAVES
#define SDF_FUSE_LEAVES 1
#endif
namespace fuse_detail {
inline uint32_t& fold_word(Rec& r) { return reinterpret_cast<uint32_t&>(r.p[kFoldParam]); }
inline uint32_t fold_word(const Rec& r) { uint32_t f; std::memcpy(&f, &r.p[kFoldParam], 4); return f; }
inline uint32_t op_of(const Rec& r) { return r.hdr & 0xffu; }
inline uint32_t slot_of(const Rec& r) { return (r.hdr >> 8) & 0xffffu; }
inline int to_kind(uint32_t op) { return op == OPX_TO_SCALE ? 1 : op == OPX_TO_AXIS_X ? 2 : op == OPX_TO_AXIS_Y ? 3 : op == OPX_TO_AXIS_Z ? 4 : 0; }
inline int from_kind(uint32_t op) { return op == OPX_FROM_SCALE ? 1 : op == OPX_FROM_AXIS_X ? 2 : op == OPX_FROM_AXIS_Y ? 3 : op == OPX_FROM_AXIS_Z ? 4 : 0; }
inline int prim_kind(uint32_t op) { return op == OP_RECTANGLE ? LEAF_RECTANGLE : op == OP_CIRCLE ? LEAF_CIRCLE : op == OP_SPHERE ? LEAF_SPHERE : op == OP_HALF_SPACE ? LEAF_HALF_SPACE : -1; }
inline int select_kind(const Rec& r)
{
    const uint32_t op = op_of(r);
    if (!(r.p[0] < 0.0f)) return 0;   // rounded (or NaN): not fused
    return op == OP_UNION ? 1 : op == OP_INTERSECTION ? 2 : op == OP_SUBTRACTION ? 3 : 0;
}
// Is POINT slot `slot` read after record `from` before it is written again?  (typed: in the distance-only program
// point slots and result slots are numbered separately)
inline bool point_slot_read_later(const std::vector<Rec>& prog, size_t from, uint32_t slot, bool typed)
{
    for (size_t i = from; i < prog.size(); ++i) {
        const Rec& r = prog[i];
        const uint32_t op = op_of(r), f = fold_word(r);
        const bool is_result_access = typed && (r.hdr & kResultKind);
        if ((f & kFoldLoad) && (f & 0xffu) == slot && !(typed && (f & kFoldLoadResult))) return true;
        if (op == OP_LOAD && slot_of(r) == slot && !is_result_access) return true;
        if (rec_arity(op) == 2 && slot_of(r) == slot && reads_point_operand(op)) return true;
        if (op == OP_STORE && slot_of(r) == slot && !is_result_access) return false;
        if ((f & kFoldStore) && ((f >> 16) & 0xffu) == slot && !(typed && (f & kFoldStoreResult))) return false;
        if (op == OP_RETURN) return false;
    }
    return false;
}
}  // namespace fuse_detail

inline void fuse_leaves(const std::vector<Rec>& prog, bool typed, std::vector<Rec>& out)
{
    using namespace fuse_detail;
    out.clear();
    if (!SDF_FUSE_LEAVES || SDF_REC_DWORDS < 16) { out = prog; return; }
    const size_t n = prog.size();
    size_t i = 0;
    while (i < n) {
        // ---- try to match a leaf starting at record i
        size_t j = i;
        Rec leaf;
        std::memset(&leaf, 0, sizeof(leaf));
        uint32_t control = 0, fold = 0, mid_slot = 0;
        bool ok = true, closed = false;      // closed: a folded store ended the match
        int parts = 0;
        bool have_mid_store = false;
        int64_t point_slot = -1;             // the slot the point in `last` was loaded from (for a following extrusion)
        auto take_fold = [&](const Rec& r, bool first) {
            const uint32_t f = fold_word(r);
            if ((f & kFoldLoad) && !first) return false;
            if (first) fold |= f & (kFoldLoad | kFoldLoadResult | 0xffu);
            return true;
        };
        // [sample point]
        if (op_of(prog[j]) == OPX_POINT && !(fold_word(prog[j]) & kFoldStore) && j + 1 < n && to_kind(op_of(prog[j + 1])) &&
            !(fold_word(prog[j + 1]) & kFoldLoad)) {
            control |= kLeafSample;
            ++j;
            ++parts;
        }
        // [to]
        if (j < n && to_kind(op_of(prog[j]))) {
            const Rec& r = prog[j];
            if (!take_fold(r, j == i)) ok = false;
            else {
                control |= (uint32_t)to_kind(op_of(r)) << kLeafToShift;
                leaf.p[kLeafTo + 0] = r.p[0]; leaf.p[kLeafTo + 1] = r.p[1]; leaf.p[kLeafTo + 2] = r.p[2];
                leaf.p[kLeafTo + 3] = r.p[4]; leaf.p[kLeafTo + 4] = r.p[5]; leaf.p[kLeafTo + 5] = r.p[6];
                const uint32_t f = fold_word(r);
                if (f & kFoldStore) {
                    if (typed && (f & kFoldStoreResult)) ok = false;
                    have_mid_store = true;
                    mid_slot = (f >> 16) & 0xffu;
                }
                ++j;
                ++parts;
            }
        } else if (j == i && j < n) {
            const uint32_t f = fold_word(prog[j]);
            if ((f & kFoldLoad) && !(typed && (f & kFoldLoadResult))) point_slot = f & 0xffu;
        }
        // primitive (required)
        if (ok && j < n && prim_kind(op_of(prog[j])) >= 0) {
            const Rec& r = prog[j];
            if (!take_fold(r, j == i)) ok = false;
            else {
                control |= (uint32_t)prim_kind(op_of(r)) << kLeafPrimShift;
                leaf.p[kLeafPrim] = r.p[0];
                leaf.p[kLeafPrim + 1] = r.p[1];
                if (fold_word(r) & kFoldStore) { fold |= fold_word(r) & (kFoldStore | kFoldStoreResult | 0xff0000u); closed = true; }
                ++j;
                ++parts;
            }
        } else ok = false;
        // [extrusion]: its point operand must be the point this leaf computed (or was handed)
        if (ok && !closed && j < n && op_of(prog[j]) == OP_EXTRUSION && !(fold_word(prog[j]) & kFoldLoad)) {
            const Rec& r = prog[j];
            const bool same_point = (have_mid_store && slot_of(r) == mid_slot) ||
                                    (!have_mid_store && !(control & (7u << kLeafToShift)) && !(control & kLeafSample) &&
                                     point_slot >= 0 && slot_of(r) == (uint32_t)point_slot);
            if (same_point) {
                control |= kLeafExtrusion;
                leaf.p[kLeafExtrude] = r.p[0];
                if (fold_word(r) & kFoldStore) { fold |= fold_word(r) & (kFoldStore | kFoldStoreResult | 0xff0000u); closed = true; }
                ++j;
                ++parts;
            }
        }
        // [from]
        if (ok && !closed && j < n && from_kind(op_of(prog[j])) && !(fold_word(prog[j]) & kFoldLoad)) {
            const Rec& r = prog[j];
            control |= (uint32_t)from_kind(op_of(r)) << kLeafFromShift;
            leaf.p[kLeafFrom + 0] = r.p[0]; leaf.p[kLeafFrom + 1] = r.p[1]; leaf.p[kLeafFrom + 2] = r.p[2];
            leaf.p[kLeafScale] = r.p[5];
            if (fold_word(r) & kFoldStore) { fold |= fold_word(r) & (kFoldStore | kFoldStoreResult | 0xff0000u); closed = true; }
            ++j;
            ++parts;
        }
        // [select] [select]
        for (int k = 0; k < 2 && ok && !closed && j < n; ++k) {
            const Rec& r = prog[j];
            const int kind = select_kind(r);
            if (!kind || (fold_word(r) & kFoldLoad) || slot_of(r) >= 256u) break;
            control |= ((uint32_t)kind | (slot_of(r) << 2)) << (k == 0 ? kLeafComb1Shift : kLeafComb2Shift);
            if (fold_word(r) & kFoldStore) { fold |= fold_word(r) & (kFoldStore | kFoldStoreResult | 0xff0000u); closed = true; }
            ++j;
            ++parts;
        }
        // [from: a scaling of the combined value] -- `to prim select select from_scale`, the tail of a repeated cross
        if (SDF_LEAF_FROM_LAST && ok && !closed && j < n && !(control & (7u << kLeafFromShift)) && (control & (3u << kLeafComb1Shift)) &&
            op_of(prog[j]) == OPX_FROM_SCALE && !(fold_word(prog[j]) & kFoldLoad)) {
            const Rec& r = prog[j];
            control |= (1u << kLeafFromShift) | kLeafFromLast;
            leaf.p[kLeafFrom + 0] = r.p[0]; leaf.p[kLeafFrom + 1] = r.p[1]; leaf.p[kLeafFrom + 2] = r.p[2];
            leaf.p[kLeafScale] = r.p[5];
            if (fold_word(r) & kFoldStore) { fold |= fold_word(r) & (kFoldStore | kFoldStoreResult | 0xff0000u); closed = true; }
            ++j;
            ++parts;
        }
        // The folded store of the transformed point: dropped when nothing reads the slot afterwards (the fused
        // extrusion has the point in registers), which includes the leaf's own final store overwriting it.
        if (ok && have_mid_store) {
            const bool own_store_overwrites = (fold & kFoldStore) && ((fold >> 16) & 0xffu) == mid_slot &&
                                              !(typed && (fold & kFoldStoreResult));
            if (!own_store_overwrites && point_slot_read_later(prog, j, mid_slot, typed)) control |= kLeafMidStore;
        }
        if (ok && parts >= 2) {
            leaf.hdr = OPX_LEAF | (mid_slot << 8);
            fold_word(leaf) = fold;
            std::memcpy(&leaf.p[kLeafControl], &control, 4);
            out.push_back(leaf);
            i = j;
        } else {
            out.push_back(prog[i]);
            ++i;
        }
    }
}

// Validate + decode.  Returns "" on success, otherwise the reason the tape is malformed.
inline std::string decode_tape(const float* tape, size_t n, DecodedTape& out)
{
    size_t pc = 0;
    bool returned = false;
    out = DecodedTape();
    while (pc < n) {
        float word = tape[pc++];
        if (!(word >= 0.0f) || word >= float(OP_COUNT * kRefRegisterCount) || word != float(uint32_t(word)))
            return "instruction word " + std::to_string(word) + " at float " + std::to_string(pc - 1) + " is not a valid opcode*512+register";
        uint32_t ins = uint32_t(word);
        uint32_t op = ins / kRefRegisterCount, reg = ins % kRefRegisterCount;
        const OpInfo& info = op_info(op);
        Rec r;
        std::memset(&r, 0, sizeof(r));
        r.hdr = op | (reg << 8);
        int np = info.params;
        if (np == kVariableParams) {
            if (pc >= n) return "truncated polygon2d";
            float cnt = tape[pc];
            if (!(cnt >= 3.0f) || cnt > 1.0e6f || cnt != float(uint32_t(cnt))) return "bad polygon2d point count";
            uint32_t count = uint32_t(cnt);
            if (pc + 1 + 2 * size_t(count) > n) return "truncated polygon2d points";
            r.p[0] = bits_f(count);
            r.p[1] = bits_f(uint32_t(out.extra.size()));
            out.extra.insert(out.extra.end(), tape + pc + 1, tape + pc + 1 + 2 * size_t(count));
            pc += 1 + 2 * size_t(count);
        } else {
            if (pc + size_t(np) > n) return std::string("truncated parameters of ") + info.name;
            const float* p = tape + pc;
            for (int i = 0; i < np; ++i) r.p[i] = p[i];
            switch (op) {
            case OP_REGULAR_POLYGON2D:
                r.p[2] = 2.0f * p[0];
                r.p[3] = p[1] * sin_(p[0]);
                r.p[4] = -(p[1] * cos_(p[0]));
                break;
            case OP_INITIAL_TRANSFORMATION_TO:
            case OP_TRANSFORMATION_TO: {
                r.p[7] = q_k(p);
                // canonical arithmetic: a zero offset component is +0, so a transformed coordinate is
                // never -0 whichever form computes it (the reference builds with -cl-no-signed-zeros)
                for (int i = 4; i < 7; ++i) r.p[i] = p[i] + 0.0f;
                const bool zx = p[0] == 0.0f, zy = p[1] == 0.0f, zz = p[2] == 0.0f;
                uint32_t special = op;
                if (!SDF_TO_SPECIAL) special = op;
                else if (zx && zy && zz) special = OPX_TO_SCALE;
                else if (zy && zz) special = OPX_TO_AXIS_X;
                else if (zx && zz) special = OPX_TO_AXIS_Y;
                else if (zx && zy) special = OPX_TO_AXIS_Z;
                if (special != op) {
                    if (op == OP_INITIAL_TRANSFORMATION_TO) {
                        Rec point;
                        std::memset(&point, 0, sizeof(point));
                        point.hdr = OPX_POINT;
                        out.recs.push_back(point);
                    }
                    r.hdr = special | (reg << 8);
                    const float q = special == OPX_TO_AXIS_X ? p[0] : special == OPX_TO_AXIS_Y ? p[1] : special == OPX_TO_AXIS_Z ? p[2] : 0.0f;
                    axis_constants(q, p[3], 1.0, r.p[0], r.p[1], r.p[2]);   // p[0..2] = A, B, C (p[3], p[7] unused)
                } else if (SDF_TO_SPECIAL) {
                    // general quaternion: two records (tape_format.hpp OPX_TO_ROW_X / OPX_TO_ROWS_YZ)
                    float m[9];
                    matrix_constants(p, 1.0, m);
                    const float ox = r.p[4], oy = r.p[5], oz = r.p[6];
                    const bool initial = op == OP_INITIAL_TRANSFORMATION_TO;
                    Rec first;
                    std::memset(&first, 0, sizeof(first));
                    first.hdr = (initial ? OPX_INIT_ROW_X : OPX_TO_ROW_X) | (reg << 8);
                    first.p[0] = m[0]; first.p[1] = m[1]; first.p[2] = m[2]; first.p[3] = ox;
                    out.recs.push_back(first);
                    std::memset(&r, 0, sizeof(r));
                    r.hdr = (initial ? OPX_INIT_ROWS_YZ : OPX_TO_ROWS_YZ) | (reg << 8);
                    r.p[0] = m[3]; r.p[1] = m[4]; r.p[2] = m[5]; r.p[3] = oy;
                    r.p[4] = m[6]; r.p[5] = m[7]; r.p[6] = m[8]; r.p[7] = oz;
                }
                break;
            }
            case OP_TRANSFORMATION_FROM: {
                float scale = q_scale(p);
                r.p[4] = q_k(p);
                r.p[5] = scale;
                r.p[6] = 1.0f / scale;
                const bool zx = p[0] == 0.0f, zy = p[1] == 0.0f, zz = p[2] == 0.0f;
                uint32_t special = op;
                if (!SDF_FROM_SPECIAL) special = op;
                else if (zx && zy && zz) special = OPX_FROM_SCALE;
                else if (zy && zz) special = OPX_FROM_AXIS_X;
                else if (zx && zz) special = OPX_FROM_AXIS_Y;
                else if (zx && zy) special = OPX_FROM_AXIS_Z;
                r.hdr = special | (reg << 8);
                if (special != op) {
                    const float q = special == OPX_FROM_AXIS_X ? p[0] : special == OPX_FROM_AXIS_Y ? p[1] : special == OPX_FROM_AXIS_Z ? p[2] : 0.0f;
                    axis_constants(q, p[3], (double)scale, r.p[0], r.p[1], r.p[2]);   // p[0..2] = A, B, C over |Q|^2; p[5] = scale
                } else if (SDF_FROM_SPECIAL) {
                    float m[9];
                    matrix_constants(p, (double)scale, m);
                    std::memset(&r, 0, sizeof(r));
                    r.hdr = OPX_FROM_MATRIX | (reg << 8);
                    for (int i = 0; i < 9; ++i) r.p[i] = m[i];
                    r.p[9] = scale;
                }
                break;
            }
            case OP_REPETITION:
                for (int i = 0; i < 3; ++i) r.p[3 + i] = 1.0f / p[i];
                break;
            case OP_CIRCULAR_REPETITION_TO:
            case OP_CIRCULAR_REPETITION_FROM:
                r.p[1] = 2.0f * p[0];
                break;
            case OP_INVOLUTE_GEAR: {
                float base = cos_(p[1]);
                float tooth = kPi / p[0];
                r.p[2] = base;
                r.p[3] = tooth;
                r.p[4] = (tooth / 2.0f + tan_(p[1])) - p[1];
                r.p[5] = 2.0f * tooth;
                r.p[6] = -(base * base);
                break;
            }
            case OP_TWIST_REVOLUTION_FROM: {
                float lip = (((p[1] - p[0]) * 2.0f) *
                             sin_(__builtin_fminf(kPi, (kPi2 * kPi2) / abs_(p[2])))) / p[0];
                r.p[3] = 0.05f * p[1];
                r.p[4] = p[1] - p[0];
                r.p[5] = __builtin_fminf(1.0f, lip);
                break;
            }
            case OP_UNION:
            case OP_INTERSECTION:
            case OP_SUBTRACTION:
                if (p[0] >= 0.0f) out.direction_feeds_distance = true;
                break;
            default: break;
            }
            pc += size_t(np);
        }
        bool uses_reg = (op == OP_STORE || op == OP_LOAD || info.arity == 2);
        if (uses_reg && int(reg) + 1 > out.n_regs) out.n_regs = int(reg) + 1;
        out.recs.push_back(r);
        ++out.n_instructions;
        if (op == OP_RETURN) { returned = true; break; }
    }
    if (!returned) return "tape does not end with _return";
    if (out.extra.empty()) out.extra.push_back(0.0f);
    {
        std::string err;
        if (!allocate_slots(out, err)) return err;
    }
    fold_moves(out.recs);
    if (!out.recs_do.empty()) fold_moves(out.recs_do);
    fuse_leaves(out.recs, false, out.fused);
    if (!out.recs_do.empty()) fuse_leaves(out.recs_do, true, out.fused_do);
    // Zero (= _return) records of padding: the interpreter fetches records in groups.
    Rec pad;
    std::memset(&pad, 0, sizeof(pad));
    for (int i = 0; i < kTapePadding; ++i) {
        out.recs.push_back(pad);
        out.fused.push_back(pad);
        if (!out.recs_do.empty()) { out.recs_do.push_back(pad); out.fused_do.push_back(pad); }
    }
    return "";
}

}  // namespace sdf
