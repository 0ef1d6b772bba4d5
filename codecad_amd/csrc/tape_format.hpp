// codecad_amd/csrc/tape_format.hpp
//
// Device-visible part of the decoded tape format: opcodes and the fixed-size record.
// Free of host-only includes so that hipRTC can compile it (specialised kernels, jit.hpp).
#pragma once

#include "sdf_math.hpp"

namespace sdf {

enum Op : uint32_t {
    OP_RETURN = 0, OP_STORE = 1, OP_LOAD = 2, OP_RECTANGLE = 3, OP_CIRCLE = 4,
    OP_REGULAR_POLYGON2D = 5, OP_POLYGON2D = 6, OP_SPHERE = 7, OP_HALF_SPACE = 8,
    OP_REVOLUTION_TO = 9, OP_TWIST_REVOLUTION_TO = 10, OP_INITIAL_TRANSFORMATION_TO = 11,
    OP_TRANSFORMATION_TO = 12, OP_TRANSFORMATION_FROM = 13, OP_MIRROR = 14,
    OP_SYMMETRICAL_TO = 15, OP_OFFSET = 16, OP_SHELL = 17, OP_REPETITION = 18,
    OP_CIRCULAR_REPETITION_TO = 19, OP_CIRCULAR_REPETITION_FROM = 20, OP_INVOLUTE_GEAR = 21,
    OP_EXTRUSION = 22, OP_REVOLUTION_FROM = 23, OP_TWIST_REVOLUTION_FROM = 24,
    OP_SYMMETRICAL_FROM = 25, OP_UNION = 26, OP_INTERSECTION = 27, OP_SUBTRACTION = 28,
    OP_COUNT = 29,
    // Internal opcodes: never in a tape, produced by the decoder for special cases that are
    // provably equal (under ==) to the general op.  transformation_from rotates a DIRECTION by
    // the quaternion; when its vector part is zero (pure scale) or has a single non-zero
    // component (rotation about a coordinate axis) most products are exact zeros.
    OPX_FROM_SCALE = 29, OPX_FROM_AXIS_X = 30, OPX_FROM_AXIS_Y = 31, OPX_FROM_AXIS_Z = 32,
    // The same for transformation_to, which rotates the sample POINT: with a zero vector part the
    // transform is a scaling, with one non-zero component a rotation about a coordinate axis, and
    // the dropped terms are exact zeros for every finite point (DESIGN.md "Canonical arithmetic":
    // this reduction is part of the arithmetic contract; the oracle applies it too).  OPX_POINT
    // loads the sample point, so that initial_transformation_to can use the same reduced forms.
    OPX_POINT = 33, OPX_TO_SCALE = 34, OPX_TO_AXIS_X = 35, OPX_TO_AXIS_Y = 36, OPX_TO_AXIS_Z = 37,
    // A general quaternion as the 3x3 rotation-and-scale matrix it is (tape.hpp matrix_constants).  A record has
    // ten free parameters and transformation_to needs twelve (nine entries, three offsets), so it becomes two
    // records: the first parks the new x in the w component of `last` (a point has no use for w) and leaves
    // x, y, z alone, the second computes y and z from them and assembles (x', y', z', 0).  transformation_from
    // (nine entries over |Q|^2 and the distance scale) fits one record.
    // The OPX_INIT_* pair is the same for initial_transformation_to: it reads the sample point itself.
    OPX_TO_ROW_X = 38, OPX_TO_ROWS_YZ = 39, OPX_FROM_MATRIX = 40, OPX_INIT_ROW_X = 41, OPX_INIT_ROWS_YZ = 42,
    // A transformed primitive combined into an accumulator -- the unit CAD tapes are made of -- as ONE record of the
    // interpreter's programs (tape.hpp fuse_leaves; per-tape code is generated from the unfused records):
    //   [sample point | last | slot] -> to (scale / axis rotation) -> rectangle | circle | sphere | half_space
    //   -> extrusion -> from (scale / axis rotation) -> up to two of union | intersection | subtraction -> [store]
    // every part optional except the primitive.  The SAME operations in the same order as the single records (the
    // results are identical); what goes is the dispatch between them: sponge(4) 56 -> 26 records.
    OPX_LEAF = 43
};

// OPX_LEAF parameter layout (16-dword records) and its control word p[kLeafControl]
constexpr int kLeafTo = 0;        // p[0..2] = A, B, C of the to-part, p[3..5] = its offsets
constexpr int kLeafPrim = 6;      // p[6], p[7]: the primitive's parameters
constexpr int kLeafExtrude = 8;   // p[8]: half height
constexpr int kLeafScale = 9;     // p[9]: the from-part's distance scale, p[11..13] = its A, B, C   (p[10] is the fold word)
constexpr int kLeafFrom = 11;
constexpr int kLeafControl = 14;
constexpr uint32_t kLeafSample = 1u;                 // the point is the sample point (OPX_POINT fused)
constexpr uint32_t kLeafToShift = 1, kLeafPrimShift = 4, kLeafFromShift = 8;   // to / from: 3 bits, 0 none, 1 scale, 2/3/4 axis x/y/z; primitive: 2 bits
constexpr uint32_t kLeafExtrusion = 1u << 7;
constexpr uint32_t kLeafComb1Shift = 11, kLeafComb2Shift = 21;   // 2 bits kind (1 union, 2 intersection, 3 subtraction) + 8 bits slot
constexpr uint32_t kLeafMidStore = 1u << 31;         // the transformed point is also stored, to the slot in hdr
// OFF: measured on one box, the same day: with the block that applies the late scaling compiled into the leaf the
// interpreter runs sponge(4) in 3.45 ms (905 vector + 750 scalar instructions per wavefront, 22 records), without it
// in 3.31 ms (863 + 654, 26 records) -- the four dispatches saved cost less than what the extra block does to the
// leaf's code.  -DSDF_LEAF_FROM_LAST=1 brings it back.
#ifndef SDF_LEAF_FROM_LAST
#define SDF_LEAF_FROM_LAST 0
#endif
constexpr uint32_t kLeafFromLast = 1u << 6;          // the from-part (a scaling) runs AFTER the selects: to prim select select from
enum LeafPrim : uint32_t { LEAF_RECTANGLE = 0, LEAF_CIRCLE = 1, LEAF_SPHERE = 2, LEAF_HALF_SPACE = 3 };



// The decoder rewrites every transformation_to / transformation_from into a reduced or a matrix form (tape.hpp), so the
// quaternion forms of the tape never reach a kernel: their cases are compiled only when the rewriting is switched off.
#ifndef SDF_TO_SPECIAL
#define SDF_TO_SPECIAL 1
#endif
#ifndef SDF_FROM_SPECIAL
#define SDF_FROM_SPECIAL 1
#endif

constexpr int kRefRegisterCount = 512;  // reference nodes/__init__.py:6
constexpr int kVariableParams = -1;
constexpr int kTapePadding = 8;         // >= interp.hpp kFetchGroup



// One decoded instruction: 16 dwords.  hdr = opcode | (slot << 8) | kResultKind?.
// `slot` is NOT the tape's register number: registers are renamed at decode time
// (allocate_slots below) to the smallest set of LDS slots that liveness allows.
constexpr uint32_t kResultKind = 0x80000000u;  // distance-only program: the slot holds a bare distance
// _load / _store folded into the neighbouring record (tape.hpp fold_moves): the last parameter dword
// says "first load `last` from a slot" and/or "afterwards store `last` to a slot".  A third of the
// sponge's instructions are such moves; folded, they cost two uniform tests instead of a dispatch each.
constexpr int kFoldParam = 10;   // p[10]: no op uses it (records have 11 parameter dwords)
constexpr uint32_t kFoldLoad = 0x100u, kFoldLoadResult = 0x200u;          // bits 0-7: slot
constexpr uint32_t kFoldStore = 0x1000000u, kFoldStoreResult = 0x2000000u;  // bits 16-23: slot
#ifndef SDF_REC_DWORDS
#define SDF_REC_DWORDS 16     // 64-byte records: one s_load_dwordx16 each, and room for a fused leaf (OPX_LEAF)
#endif
struct alignas(SDF_REC_DWORDS == 16 ? 64 : 16) Rec {
    uint32_t hdr;
    float p[SDF_REC_DWORDS - 1];
};
static_assert(sizeof(Rec) == 4 * SDF_REC_DWORDS, "unexpected Rec size");


}  // namespace sdf
