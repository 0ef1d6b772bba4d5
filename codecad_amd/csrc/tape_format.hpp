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
    OPX_TO_ROW_X = 38, OPX_TO_ROWS_YZ = 39, OPX_FROM_MATRIX = 40, OPX_INIT_ROW_X = 41, OPX_INIT_ROWS_YZ = 42
};


constexpr int kRefRegisterCount = 512;  // reference nodes/__init__.py:6
constexpr int kVariableParams = -1;
constexpr int kTapePadding = 8;         // >= interp.hpp kFetchGroup



// One decoded instruction: 12 dwords.  hdr = opcode | (slot << 8) | kResultKind?.
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
#define SDF_REC_DWORDS 12
#endif
struct alignas(SDF_REC_DWORDS == 16 ? 64 : 16) Rec {
    uint32_t hdr;
    float p[SDF_REC_DWORDS - 1];
};
static_assert(sizeof(Rec) == 4 * SDF_REC_DWORDS, "unexpected Rec size");


}  // namespace sdf
