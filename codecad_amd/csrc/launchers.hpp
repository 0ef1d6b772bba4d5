// codecad_amd/csrc/launchers.hpp -- what hip_util.hip calls in render.hip.
//
// Two translation units hold kernels (hip_util/builder.py lists the flags of each):
//   hip_util.hip  the dense / leaf-block / classification kernels over the tape interpreter, built with
//                 -mllvm -structurizecfg-skip-uniform-regions (the interpreter's dispatch loop needs it);
//   render.hip    every other kernel -- ray caster, bitmap, 2D contouring, the mass-integral reduction, the
//                 arithmetic self-test -- built WITHOUT it: that option once let a scalar branch choose a per-lane
//                 value in a divergent loop (csrc/exchange.hip), so it stays confined to the kernels that are
//                 nothing but the interpreter's wave-uniform loop around branch-free ops.
// Each function enqueues one launch and returns hipGetLastError().
#pragma once

#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace hu_render {

hipError_t allow_big_lds(size_t bytes);   // dynamic LDS above 64 KiB for the interpreter instantiations of this unit
hipError_t ray_caster(const sdfk::InterpEval<false>& ev, const sdfk::RayCasterArgs& a, uint32_t blocks, uint32_t block, size_t lds,
                      hipStream_t stream);
hipError_t bitmap(bool distance_only, const sdf::Rec* prog, const float* extra, uint32_t n4, float ox, float oy, float oz,
                  float step_size, uint32_t width, uint32_t height, uint8_t* out, uint32_t blocks, uint32_t block, size_t lds,
                  hipStream_t stream);
hipError_t process_polygon(bool batch, const sdfk::PolygonArgs& a, dim3 grid, hipStream_t stream);
hipError_t mass_integrals(const double4* parents, const uint32_t* sums, uint32_t n_parents, uint32_t per_row, const uint32_t* n_parents_dev,
                          double s, double* out, uint32_t rows, hipStream_t stream);
hipError_t selftest_math(unsigned long long* counts_dev);
hipError_t selftest_minmax3(unsigned long long* counts_dev);

}  // namespace hu_render
