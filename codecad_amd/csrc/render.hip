// codecad_amd/csrc/render.hip
//
// The kernels that are NOT built with -mllvm -structurizecfg-skip-uniform-regions (launchers.hpp tells which and
// why): the ray caster and the bitmap renderer over the tape interpreter (reference rendering/ray_caster.cl:146-256,
// rendering/bitmap.cl:1-18), 2D contouring (rendering/polygon2d.cl:82-175), the reduction of the per-parent moment
// sums to a level's ten integrals (mass_properties.py:119-148) and the self-test of the arithmetic contract.
// hip_util.hip validates arguments and calls the launch functions at the end of this file.
#include "launchers.hpp"

using namespace sdfk;

namespace {

// ------------------------------------------------------------------------------------------
// mass_properties: per-parent index sums -> the ten integrals of this level, on the device.
// The reference does this on the host, block by block, in Python doubles with Kahan sums
// (mass_properties.py:119-148).  Same per-block formulas in fp64 (no contraction), summed
// deterministically: workgroup g takes the g-th contiguous slice of the parents, thread t of it
// Kahan-accumulates the slice's parents t, t+1024, ... and the 1024 partial sums are combined by a
// fixed tree into row g of the output; the caller adds the rows in order.  Nothing depends on launch
// timing.  (One workgroup for a whole level -- the first version -- took 172 us for the 167 k leaf
// parents of sponge(4) at 1/512, 15 % of the whole mass_properties call.)
// out[g][10] order: 1, x, y, z, xx, yy, zz, xy, xz, yz.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024)
k_mass_integrals(const double4* __restrict__ parents, const uint32_t* __restrict__ sums, uint32_t n, uint32_t per_row,
                 const uint32_t* __restrict__ n_dev, double s, double* __restrict__ out)
{
    // indirect form (hu_mass_integrals_indirect): the number of parents lives on the device (at most `n`, the list's
    // capacity, are there); the slices are cut from it, so the rows mean the same as in the direct form
    if (n_dev) {
        const uint32_t have = *n_dev;
        n = have < n ? have : n;
        per_row = (n + gridDim.x - 1) / gridDim.x;
    }
    const uint64_t first = (uint64_t)blockIdx.x * per_row;              // rows past the last parent get an empty slice
    const uint32_t begin = first < n ? (uint32_t)first : n;
    const uint32_t end = (n - begin < per_row) ? n : begin + per_row;
    __shared__ double part[10][1024];
    double acc[10], comp[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) acc[k] = comp[k] = 0.0;
    const double s2 = s * s, s3 = s * s2, h = s / 2, twelfth = s2 / 12;
    for (uint32_t p = begin + threadIdx.x; p < end; p += 1024) {
        const double4 c = parents[p];
        const uint32_t* u = sums + (size_t)p * 10;
        const double sxx = u[0], sxy = u[1], sxz = u[2], sx = u[3], syy = u[4], syz = u[5], sy = u[6], szz = u[7],
                     sz = u[8], cnt = u[9];
        const double bx = c.x + h, by = c.y + h, bz = c.z + h;
        const double tx = s * sx, ty = s * sy, tz = s * sz;
        const double v[10] = {
            s3 * cnt,
            s3 * (cnt * bx + tx), s3 * (cnt * by + ty), s3 * (cnt * bz + tz),
            s3 * (cnt * (bx * bx + twelfth) + 2 * bx * tx + s2 * sxx),
            s3 * (cnt * (by * by + twelfth) + 2 * by * ty + s2 * syy),
            s3 * (cnt * (bz * bz + twelfth) + 2 * bz * tz + s2 * szz),
            s3 * (cnt * bx * by + bx * ty + by * tx + s2 * sxy),
            s3 * (cnt * bx * bz + bx * tz + bz * tx + s2 * sxz),
            s3 * (cnt * by * bz + by * tz + bz * ty + s2 * syz)};
#pragma unroll
        for (int k = 0; k < 10; ++k) {  // Kahan, like the reference's util.KahanSummation
            const double y = v[k] - comp[k];
            const double t = acc[k] + y;
            comp[k] = (t - acc[k]) - y;
            acc[k] = t;
        }
    }
#pragma unroll
    for (int k = 0; k < 10; ++k) part[k][threadIdx.x] = acc[k];
    __syncthreads();
    for (uint32_t stride = 512; stride > 0; stride >>= 1) {
        if (threadIdx.x < stride) {
#pragma unroll
            for (int k = 0; k < 10; ++k) part[k][threadIdx.x] += part[k][threadIdx.x + stride];
        }
        __syncthreads();
    }
    if (threadIdx.x < 10) out[(size_t)blockIdx.x * 10 + threadIdx.x] = part[threadIdx.x][0];
}

// all 2^32 inputs through sqrt_cr / sqrt_inv_cr against the compiler's IEEE expansions (kernels.hpp same_bits)
__global__ void __launch_bounds__(256) k_selftest_math(unsigned long long* counts)
{
    unsigned long long bad[3] = {0, 0, 0}, fast = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
        const float x0 = __uint_as_float((uint32_t)i), x1 = __uint_as_float(~(uint32_t)i);
        const float s_ref0 = sdf::sqrt_(x0), s_ref1 = sdf::sqrt_(x1);
        const float r_ref0 = 1.0f / s_ref0, r_ref1 = 1.0f / s_ref1;
        // one voxel per lane
        float s, r;
        sdf::sqrt_inv_cr(x0, sdf::mask_of<float>::all(), s, r);
        bad[0] += !same_bits(sdf::sqrt_cr(x0, sdf::mask_of<float>::all()), s_ref0);
        bad[1] += !same_bits(s, s_ref0);
        bad[2] += !same_bits(r, r_ref0);
        // two voxels per lane
        const sdf::f2 x = sdf::make_f2(x0, x1);
        sdf::f2 s2, r2;
        sdf::sqrt_inv_cr(x, sdf::mask_of<sdf::f2>::all(), s2, r2);
        const sdf::f2 q2 = sdf::sqrt_cr(x, sdf::mask_of<sdf::f2>::all());
        bad[0] += !same_bits(q2.x, s_ref0) + !same_bits(q2.y, s_ref1);
        bad[1] += !same_bits(s2.x, s_ref0) + !same_bits(s2.y, s_ref1);
        bad[2] += !same_bits(r2.x, r_ref0) + !same_bits(r2.y, r_ref1);
        fast += !sdf::outside_fast_range(x0).v;
    }
    for (int k = 0; k < 3; ++k)
        if (bad[k]) atomicAdd(&counts[k], bad[k]);
    atomicAdd(&counts[3], fast);
}

// min3_ / max3_ (interp.hpp) against the two instructions they replace: every ordered triple of 64 special values, then
// 2^26 triples of random bit patterns; one and two voxels per lane.  counts[0], counts[1] <- disagreements, counts[2] <- triples
__device__ __forceinline__ float minmax3_special(uint32_t k)
{
    const uint32_t mags[16] = {0x00000000u, 0x00000001u, 0x007fffffu, 0x00800000u, 0x00800001u, 0x3f7fffffu, 0x3f800000u, 0x3f800001u,
                               0x7f7fffffu, 0x7f800000u, 0x7f800001u, 0x7fbfffffu, 0x7fc00000u, 0x7fc00001u, 0x7fffffffu, 0x40490fdbu};
    // 64 values: the sixteen magnitudes with both signs, and the same again with another mantissa bit flipped
    return __uint_as_float((mags[k & 15u] | ((k & 16u) << 27)) ^ ((k & 32u) ? 0x00000400u : 0u));
}
__global__ void __launch_bounds__(256) k_selftest_minmax3(unsigned long long* counts)
{
    unsigned long long bad_min = 0, bad_max = 0, n = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x, first = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    auto check = [&](float a, float b, float c) {
        bad_min += !same_bits(sdf::min3_(a, b, c), sdf::min_(sdf::min_(a, b), c));
        bad_max += !same_bits(sdf::max3_(a, b, c), sdf::max_(sdf::max_(a, b), c));
        const sdf::f2 a2 = sdf::make_f2(a, c), b2 = sdf::make_f2(b, a), c2 = sdf::make_f2(c, b);
        const sdf::f2 m = sdf::min3_(a2, b2, c2), mr = sdf::min_(sdf::min_(a2, b2), c2), M = sdf::max3_(a2, b2, c2), Mr = sdf::max_(sdf::max_(a2, b2), c2);
        bad_min += !same_bits(m.x, mr.x) + !same_bits(m.y, mr.y);
        bad_max += !same_bits(M.x, Mr.x) + !same_bits(M.y, Mr.y);
        ++n;
    };
    for (uint64_t i = first; i < 64ull * 64ull * 64ull; i += stride)
        check(minmax3_special((uint32_t)i & 63u), minmax3_special((uint32_t)(i >> 6) & 63u), minmax3_special((uint32_t)(i >> 12) & 63u));
    for (uint64_t i = first; i < (1ull << 26); i += stride) {
        uint64_t h = i * 0x9e3779b97f4a7c15ull + 0x1234567ull;       // splitmix64
        uint32_t w[3];
        for (int k = 0; k < 3; ++k) {
            h += 0x9e3779b97f4a7c15ull;
            uint64_t z = h;
            z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
            z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
            w[k] = (uint32_t)((z ^ (z >> 31)) >> 16);
        }
        check(__uint_as_float(w[0]), __uint_as_float(w[1]), __uint_as_float(w[2]));
    }
    if (bad_min) atomicAdd(&counts[0], bad_min);
    if (bad_max) atomicAdd(&counts[1], bad_max);
    atomicAdd(&counts[2], n);
}

}  // namespace

namespace hu_render {

hipError_t allow_big_lds(size_t bytes)
{
    hipError_t e = hipFuncSetAttribute((const void*)k_ray_caster<InterpEval<false>>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_bitmap<InterpEval<false>>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_bitmap<InterpEval<true>>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    return e;
}

hipError_t ray_caster(const InterpEval<false>& ev, const RayCasterArgs& a, uint32_t blocks, uint32_t block, size_t lds, hipStream_t stream)
{
    hipLaunchKernelGGL((k_ray_caster<InterpEval<false>>), dim3(blocks), dim3(block), lds, stream, ev, a);
    return hipGetLastError();
}

hipError_t bitmap(bool distance_only, const sdf::Rec* prog, const float* extra, uint32_t n4, float ox, float oy, float oz,
                  float step_size, uint32_t width, uint32_t height, uint8_t* out, uint32_t blocks, uint32_t block, size_t lds,
                  hipStream_t stream)
{
    if (distance_only)
        hipLaunchKernelGGL((k_bitmap<InterpEval<true>>), dim3(blocks), dim3(block), lds, stream, (InterpEval<true>{prog, extra, n4}), ox, oy,
                           oz, step_size, width, height, out);
    else
        hipLaunchKernelGGL((k_bitmap<InterpEval<false>>), dim3(blocks), dim3(block), lds, stream, (InterpEval<false>{prog, extra, n4}), ox,
                           oy, oz, step_size, width, height, out);
    return hipGetLastError();
}

hipError_t process_polygon(bool batch, const PolygonArgs& a, dim3 grid, hipStream_t stream)
{
    if (batch) hipLaunchKernelGGL(k_process_polygon<true>, grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(k_process_polygon<false>, grid, dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t mass_integrals(const double4* parents, const uint32_t* sums, uint32_t n_parents, uint32_t per_row, const uint32_t* n_parents_dev,
                          double s, double* out, uint32_t rows, hipStream_t stream)
{
    hipLaunchKernelGGL(k_mass_integrals, dim3(rows), dim3(1024), 0, stream, parents, sums, n_parents, per_row, n_parents_dev, s, out);
    return hipGetLastError();
}

hipError_t selftest_math(unsigned long long* counts_dev)
{
    hipLaunchKernelGGL(k_selftest_math, dim3(256 * 32), dim3(256), 0, nullptr, counts_dev);
    return hipGetLastError();
}

hipError_t selftest_minmax3(unsigned long long* counts_dev)
{
    hipLaunchKernelGGL(k_selftest_minmax3, dim3(256 * 8), dim3(256), 0, nullptr, counts_dev);
    return hipGetLastError();
}

}  // namespace hu_render
