// codecad_amd/csrc/sort.hip
//
// hu_sort_blocks: order a list of integer block corners (int32[4] rows) by (x, y, z) on the device.
// The compaction of a subdivision level leaves its survivors in the order workgroups raced for list
// space; consumers with per-block output (meshes, contours) want a reproducible order, and reading
// the list back to sort it on the host cost more than the whole traversal (sponge(4) at 1/512: 1.8 of
// 2.2 ms).  Keys are packed into 64 bits and sorted with rocPRIM's radix sort (a plain library sort:
// nothing here is specific to the path); this file is its own translation unit so that the rocPRIM
// templates are compiled once, away from the kernels.
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <cstdint>
#include <string>

#include "../../include/hip_util.h"

int hu_fail_external(int code, const char* message);   // hip_util.hip: sets the thread's last error

namespace {

constexpr int32_t kBias = 1 << 20;   // corners are in resolution units, |c| < 2^20 by the range check

__global__ void __launch_bounds__(256) k_block_keys(const int4* __restrict__ rows, uint32_t n, unsigned long long* keys,
                                                    uint32_t* index, uint32_t* out_of_range)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int4 r = rows[i];
    const uint32_t x = (uint32_t)(r.x + kBias), y = (uint32_t)(r.y + kBias), z = (uint32_t)(r.z + kBias);
    if ((x | y | z) >> 21) atomicOr(out_of_range, 1u);
    keys[i] = ((unsigned long long)(x & 0x1fffffu) << 42) | ((unsigned long long)(y & 0x1fffffu) << 21) | (z & 0x1fffffu);
    index[i] = i;
}

__global__ void __launch_bounds__(256) k_block_gather(const int4* __restrict__ rows, const uint32_t* __restrict__ index,
                                                      uint32_t n, int4* __restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = rows[index[i]];
}

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

extern "C" int hu_sort_blocks(int32_t* blocks_dev, uint32_t n, void* scratch_dev, size_t scratch_bytes, size_t* needed,
                              void* stream)
{
    if (!needed) return hu_fail_external(HU_ERR_BAD_ARG, "needed is NULL");
    hipStream_t s = (hipStream_t)stream;
    size_t sort_bytes = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, sort_bytes, (unsigned long long*)nullptr, (unsigned long long*)nullptr,
                                             (uint32_t*)nullptr, (uint32_t*)nullptr, n ? n : 1u, 0, 63, s);
    if (e != hipSuccess) return hu_fail_external(HU_ERR_HIP, hipGetErrorString(e));
    // layout: keys in | keys out | index in | index out | sorted rows | flag | rocPRIM's scratch
    const size_t k = align256((size_t)n * 8), v = align256((size_t)n * 4), r = align256((size_t)n * 16);
    *needed = 2 * k + 2 * v + r + 256 + align256(sort_bytes);
    if (!scratch_dev || scratch_bytes < *needed || n < 2) return HU_OK;   // size query (or nothing to do)
    if (!blocks_dev) return hu_fail_external(HU_ERR_BAD_ARG, "blocks_dev is NULL");
    char* p = static_cast<char*>(scratch_dev);
    unsigned long long* keys_in = (unsigned long long*)p;
    unsigned long long* keys_out = (unsigned long long*)(p + k);
    uint32_t* idx_in = (uint32_t*)(p + 2 * k);
    uint32_t* idx_out = (uint32_t*)(p + 2 * k + v);
    int4* rows_out = (int4*)(p + 2 * k + 2 * v);
    uint32_t* flag = (uint32_t*)(p + 2 * k + 2 * v + r);
    void* temp = p + 2 * k + 2 * v + r + 256;
    const dim3 grid((n + 255) / 256), block(256);
    e = hipMemsetAsync(flag, 0, 4, s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_block_keys, grid, block, 0, s, (const int4*)blocks_dev, n, keys_in, idx_in, flag);
        e = rocprim::radix_sort_pairs(temp, sort_bytes, keys_in, keys_out, idx_in, idx_out, n, 0, 63, s);
    }
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_block_gather, grid, block, 0, s, (const int4*)blocks_dev, idx_out, n, rows_out);
        e = hipMemcpyAsync(blocks_dev, rows_out, (size_t)n * 16, hipMemcpyDeviceToDevice, s);
    }
    uint32_t bad = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&bad, flag, 4, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return hu_fail_external(HU_ERR_HIP, hipGetErrorString(e));
    if (bad) return hu_fail_external(HU_ERR_BAD_ARG, "block corners outside +-2^20 resolution units cannot be packed into sort keys");
    return HU_OK;
}
