// codecad_amd/csrc/exchange.hip
//
// hu_slice_rows: the device side of the multi-GPU level exchange (the reference has one
// device, reference cl_util/opencl_manager.py:89-98, and no counterpart).  After the all-gather every rank holds
// `world` fixed-size pieces [header row | rows...] (header word 0 = that rank's row count).  The kernel takes the
// rank's balanced share [begin, end) of the concatenation -- the rule of codecad_amd.dist.balanced_slice: sizes
// differ by at most one -- and writes it as [header | rows] again, so the next level's launch reads its parent
// count from the device and the host never waits between levels.
//
// Its own translation unit ON PURPOSE: hip_util.hip is compiled with -mllvm -structurizecfg-skip-uniform-regions
// (the interpreter's dispatch loop needs it, builder.py), and that option miscompiled the first version of this
// kernel: a divergent search loop with a second, uniform exit had its exit-dependent value chosen by a SCALAR
// branch ("some lane left through exit A" => all lanes take A's value) -- out-of-bounds reads on two ranks.
// Code outside the interpreter is built without the option (builder.py: per-source flags); the search below is
// written branch-free as well.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "../../include/hip_util.h"

int hu_fail_external(int code, const char* message);   // hip_util.hip: sets the thread's last error

namespace {

constexpr uint32_t kMaxWorld = 64;

// stats[0] = rows over all ranks, stats[1] != 0: a piece or the share was truncated
__global__ void __launch_bounds__(256)
k_slice_rows(const uint4* __restrict__ gathered, uint32_t world, uint32_t piece_rows, uint32_t row_u4, uint32_t rank,
             uint32_t sharers, uint4* __restrict__ out, uint32_t out_capacity, uint32_t* __restrict__ stats)
{
    // `world` pieces are concatenated; the concatenation is shared out among `sharers` ranks (hu_slice_rows: the same
    // number; hu_slice_rows_of: ONE piece that every rank computed for itself, shared out among all ranks)
    __shared__ uint32_t first[kMaxWorld + 1];   // exclusive prefix of the (clamped) counts
    __shared__ uint32_t truncated;
    if (threadIdx.x == 0) {
        uint32_t total = 0, over = 0;
        for (uint32_t r = 0; r < world; ++r) {
            uint32_t c = gathered[(size_t)r * piece_rows * row_u4].x;
            if (c > piece_rows - 1u) { c = piece_rows - 1u; over = 1u; }
            first[r] = total;
            total += c;
        }
        first[world] = total;
        truncated = over;
    }
    __syncthreads();
    const uint32_t total = first[world];
    const uint32_t base = total / sharers, extra = total - base * sharers;
    const uint32_t begin = rank * base + (rank < extra ? rank : extra);
    uint32_t count = base + (rank < extra ? 1u : 0u);
    const bool over = truncated != 0u || count > out_capacity;
    if (count > out_capacity) count = out_capacity;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        out[0] = make_uint4(count, 0u, 0u, 0u);
        for (uint32_t k = 1; k < row_u4; ++k) out[k] = make_uint4(0u, 0u, 0u, 0u);
        stats[0] = total;
        stats[1] = over ? 1u : 0u;
    }
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint32_t g = begin + i;
    // the piece that holds row g of the concatenation: the number of pieces that end at or before g
    // (a uniform loop with a per-lane sum: no divergent exit)
    uint32_t r = 0;
    for (uint32_t q = 1; q < world; ++q) r += (g >= first[q]) ? 1u : 0u;
    const uint4* src = gathered + ((size_t)r * piece_rows + 1u + (g - first[r])) * row_u4;
    uint4* dst = out + (size_t)(1u + i) * row_u4;
    for (uint32_t k = 0; k < row_u4; ++k) dst[k] = src[k];
}

}  // namespace

static int slice_rows(const void* gathered_dev, uint32_t pieces, uint32_t piece_rows, uint32_t row_bytes, uint32_t rank, uint32_t sharers,
                      void* out_dev, uint32_t out_capacity, uint32_t* stats_dev, void* stream)
{
    if (!gathered_dev || !out_dev || !stats_dev) return hu_fail_external(HU_ERR_BAD_ARG, "NULL argument");
    if (pieces == 0 || pieces > kMaxWorld || sharers == 0 || sharers > kMaxWorld || rank >= sharers)
        return hu_fail_external(HU_ERR_BAD_ARG, "world must be in 1..64 and rank below it");
    if (row_bytes == 0 || row_bytes % 16 != 0) return hu_fail_external(HU_ERR_BAD_ARG, "row_bytes must be a multiple of 16");
    if (piece_rows == 0) return hu_fail_external(HU_ERR_BAD_ARG, "a piece has at least its header row");
    const uint32_t blocks = out_capacity / 256u + 1u;
    hipLaunchKernelGGL(k_slice_rows, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)gathered_dev, pieces,
                       piece_rows, row_bytes / 16u, rank, sharers, (uint4*)out_dev, out_capacity, stats_dev);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hu_fail_external(HU_ERR_HIP, (std::string("k_slice_rows: ") + hipGetErrorString(e)).c_str());
    return HU_OK;
}

extern "C" int hu_slice_rows(const void* gathered_dev, uint32_t world, uint32_t piece_rows, uint32_t row_bytes, uint32_t rank,
                             void* out_dev, uint32_t out_capacity, uint32_t* stats_dev, void* stream)
{
    return slice_rows(gathered_dev, world, piece_rows, row_bytes, rank, world, out_dev, out_capacity, stats_dev, stream);
}

extern "C" int hu_slice_rows_of(const void* piece_dev, uint32_t piece_rows, uint32_t row_bytes, uint32_t rank, uint32_t world,
                                void* out_dev, uint32_t out_capacity, uint32_t* stats_dev, void* stream)
{
    return slice_rows(piece_dev, 1u, piece_rows, row_bytes, rank, world, out_dev, out_capacity, stats_dev, stream);
}
