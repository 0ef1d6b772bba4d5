// tools/experiments/store_patterns.hip -- STORES ALONE: a 512^3 float4 grid (2 GiB, z fastest, then y, then x -- the layout
// grid_eval's ABI fixes) written by workgroups of 256 lanes in different orders, nothing else in the kernels.  The question
// (DESIGN.md section 5.1): the dense kernel over 16^3 boxes is held at 0.39 ms by its store stream where runs along z reach
// 0.32 ms -- is it the 128-byte segments, the 256-byte rows of a box, or the footprint of the boxes in flight?
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/experiments/build/store_patterns tools/experiments/store_patterns.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr uint32_t N = 512;

__device__ __forceinline__ float4 value(uint32_t x, uint32_t y, uint32_t z) { return make_float4((float)x, (float)y, (float)z, 1.0f); }

// runs along z: a workgroup writes 512 consecutive voxels (two per lane, 64 lanes apart)
__global__ void __launch_bounds__(256) k_runs(float4* out)
{
    const size_t base = (size_t)blockIdx.x * 512u;
    for (int i = 0; i < 2; ++i) {
        const size_t lin = base + (threadIdx.x & 63u) + 128u * (threadIdx.x >> 6) + 64u * i;
        out[lin] = value((uint32_t)(lin >> 18), (uint32_t)(lin >> 9) & 511u, (uint32_t)lin & 511u);
    }
}

// a box of BX x BY x BZ voxels per workgroup (4096 voxels), a lane walks along x; a wavefront's store instruction covers
// LZ voxels along z, 64 / LZ / LX rows along y and LX planes (LX = 2: the second voxel of a lane two planes on, as box_eval)
template <int BX, int BY, int BZ, int LZ, int ORDER>
__global__ void __launch_bounds__(256) k_boxes(float4* out)
{
    constexpr uint32_t nbx = N / BX, nby = N / BY, nbz = N / BZ;
    uint32_t b = blockIdx.x;
    if (ORDER == 1) {   // contiguous chunks per XCD (workgroup i runs on XCD i % 8)
        const uint32_t per = (nbx * nby * nbz) / 8u;
        b = (b & 7u) * per + (b >> 3);
    }
    uint32_t qz, qy, qx;
    if (ORDER == 2) { qx = b % nbx; qz = (b / nbx) % nbz; qy = b / (nbx * nbz); }   // x fastest
    else { qz = b % nbz; qy = (b / nbz) % nby; qx = b / (nbz * nby); }
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    // the (y, z) face of the box: BY x BZ columns, 64 * 4 lanes x ... each lane takes columns so that a wavefront covers
    // LZ along z and 32 / LZ (x 2 planes) or 64 / LZ rows
    constexpr uint32_t rows_per_wave = 32u / LZ;            // with two planes per instruction
    constexpr uint32_t cols = BY * BZ;                       // columns of the face
    constexpr uint32_t per_wave_cols = rows_per_wave * LZ;   // columns a wavefront covers at once (32)
    const uint32_t lz = lane % LZ, ly = (lane / LZ) % rows_per_wave, lx = lane / 32u;
    for (uint32_t c = wave * per_wave_cols; c < cols; c += 4u * per_wave_cols) {
        // tile of the face: LZ along z, rows_per_wave along y
        const uint32_t tiles_z = BZ / LZ;
        const uint32_t tile = c / per_wave_cols;
        const uint32_t z = (tile % tiles_z) * LZ + lz, y = (tile / tiles_z) * rows_per_wave + ly;
        for (uint32_t x = lx; x < BX; x += 4u) {
            for (uint32_t i = 0; i < 2u; ++i) {
                const uint32_t X = qx * BX + x + 2u * i, Y = qy * BY + y, Z = qz * BZ + z;
                out[((size_t)X * N + Y) * N + Z] = value(X, Y, Z);
            }
        }
    }
}

template <class F> float time_it(F launch, int reps)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) launch();
    std::vector<float> ms;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(a, 0);
        launch();
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float t = 0; hipEventElapsedTime(&t, a, b); ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2];
}

int check(float4* out, std::vector<float4>& host, const char* name)
{
    // every voxel written with its own coordinates? (a sample of planes)
    for (uint32_t x : {0u, 17u, 255u, 511u}) {
        if (hipMemcpy(host.data(), out + (size_t)x * N * N, (size_t)N * N * sizeof(float4), hipMemcpyDeviceToHost) != hipSuccess) return 1;
        for (uint32_t y = 0; y < N; ++y)
            for (uint32_t z = 0; z < N; ++z) {
                const float4 v = host[(size_t)y * N + z];
                if (v.x != (float)x || v.y != (float)y || v.z != (float)z) { std::printf("%s: wrong voxel at %u %u %u\n", name, x, y, z); return 1; }
            }
    }
    return 0;
}

#define RUN_BOX(BX, BY, BZ, LZ, ORDER) RUN_BOX_LDS(BX, BY, BZ, LZ, ORDER, 0)
// LDS: dynamic shared memory the kernel never touches -- it only limits the workgroups (of four wavefronts) per CU: 160 KiB / LDS
#define RUN_BOX_LDS(BX, BY, BZ, LZ, ORDER, LDS)                                                                                 \
    do {                                                                                                                        \
        CHECK(hipMemset(out, 0xff, bytes));                                                                                     \
        const float ms = time_it([&] { hipLaunchKernelGGL((k_boxes<BX, BY, BZ, LZ, ORDER>), dim3(boxes), dim3(256), LDS, 0, out); }, reps); \
        char name[112]; std::snprintf(name, sizeof name, "boxes %2d x %2d x %3d, %2d voxels along z per store, order %d, %2d KiB LDS", BX, BY, BZ, LZ, ORDER, LDS / 1024); \
        if (check(out, host, name)) return 1;                                                                                   \
        std::printf("%-70s %.4f ms  %.2f TB/s\n", name, ms, bytes / ms * 1e-9);                                                \
    } while (0)

int main(int argc, char** argv)
{
    const int reps = argc > 1 ? std::atoi(argv[1]) : 15;
    const size_t bytes = (size_t)N * N * N * sizeof(float4);
    float4* out = nullptr;
    CHECK(hipMalloc((void**)&out, bytes));
    std::vector<float4> host((size_t)N * N);
    const uint32_t boxes = N * N * N / 4096u;
    {
        CHECK(hipMemset(out, 0xff, bytes));
        const float ms = time_it([&] { hipLaunchKernelGGL(k_runs, dim3(N * N * N / 512u), dim3(256), 0, 0, out); }, reps);
        if (check(out, host, "runs")) return 1;
        std::printf("%-70s %.4f ms  %.2f TB/s\n", "runs along z (512 voxels per workgroup)", ms, bytes / ms * 1e-9);
    }
    for (int lds : {16, 27, 32, 40, 53, 64}) {      // 10, 5 (6 would be 26.6), 5, 4, 3, 2 workgroups per CU
        const float ms = time_it([&] { hipLaunchKernelGGL(k_runs, dim3(N * N * N / 512u), dim3(256), lds * 1024, 0, out); }, reps);
        std::printf("runs along z, %2d KiB of idle LDS per workgroup %31s %.4f ms  %.2f TB/s\n", lds, "", ms, bytes / ms * 1e-9);
    }
    RUN_BOX_LDS(16, 16, 16, 8, 0, 27 * 1024);
    RUN_BOX_LDS(16, 16, 16, 8, 0, 32 * 1024);
    RUN_BOX_LDS(16, 16, 16, 8, 0, 40 * 1024);
    RUN_BOX_LDS(16, 16, 16, 8, 0, 53 * 1024);
    RUN_BOX_LDS(16, 16, 16, 8, 0, 64 * 1024);
    RUN_BOX_LDS(16, 16, 16, 8, 1, 40 * 1024);
    RUN_BOX_LDS(16, 16, 16, 16, 0, 40 * 1024);
    RUN_BOX_LDS(16, 4, 64, 32, 0, 40 * 1024);
    RUN_BOX_LDS(8, 8, 64, 32, 0, 40 * 1024);
    RUN_BOX(16, 16, 16, 8, 0);     // box_eval today: 128-byte segments, 16^3 boxes, z fastest box order
    RUN_BOX(16, 16, 16, 16, 0);    // 256-byte segments
    RUN_BOX(16, 16, 16, 8, 1);     // contiguous chunks per XCD
    RUN_BOX(16, 16, 16, 8, 2);     // x fastest box order
    RUN_BOX(16, 8, 32, 8, 0);
    RUN_BOX(16, 8, 32, 32, 0);
    RUN_BOX(16, 4, 64, 16, 0);
    RUN_BOX(16, 4, 64, 32, 0);
    RUN_BOX(16, 2, 128, 32, 0);
    RUN_BOX(8, 8, 64, 32, 0);
    RUN_BOX(8, 4, 128, 32, 0);
    RUN_BOX(4, 16, 64, 32, 0);
    RUN_BOX(16, 4, 64, 32, 1);
    RUN_BOX(16, 4, 64, 32, 2);
    RUN_BOX(32, 2, 64, 32, 0);
    RUN_BOX(64, 1, 64, 32, 0);
    hipFree(out);
    return 0;
}
