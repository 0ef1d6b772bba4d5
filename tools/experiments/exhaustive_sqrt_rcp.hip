// Exhaustive search over ALL 2^32 binary32 inputs: which short v_rsq/v_rcp + fma sequences give
// exactly the IEEE-754 correctly rounded sqrt(x) and 1/sqrt_rounded(x)?
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt exhaustive_sqrt_rcp.hip -o /tmp/exh
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>

constexpr int kCand = 8;
struct Stats { unsigned long long bad_s[kCand], bad_r[kCand], bad_s_safe[kCand], bad_r_safe[kCand]; uint32_t ex_s[kCand][8], ex_r[kCand][8]; uint32_t n_ex_s[kCand], n_ex_r[kCand]; };

__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

template <int C> __device__ __forceinline__ void cand(float x, float& s, float& r)
{
    if (C == 0) {  // rsq seed, one correction each
        float y = __builtin_amdgcn_rsqf(x);
        float s0 = x * y, h = 0.5f * y;
        float e = fma_(-s0, s0, x);
        s = fma_(e, h, s0);
        float e2 = fma_(-s, y, 1.0f);
        r = fma_(e2, y, y);
    } else if (C == 1) {  // rsq seed, one sqrt correction, two rcp corrections
        float y = __builtin_amdgcn_rsqf(x);
        float s0 = x * y, h = 0.5f * y;
        float e = fma_(-s0, s0, x);
        s = fma_(e, h, s0);
        float e2 = fma_(-s, y, 1.0f);
        float r1 = fma_(e2, y, y);
        float e3 = fma_(-s, r1, 1.0f);
        r = fma_(e3, r1, r1);
    } else if (C == 2) {  // two sqrt corrections, two rcp corrections
        float y = __builtin_amdgcn_rsqf(x);
        float s0 = x * y, h = 0.5f * y;
        float e = fma_(-s0, s0, x);
        float s1 = fma_(e, h, s0);
        float e1 = fma_(-s1, s1, x);
        s = fma_(e1, h, s1);
        float e2 = fma_(-s, y, 1.0f);
        float r1 = fma_(e2, y, y);
        float e3 = fma_(-s, r1, 1.0f);
        r = fma_(e3, r1, r1);
    } else if (C == 3) {  // hardware sqrt seed + rsq for h and r
        float y = __builtin_amdgcn_rsqf(x);
        float s0 = __builtin_amdgcn_sqrtf(x), h = 0.5f * y;
        float e = fma_(-s0, s0, x);
        s = fma_(e, h, s0);
        float e2 = fma_(-s, y, 1.0f);
        r = fma_(e2, y, y);
    } else if (C == 4) {  // Goldschmidt-style coupled: refine h too
        float y = __builtin_amdgcn_rsqf(x);
        float g = x * y, h = 0.5f * y;
        float rr = fma_(-g, h, 0.5f);
        g = fma_(g, rr, g);
        h = fma_(h, rr, h);
        float e = fma_(-g, g, x);
        s = fma_(e, h, g);
        float y2 = h + h;
        float e2 = fma_(-s, y2, 1.0f);
        r = fma_(e2, y2, y2);
    } else if (C == 5) {  // like 0 but rcp seed from v_rcp(s)
        float y = __builtin_amdgcn_rsqf(x);
        float s0 = x * y, h = 0.5f * y;
        float e = fma_(-s0, s0, x);
        s = fma_(e, h, s0);
        float r0 = __builtin_amdgcn_rcpf(s);
        float e2 = fma_(-s, r0, 1.0f);
        r = fma_(e2, r0, r0);
    } else if (C == 6) {  // rcp only: plain 1/x of the INPUT via v_rcp + one correction (r), s unused
        s = __builtin_sqrtf(x);
        float r0 = __builtin_amdgcn_rcpf(x);
        float e2 = fma_(-x, r0, 1.0f);
        r = fma_(e2, r0, r0);
    } else {  // rcp only, two corrections
        s = __builtin_sqrtf(x);
        float r0 = __builtin_amdgcn_rcpf(x);
        float e2 = fma_(-x, r0, 1.0f);
        float r1 = fma_(e2, r0, r0);
        float e3 = fma_(-x, r1, 1.0f);
        r = fma_(e3, r1, r1);
    }
}

__device__ __forceinline__ bool same(float a, float b)
{
    return __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b);
}

template <int C> __device__ __forceinline__ void check(float x, bool safe, Stats* st, unsigned long long (&ls)[kCand][4])
{
    float s, r;
    cand<C>(x, s, r);
    const float s_ref = __builtin_sqrtf(x);
    const float r_ref = (C >= 6) ? 1.0f / x : 1.0f / s_ref;
    if (!same(s, s_ref)) {
        ls[C][0]++;
        if (safe) {
            ls[C][2]++;
            uint32_t k = atomicAdd(&st->n_ex_s[C], 1u);
            if (k < 8) st->ex_s[C][k] = __float_as_uint(x);
        }
    }
    if (!same(r, r_ref)) {
        ls[C][1]++;
        if (safe) {
            ls[C][3]++;
            uint32_t k = atomicAdd(&st->n_ex_r[C], 1u);
            if (k < 8) st->ex_r[C][k] = __float_as_uint(x);
        }
    }
}

__global__ void __launch_bounds__(256) k_exhaustive(Stats* st, float lo, float hi)
{
    unsigned long long ls[kCand][4] = {};
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
        const float x = __uint_as_float((uint32_t)i);
        const bool safe = x >= lo && x <= hi;
        check<0>(x, safe, st, ls); check<1>(x, safe, st, ls); check<2>(x, safe, st, ls); check<3>(x, safe, st, ls);
        check<4>(x, safe, st, ls); check<5>(x, safe, st, ls); check<6>(x, safe, st, ls); check<7>(x, safe, st, ls);
    }
    for (int c = 0; c < kCand; ++c) {
        if (ls[c][0]) atomicAdd(&st->bad_s[c], ls[c][0]);
        if (ls[c][1]) atomicAdd(&st->bad_r[c], ls[c][1]);
        if (ls[c][2]) atomicAdd(&st->bad_s_safe[c], ls[c][2]);
        if (ls[c][3]) atomicAdd(&st->bad_r_safe[c], ls[c][3]);
    }
}

int main()
{
    Stats* d;
    hipMalloc(&d, sizeof(Stats));
    for (int pass = 0; pass < 2; ++pass) {
        const float lo = pass == 0 ? 0x1p-80f : 0x1p-100f, hi = pass == 0 ? 0x1p80f : 0x1p100f;
        hipMemset(d, 0, sizeof(Stats));
        hipLaunchKernelGGL(k_exhaustive, dim3(256 * 16), dim3(256), 0, 0, d, lo, hi);
        Stats h;
        hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("safe range [%g, %g]\n", lo, hi);
        for (int c = 0; c < kCand; ++c) {
            printf("cand %d: sqrt bad %llu (safe %llu)  rcp bad %llu (safe %llu)\n", c, h.bad_s[c], h.bad_s_safe[c], h.bad_r[c], h.bad_r_safe[c]);
            for (uint32_t k = 0; k < h.n_ex_s[c] && k < 8; ++k) { float f; memcpy(&f, &h.ex_s[c][k], 4); printf("   s ex 0x%08x %a\n", h.ex_s[c][k], f); }
            for (uint32_t k = 0; k < h.n_ex_r[c] && k < 8; ++k) { float f; memcpy(&f, &h.ex_r[c][k], 4); printf("   r ex 0x%08x %a\n", h.ex_r[c][k], f); }
        }
    }
    return 0;
}
