// Device-side helpers shared by the RIS-VEC kernels (gfx950 / CDNA4, wave64).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "risvec.h"

namespace risvec {

// Kernel arguments arrive through scalar loads from a cold scalar cache (every launch starts cold): left alone, the
// compiler fetches the argument a first branch depends on, waits (~0.35 us), branches, and only then requests the
// pointers -- two or more dependent round trips in front of a kernel's first memory request.  Naming the arguments a
// kernel needs first in one `asm volatile("" :: "s"(..))` makes them one batch, one wait.  Matters for every kernel
// that lives only a few microseconds (k_step_fused_lat at BASELINE configs[1]: 0.74 -> 0.44 us before the first
// request is out; the C4 shard 8.1 -> 7.2 us per step).
#define RISVEC_ARGS_IN_ONE_TRIP(...) asm volatile("" ::__VA_ARGS__)

constexpr int kWave = 64;      // CDNA wavefront width (hard-coded on purpose)
constexpr int kBlock = 256;    // 4 waves per workgroup, one per SIMD

// Launch-time dimensions (scalars, passed by value).
struct Dims {
    int E, V, M, cbit;
    long long env_offset;
};

// ---------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11).  Counter = (global env id, vehicle,
// counter, site); key = 64-bit seed.  Restated bit-exactly in
// oracle/risvec_oracle.py::philox4x32.
// ---------------------------------------------------------------------------
enum Site : uint32_t {
    kSiteArrivals = 0, kSiteTurnA = 1, kSiteTurnB = 2, kSiteSpawn = 3,
    kSiteBuf0 = 4, kSite3gpp = 5, kSitePhase = 6
};

__device__ __forceinline__ uint4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                               uint64_t seed) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0;
        c1 = lo1;
        c2 = hi0 ^ c3 ^ k1;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return make_uint4(c0, c1, c2, c3);
}

// uint32 -> float in [0,1): top 24 bits, exact in fp32.
__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * 0x1p-24f; }

// numpy.random.randint(low, high) replacement: low + floor(x (high-low) / 2^32).
__device__ __forceinline__ int randint_u32(uint32_t x, int low, int high) {
    return low + (int)(((uint64_t)x * (uint64_t)(uint32_t)(high - low)) >> 32);
}

// Poisson(lambda) by inversion against the host-built float32 CDF table: the count
// of table entries <= u.  The loop is wave-uniform (exits when no lane advances).
// The first 8 entries are counted branch-free (a compare + add each: at the shipped rates 1 and 3 a draw above
// 7 has probability 1e-6 / 4e-3); the loop over the rest of the table runs only when some lane needs it.
__device__ __forceinline__ int poisson_from_u(float u, const float* __restrict__ cdf) {
    int n = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) n += u >= cdf[k] ? 1 : 0;
    if (__any(u >= cdf[7])) {
#pragma unroll 1
        for (int k = 8; k < RISVEC_POISSON_TABLE; ++k) {
            const bool ge = u >= cdf[k];
            if (!__any(ge)) break;
            n += ge ? 1 : 0;
        }
    }
    return n;
}

// standard normal pair from two uint32 (Box-Muller, fp32) -- production 3GPP path only;
// parity tests inject the reference's own draws instead.
__device__ __forceinline__ float2 normal2(uint32_t a, uint32_t b) {
    const float u1 = ((float)(a >> 8) + 1.0f) * 0x1p-24f;   // (0,1]
    const float u2 = u01(b);
    const float r = sqrtf(-2.0f * logf(u1));
    float s, c;
    sincospif(2.0f * u2, &s, &c);
    return make_float2(r * c, r * s);
}

// ---------------------------------------------------------------------------
// cross-lane: all-reduce over aligned groups of W lanes (W a power of two <= 64)
// ---------------------------------------------------------------------------
template <int W>
__device__ __forceinline__ float group_sum(float x) {
#pragma unroll
    for (int o = W / 2; o > 0; o >>= 1) x += __shfl_xor(x, o, kWave);
    return x;
}

template <int W>
__device__ __forceinline__ double group_sum(double x) {
#pragma unroll
    for (int o = W / 2; o > 0; o >>= 1) x += __shfl_xor(x, o, kWave);
    return x;
}

__host__ __device__ constexpr int pow2_ceil(int x) {
    int p = 1;
    while (p < x) p <<= 1;
    return p;
}

// complex helpers on float2 = (re, im)
// (explicit fused forms: an `a*b - c*d` expression leaves the choice of which product is fused to the
// compiler, and it chose differently in different kernels -- results must not depend on the kernel taken)
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(fmaf(a.x, b.x, -(a.y * b.y)), fmaf(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ float2 cfma(float2 a, float2 b, float2 acc) {
    acc.x = fmaf(a.x, b.x, acc.x);
    acc.x = fmaf(-a.y, b.y, acc.x);
    acc.y = fmaf(a.x, b.y, acc.y);
    acc.y = fmaf(a.y, b.x, acc.y);
    return acc;
}

}  // namespace risvec
