// Achievable-bandwidth yardstick for the roofline in DESIGN.md: the simplest possible streaming
// kernels on the same device, same sizes as the headline kernel's working set.  Not part of the
// product library.   read: every lane float4-loads a grid-strided stream and keeps a running sum
// (one float written per lane at the end); mix: reads `n_read` float4s and writes `n_write`.
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void __launch_bounds__(256)
k_read(const float4* __restrict__ src, long long n, float* __restrict__ sink) {
    const long long stride = (long long)gridDim.x * 256;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    float4 a0 = make_float4(0, 0, 0, 0), a1 = a0, a2 = a0, a3 = a0;
    for (; i + 3 * stride < n; i += 4 * stride) {
        const float4 x0 = src[i], x1 = src[i + stride], x2 = src[i + 2 * stride], x3 = src[i + 3 * stride];
        a0.x += x0.x; a0.y += x0.y; a0.z += x0.z; a0.w += x0.w;
        a1.x += x1.x; a1.y += x1.y; a1.z += x1.z; a1.w += x1.w;
        a2.x += x2.x; a2.y += x2.y; a2.z += x2.z; a2.w += x2.w;
        a3.x += x3.x; a3.y += x3.y; a3.z += x3.z; a3.w += x3.w;
    }
    for (; i < n; i += stride) { const float4 x = src[i]; a0.x += x.x; a0.y += x.y; a0.z += x.z; a0.w += x.w; }
    sink[(long long)blockIdx.x * 256 + threadIdx.x] =
        a0.x + a0.y + a0.z + a0.w + a1.x + a1.y + a1.z + a1.w + a2.x + a2.y + a2.z + a2.w + a3.x + a3.y + a3.z + a3.w;
}

__global__ void __launch_bounds__(256)
k_copy(const float4* __restrict__ src, float4* __restrict__ dst, long long n) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

extern "C" int membench_read(const void* src, long long n_float4, void* sink, int blocks, void* stream) {
    hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float4*)src, n_float4, (float*)sink);
    return (int)hipGetLastError();
}
extern "C" int membench_copy(const void* src, void* dst, long long n_float4, int blocks, void* stream) {
    hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float4*)src, (float4*)dst, n_float4);
    return (int)hipGetLastError();
}

// ---- round 2: what is the ceiling for a RE-READ working set of the headline's size? ------------------------
// variants: loads in flight per lane (U), chunked (a workgroup owns a contiguous chunk, as the fused kernel's
// wavefronts own env groups) vs grid-strided, non-temporal loads.
template <int U, bool CHUNK, bool NT>
__global__ void __launch_bounds__(256)
k_read2(const float4* __restrict__ src, long long n, float* __restrict__ sink) {
    float4 acc[U];
#pragma unroll
    for (int k = 0; k < U; ++k) acc[k] = make_float4(0, 0, 0, 0);
    long long i, end, step;
    if (CHUNK) {
        const long long per = (n + gridDim.x - 1) / gridDim.x;
        i = (long long)blockIdx.x * per + threadIdx.x;
        end = (long long)(blockIdx.x + 1) * per;
        if (end > n) end = n;
        step = 256;
    } else {
        i = (long long)blockIdx.x * 256 + threadIdx.x;
        end = n;
        step = (long long)gridDim.x * 256;
    }
    for (; i + (U - 1) * step < end; i += U * step) {
        float4 x[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            if (NT) {
                typedef float v4f __attribute__((ext_vector_type(4)));
                const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(src + i + k * step));
                x[k] = make_float4(t.x, t.y, t.z, t.w);
            } else {
                x[k] = src[i + k * step];
            }
        }
#pragma unroll
        for (int k = 0; k < U; ++k) { acc[k].x += x[k].x; acc[k].y += x[k].y; acc[k].z += x[k].z; acc[k].w += x[k].w; }
    }
    for (; i < end; i += step) { const float4 x = src[i]; acc[0].x += x.x; acc[0].y += x.y; acc[0].z += x.z; acc[0].w += x.w; }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < U; ++k) s += acc[k].x + acc[k].y + acc[k].z + acc[k].w;
    sink[(long long)blockIdx.x * 256 + threadIdx.x] = s;
}

extern "C" int membench_read2(const void* src, long long n, void* sink, int blocks, int unroll, int chunk, int nt, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const float4* s = (const float4*)src;
    float* k = (float*)sink;
#define RB_CASE(U, C, N) if (unroll == U && chunk == C && nt == N) { hipLaunchKernelGGL((k_read2<U, C, N>), dim3(blocks), dim3(256), 0, st, s, n, k); return (int)hipGetLastError(); }
    RB_CASE(4, false, false) RB_CASE(8, false, false) RB_CASE(16, false, false)
    RB_CASE(4, true, false) RB_CASE(8, true, false) RB_CASE(16, true, false)
    RB_CASE(4, false, true) RB_CASE(8, false, true) RB_CASE(8, true, true)
#undef RB_CASE
    return -1;
}

// ---- round 3 probe: which load cache-policy bits keep a SMALL re-read buffer resident in the Infinity Cache while a
// LARGE stream goes past it?  Raw buffer loads so that the policy is an operand (aux: 1 = sc0, 2 = nt, 16 = sc1).
template <int AUX>
__global__ void __launch_bounds__(256)
k_read_aux(const float4* __restrict__ src, long long n, float* __restrict__ sink) {
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    const long long per = (n + gridDim.x - 1) / gridDim.x;                 // float4 per workgroup (contiguous chunk)
    const long long base = (long long)blockIdx.x * per;
    long long cnt = n - base < per ? n - base : per;
    if (cnt < 0) cnt = 0;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(src + base), 0, (int)(cnt * 16), 0x00027000);
    float a = 0.f, b = 0.f, c = 0.f, d2 = 0.f;
    for (long long i = threadIdx.x; i < cnt; i += 8 * 256) {
        v4u x[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)((i + k * 256) * 16), 0, AUX);   // out of range: 0
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            a += __builtin_bit_cast(float, x[k].x); b += __builtin_bit_cast(float, x[k].y);
            c += __builtin_bit_cast(float, x[k].z); d2 += __builtin_bit_cast(float, x[k].w);
        }
    }
    sink[(long long)blockIdx.x * 256 + threadIdx.x] = a + b + c + d2;
}

extern "C" int membench_read_aux(const void* src, long long n_float4, void* sink, int blocks, int aux, void* stream) {
    const float4* s = (const float4*)src; float* k = (float*)sink; hipStream_t st = (hipStream_t)stream;
#define RA(A) case A: hipLaunchKernelGGL((k_read_aux<A>), dim3(blocks), dim3(256), 0, st, s, n_float4, k); break;
    switch (aux) { RA(0) RA(1) RA(2) RA(3) RA(16) RA(17) RA(18) RA(19) default: return -1; }
#undef RA
    return (int)hipGetLastError();
}
