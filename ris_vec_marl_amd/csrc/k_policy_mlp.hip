// f3: the whole PolicyNetwork.forward (sac_agent.py:62-78) of every agent in ONE kernel --
// fc1 + LayerNorm + ReLU -> fc2 -> LayerNorm + ReLU -> {mu, log_std, intent_logits} -- with the fc2
// product on the fp16 matrix cores at float32 accuracy and neither hidden layer ever written to HBM
// (the three-launch form moves 805 + 805 + 268 + 268 MB per step at E = 32 768; this one reads the
// observations and writes the heads).
//
// Orientation.  D = A.B with A = fc2 weight (rows = the 32*MT output features), B = hidden layer
// (columns = 32*NT envs of this wavefront), so after the K loop a LANE holds, for its env (lane & 31),
// half of that env's output features in registers (C/D map of v_mfma_f32_32x32x16: col = lane & 31,
// row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)); the other half sits in lane ^ 32.  LayerNorm over
// the features and the head GEMV are therefore in-register sums plus ONE cross-lane exchange.
//
// B operand, generated on the fly.  fc1 has K = input_dims (5): a lane computes the 8 hidden features
// of its k-slot directly from the observation.  LayerNorm needs the row statistics first; fc1 being
// linear they have a closed form: with the weights centred over the feature axis, Wc[i][f] = W1[i][f] -
// mean_f W1[i] (bias as the row i = IN with x = 1), z_f - mean = sum_i x_i Wc[i][f] and var = x^T G x,
// G = Wc Wc^T / F1 (a sum of squares: no cancellation).  Wc and G are prepared once per weight update.
//
// Precision.  Both operands are split, x = hi + lo with hi = fp16(x), lo = fp16(x - hi), and the three
// significant partial products hi.hi + hi.lo + lo.hi taken as three MFMAs per 16-feature chunk with float32
// accumulation: 2^-22 relative per product.  The weight is pre-multiplied by a power of two (per agent, its
// largest entry lands in [64, 128)) so that its low part stays in the float16 normal range; the accumulator
// is multiplied back (exactly) when the fc2 bias is added.  The low part of a small activation (< 0.125) can be
// a float16 subnormal: absolute error <= 2^-25 per such product, below the float32 rounding of the sum.
//
// A operand.  The split weight is stored in FRAGMENT order [chunk][hi|lo][m][lane][8 halfs], so a chunk is
// 2*MT KiB of contiguous memory that goes global -> LDS by LDS-direct loads (no registers) into the buffer
// the MFMAs of the current chunk are not reading, and every wavefront reads it back conflict-free.
//
// One wavefront per SIMD (256 accumulator registers per lane), so latency is hidden INSIDE the wavefront:
// the B fragments of chunk c+1 are computed while the MFMAs of chunk c run (the group barriers lay the two
// independent instruction streams out interleaved).
#include "risvec_launch.hpp"
#include "risvec_step.hpp"

namespace risvec {
namespace {

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x8_t __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(1))) void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;

struct MlpArgs {
    int E, V, IN, F1, H;
    const float* obs;        // [E,V,IN]
    const float* Wc;         // [V,IN1,F1] centred fc1 weight x LayerNorm-1 weight, row IN = centred bias, rows above = 0
    const float* G;          // [V,IN1,IN1] Gram matrix of the centred weight (without the LayerNorm weight) / F1
    const float* ln1b;       // [V,F1]
    const uint4* W2f;        // [V,F1/16,2,MT,64] 16-byte fragments of the scaled weight: hi, lo
    const float* gscale;     // [V] 2^-S, undoes the weight scaling
    const float* b2; const float* ln2w; const float* ln2b;   // [V,F2]
    const uint4* WhF;        // [V,MT,2,2,64] 16-byte fragments of the scaled head weight (32 rows, zero padded): hi, lo
    const float* hscale;     // [V] undoes the head-weight scaling
    const float* bh;         // [V,H]
    float* heads;            // [V,E,H]
};

constexpr float kLnEps = 1e-5f;

template <int MT, int NT, int IN1>
__global__ void __launch_bounds__(kBlock)
k_policy_mlp(MlpArgs A) {
    constexpr int F2 = 32 * MT;
    constexpr int kChunkVec = 2 * MT * kWave;                 // uint4 per chunk
    constexpr int kStage = kChunkVec / kBlock;                // uint4 per thread per chunk
    static_assert(kChunkVec % kBlock == 0, "chunk must split evenly over the workgroup");
    extern __shared__ uint4 s_raw[];
    uint4* s_a = s_raw;                                       // [2][2][MT][64]
    float* s_wc = reinterpret_cast<float*>(s_a + 2 * kChunkVec);   // [IN1][F1]
    const int F1 = A.F1, H = A.H;
    float* s_l1 = s_wc + IN1 * F1;                            // [F1]: ln1 bias
    float* s_p2 = s_l1 + F1;                                  // [3][F2]: b2, ln2 weight, bias
    float* s_bh = s_p2 + 3 * F2;                              // [32]: head bias, zero padded
    float* s_g = s_bh + 32;                                   // [IN1][IN1]

    const int v = blockIdx.y, tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    for (int i = tid; i < IN1 * F1; i += kBlock) s_wc[i] = A.Wc[(size_t)v * IN1 * F1 + i];
    for (int i = tid; i < F1; i += kBlock) s_l1[i] = A.ln1b[(size_t)v * F1 + i];
    for (int i = tid; i < F2; i += kBlock) {
        s_p2[i] = A.b2[(size_t)v * F2 + i]; s_p2[F2 + i] = A.ln2w[(size_t)v * F2 + i]; s_p2[2 * F2 + i] = A.ln2b[(size_t)v * F2 + i];
    }
    for (int i = tid; i < 32; i += kBlock) s_bh[i] = i < H ? A.bh[(size_t)v * H + i] : 0.0f;
    for (int i = tid; i < IN1 * IN1; i += kBlock) s_g[i] = A.G[(size_t)v * IN1 * IN1 + i];
    const int NC = F1 / 16;
    const uint4* wsrc = A.W2f + (size_t)v * NC * kChunkVec;
#pragma unroll
    for (int q = 0; q < kStage; ++q) s_a[q * kBlock + tid] = wsrc[q * kBlock + tid];
    __syncthreads();

    // this lane's envs (one per N tile), their inputs and LayerNorm-1 scale
    const long long row0 = ((long long)blockIdx.x * (kBlock / kWave) + wave) * (32 * NT);
    float x[NT][IN1], rstd[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const long long e = row0 + 32 * n + r;
        const float* xin = A.obs + ((e < A.E ? e : 0) * A.V + v) * A.IN;
#pragma unroll
        for (int i = 0; i < IN1; ++i) x[n][i] = i < A.IN ? xin[i < A.IN ? i : 0] : (i == A.IN ? 1.0f : 0.0f);
        float var = 0.0f;
#pragma unroll
        for (int i = 0; i < IN1; ++i) {
            float s = 0.0f;
#pragma unroll
            for (int k = 0; k < IN1; ++k) s = fmaf(s_g[i * IN1 + k], x[n][k], s);
            var = fmaf(x[n][i], s, var);
        }
        rstd[n] = rsqrtf(fmaxf(var, 0.0f) + kLnEps);
#pragma unroll
        for (int i = 0; i < IN1; ++i) x[n][i] *= rstd[n];        // the fc1 rows already carry the LayerNorm weight
    }

    f32x16_t acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[m][n][q] = 0.0f;

    // hidden features 16c + 8h .. +7 of this lane's k-slot for each of its envs, split into hi and lo
    auto make_b = [&](int c, half8_t (&bf)[2][NT]) {
        const int f0 = 16 * c + 8 * h;
        f32x8_t wc[IN1];
#pragma unroll
        for (int i = 0; i < IN1; ++i) wc[i] = *reinterpret_cast<const f32x8_t*>(s_wc + i * F1 + f0);
        const f32x8_t lb = *reinterpret_cast<const f32x8_t*>(s_l1 + f0);
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            f32x8_t a = lb;
#pragma unroll
            for (int i = 0; i < IN1; ++i) a += wc[i] * x[n][i];
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = fmaxf(a[j], 0.0f);
            const half8_t hi = __builtin_convertvector(a, half8_t);
            bf[0][n] = hi;
            bf[1][n] = __builtin_convertvector(a - __builtin_convertvector(hi, f32x8_t), half8_t);
        }
    };

    auto mfma_chunk = [&](const uint4* sa, const half8_t (&bf)[2][NT]) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const half8_t ah = __builtin_bit_cast(half8_t, sa[m * kWave + lane]);
            const half8_t al = __builtin_bit_cast(half8_t, sa[(MT + m) * kWave + lane]);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bf[0][n], acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bf[0][n], acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bf[1][n], acc[m][n], 0, 0, 0);
            }
        }
    };

    half8_t bcur[2][NT], bnext[2][NT];
    make_b(0, bcur);
    int buf = 0;
    for (int c = 0; c + 1 < NC; ++c) {
        // fragments of the next chunk first: their LDS reads (fc1 weight, LayerNorm-1 bias) must precede the
        // LDS-direct loads in program order, or the compiler, unable to tell the two LDS regions apart, drains
        // the loads (s_waitcnt vmcnt(0)) before the first such read
        make_b(c + 1, bnext);
        {
            const uint4* src = wsrc + (size_t)(c + 1) * kChunkVec;
            uint4* sb = s_a + (buf ^ 1) * kChunkVec;
#pragma unroll
            for (int q = 0; q < kStage; ++q)
                __builtin_amdgcn_global_load_lds((const gvoid_t*)(src + q * kBlock + tid), (lvoid_t*)(sb + q * kBlock + tid),
                                                 16, 0, 0);
        }
        mfma_chunk(s_a + buf * kChunkVec, bcur);
#pragma unroll
        for (int i = 0; i < MT * NT; ++i) {                             // per 3 MFMAs: 2 LDS reads, 7 VALU
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int n = 0; n < NT; ++n) bcur[t][n] = bnext[t][n];
        __syncthreads();
        buf ^= 1;
    }
    mfma_chunk(s_a + buf * kChunkVec, bcur);

    // ---- fc2 bias + LayerNorm + ReLU, in registers: this lane owns features 32m + (q & 3) + 8 (q >> 2) + 4h of its env
    // The head weight (fragment order, 4*MT KiB) takes over the staging buffers.
    __syncthreads();
    {
        const uint4* hsrc = A.WhF + (size_t)v * (4 * MT * kWave);
        for (int i = tid; i < 4 * MT * kWave; i += kBlock) s_a[i] = hsrc[i];
    }
    const float inv_f2 = 1.0f / (float)F2, unscale = A.gscale[v];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        float s = 0.0f;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 b = *reinterpret_cast<const float4*>(s_p2 + 32 * m + 8 * g + 4 * h);
                const float bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    acc[m][n][4 * g + k] = fmaf(acc[m][n][4 * g + k], unscale, bb[k]);
                    s += acc[m][n][4 * g + k];
                }
            }
        s += __shfl_xor(s, 32, kWave);
        const float mean = s * inv_f2;
        float s2 = 0.0f;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int q = 0; q < 16; ++q) { const float d = acc[m][n][q] - mean; s2 = fmaf(d, d, s2); }
        s2 += __shfl_xor(s2, 32, kWave);
        const float rs = rsqrtf(s2 * inv_f2 + kLnEps);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 w = *reinterpret_cast<const float4*>(s_p2 + F2 + 32 * m + 8 * g + 4 * h);
                const float4 b = *reinterpret_cast<const float4*>(s_p2 + 2 * F2 + 32 * m + 8 * g + 4 * h);
                const float ww[4] = {w.x, w.y, w.z, w.w}, bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    acc[m][n][4 * g + k] = fmaxf(fmaf((acc[m][n][4 * g + k] - mean) * rs, ww[k], bb[k]), 0.0f);
            }
    }
    __syncthreads();

    // ---- heads on the matrix cores: D[head][env] = Wh^T . y with the accumulator registers themselves as the B
    // operand (registers 8u .. 8u+7 of tile m are k-step (m, u); the head weight was laid out in that k order),
    // split hi + lo like the fc2 product
    f32x16_t hacc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int q = 0; q < 16; ++q) hacc[n][q] = 0.0f;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const half8_t wh = __builtin_bit_cast(half8_t, s_a[((m * 2 + u) * 2 + 0) * kWave + lane]);
            const half8_t wl = __builtin_bit_cast(half8_t, s_a[((m * 2 + u) * 2 + 1) * kWave + lane]);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                f32x8_t y;
#pragma unroll
                for (int j = 0; j < 8; ++j) y[j] = acc[m][n][8 * u + j];
                const half8_t yh = __builtin_convertvector(y, half8_t);
                const half8_t yl = __builtin_convertvector(y - __builtin_convertvector(yh, f32x8_t), half8_t);
                hacc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, yh, hacc[n], 0, 0, 0);
                hacc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, yh, hacc[n], 0, 0, 0);
                hacc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, yl, hacc[n], 0, 0, 0);
            }
        }
    const float hs = A.hscale[v];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const long long e = row0 + 32 * n + r;
        if (e >= A.E) continue;
        float* o = A.heads + ((size_t)v * A.E + e) * H;
#pragma unroll
        for (int q = 0; q < 12; ++q) {                           // head rows (q & 3) + 8 (q >> 2) + 4h < 24
            const int hd = (q & 3) + 8 * (q >> 2) + 4 * h;
            if (hd < H) o[hd] = fmaf(hacc[n][q], hs, s_bh[hd]);
        }
    }
}

template <int MT, int NT, int IN1>
hipError_t launch_mlp(const MlpArgs& a, hipStream_t st) {
    const int F2 = 32 * MT;
    const size_t lds = (size_t)2 * 2 * MT * kWave * sizeof(uint4)
                       + ((size_t)IN1 * a.F1 + a.F1 + 3 * F2 + 32 + IN1 * IN1) * sizeof(float);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    auto kern = k_policy_mlp<MT, NT, IN1>;
    if (lds > 64 * 1024) {
        hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (err != hipSuccess) return err;
    }
    const int rows = (kBlock / kWave) * 32 * NT;
    hipLaunchKernelGGL(kern, dim3((unsigned)((a.E + rows - 1) / rows), (unsigned)a.V), dim3(kBlock), lds, st, a);
    return hipGetLastError();
}

}  // namespace

// Instantiated for the reference's shapes: input_dims <= 5, fc2 = 256 or 128, up to 24 heads (4 + V, V <= 20).  Anything else reports hipErrorInvalidValue and the caller uses the three-launch path.
bool policy_mlp_supported(int IN, int F1, int F2, int H) {
    return IN >= 1 && IN <= 5 && F1 >= 16 && F1 % 16 == 0 && F1 <= 1024 && (F2 == 256 || F2 == 128) && H >= 1 && H <= 24;
}

hipError_t launch_policy_mlp(int E, int V, int IN, int F1, int F2, int H, const float* obs, const float* Wc, const float* G,
                             const float* ln1b, const void* W2f, const float* gscale, const float* b2, const float* ln2w,
                             const float* ln2b, const void* WhF, const float* hscale, const float* bh, float* heads,
                             hipStream_t st) {
    if (!policy_mlp_supported(IN, F1, F2, H)) return hipErrorInvalidValue;
    MlpArgs a{E, V, IN, F1, H, obs, Wc, G, ln1b, static_cast<const uint4*>(W2f), gscale, b2, ln2w, ln2b,
              static_cast<const uint4*>(WhF), hscale, bh, heads};
    return F2 == 256 ? launch_mlp<8, 2, 6>(a, st) : launch_mlp<4, 2, 6>(a, st);
}

}  // namespace risvec
