// f3, batched choose_action: the sampling epilogue of the reference's SAC policy
// (sac_agent.py, SAC below) for every (env, agent) at once, fused with the marshalling of
// marl_train_bcd.py (TRAIN).  The three small GEMMs of PolicyNetwork.forward (SAC:62-78) are library
// work (rocBLAS through torch.bmm, see policy.py); this kernel takes their output -- per agent a row
// (mu[2], log_std[2], intent_logits[V]) -- and does everything after it in one pass:
//   log_std clamp (SAC:72), Normal(mu, std).sample() and tanh (SAC:83-86), the feasibility mask on the
//   logits incl. the all-zero-row rule (SAC:91-104), the soft Gumbel-softmax (SAC:110-113), the arg-max
//   one-hot of choose_action (SAC:215-216), and -- optionally -- the env action, pairing power and replay
//   action row of TRAIN:1386-1396, 1601-1608, 1776-1784 (what k_marshal_* would do in a second launch).
// Draws are injected (parity) or Philox.  pow2ceil(V) lanes per (env, agent) row; V <= 64.
#include <cfloat>

#include "risvec_launch.hpp"
#include "risvec_step.hpp"      // DPP exchanges (xchg / gsum)

namespace risvec {
namespace {

constexpr uint32_t kSitePolicyEps = 9, kSitePolicyGumbel = 10;

struct PolicyArgs {
    int E, V;
    long long env_offset;
    const float* heads;        // [V, E, 4 + V]
    const uint8_t* mask;       // [E, V, V] or NULL
    const float* tau;          // [V]
    const uint8_t* hard;       // [V] or NULL: straight-through one-hot (F.gumbel_softmax(hard=True)) per agent
    const float* eps;          // [E, V, 2] or NULL
    const float* expo;         // [E, V, V] or NULL
    uint64_t seed;
    uint32_t counter;
    float floor_eff;
    float* power_raw;          // [E, V, 2]
    float* probs;              // [E, V, V]
    float* onehot;             // [E, V, V] or NULL
    float* action_env;         // [E, 2, V] or NULL
    float* p_off01;            // [E, V] or NULL
    float* action_store;       // [E, V, V + 2] or NULL
};

// VP = pow2ceil(V) lanes per (env, agent) row, 64 / VP rows per wavefront: lane k owns logit k, so the
// heads / draws / mask rows are read and the probability rows written with consecutive lanes on consecutive
// words; max / arg-max / sum over a row are log2(VP) DPP exchanges.  Lane 0 of a row also owns the power head.
template <int VP>
__global__ void __launch_bounds__(kBlock)
k_policy_sample(PolicyArgs A) {
    const int V = A.V, H = 4 + V;
    const long long t = (long long)blockIdx.x * kBlock + threadIdx.x;
    const long long row_raw = t / VP, n_rows = (long long)A.E * V;
    const int k = (int)(t % VP);
    const bool live_row = row_raw < n_rows;
    const long long gid = live_row ? row_raw : n_rows - 1;     // dead lanes shadow the last row, store nothing
    const bool mine = live_row && k < V;
    const long long e = gid / V;
    const int v = (int)(gid % V);
    const float* h = A.heads + ((long long)v * A.E + e) * H;
    const uint32_t genv = (uint32_t)(A.env_offset + e);
    // ---- discrete head: masked logits, Gumbel-softmax  (SAC:91-113) --------------------------------------
    const bool open_k = k < V && (!A.mask || A.mask[gid * V + k] != 0);
    const bool any_open = gsum<VP>(A.mask && open_k ? 1.0f : 0.0f) > 0.0f;
    const bool blocked = A.mask && any_open && !open_k;        // an all-zero row is opened up (SAC:97-100)
    float z = -INFINITY;
    if (k < V) {
        float ex;
        if (A.expo) ex = A.expo[gid * V + k];
        else {
            const uint4 r = philox4x32_10(genv, (uint32_t)v, A.counter, kSitePolicyGumbel + 0x100u * (k >> 2), A.seed);
            const uint32_t x = (k & 3) == 0 ? r.x : (k & 3) == 1 ? r.y : (k & 3) == 2 ? r.z : r.w;
            ex = -logf(((float)(x >> 8) + 1.0f) * 0x1p-24f);          // Exp(1), u in (0, 1]
        }
        const float ml = blocked ? -FLT_MAX / 2.0f : h[4 + k];        // torch.finfo(float32).min / 2  (SAC:103)
        z = (ml + -logf(ex)) / A.tau[v];                              // (logits + gumbel) / tau
    }
    float zmax = z;
    int arg = k < V ? k : 0x7fffffff;
#pragma unroll
    for (int o = 1; o < VP; o <<= 1) {                                // arg-max, first index on ties
        float oz; int oa;
        if (o == 1) { oz = xchg<1>(zmax); oa = __builtin_bit_cast(int, xchg<1>(__builtin_bit_cast(float, arg))); }
        else if (o == 2) { oz = xchg<2>(zmax); oa = __builtin_bit_cast(int, xchg<2>(__builtin_bit_cast(float, arg))); }
        else if (o == 4) { oz = xchg<4>(zmax); oa = __builtin_bit_cast(int, xchg<4>(__builtin_bit_cast(float, arg))); }
        else if (o == 8) { oz = xchg<8>(zmax); oa = __builtin_bit_cast(int, xchg<8>(__builtin_bit_cast(float, arg))); }
        else if (o == 16) { oz = xchg<16>(zmax); oa = __builtin_bit_cast(int, xchg<16>(__builtin_bit_cast(float, arg))); }
        else { oz = xchg<32>(zmax); oa = __builtin_bit_cast(int, xchg<32>(__builtin_bit_cast(float, arg))); }
        if (oz > zmax || (oz == zmax && oa < arg)) { zmax = oz; arg = oa; }
    }
    const float ez = k < V ? expf(z - zmax) : 0.0f;
    const float sum = gsum<VP>(ez);
    float pk = ez / sum;
    if (A.hard && A.hard[v]) pk = ((k == arg ? 1.0f : 0.0f) - pk) + pk;   // y_hard - y_soft + y_soft (SAC:110-113)
    if (mine) {
        A.probs[gid * V + k] = pk;
        if (A.onehot) A.onehot[gid * V + k] = k == arg ? 1.0f : 0.0f;     // choose_action, SAC:215-216
        if (A.action_store) A.action_store[gid * (V + 2) + k] = k == v ? 0.0f : pk;   // TRAIN:1390, 1776-1784
    }
    // ---- continuous head + marshalling, lane 0 of the row  (SAC:72, 83-86; TRAIN:1391-1396, 1601-1608) ---
    if (live_row && k == 0) {
        float e0, e1;
        if (A.eps) { e0 = A.eps[gid * 2]; e1 = A.eps[gid * 2 + 1]; }
        else {
            const uint4 r = philox4x32_10(genv, (uint32_t)v, A.counter, kSitePolicyEps, A.seed);
            const float2 n = normal2(r.x, r.y);
            e0 = n.x; e1 = n.y;
        }
        const float ls0 = fminf(fmaxf(h[2], -20.0f), 2.0f), ls1 = fminf(fmaxf(h[3], -20.0f), 2.0f);
        const float p0 = tanhf(e0 * expf(ls0) + h[0]), p1 = tanhf(e1 * expf(ls1) + h[1]);
        A.power_raw[gid * 2] = p0;
        A.power_raw[gid * 2 + 1] = p1;
        if (A.action_store) {
            A.action_store[gid * (V + 2) + V] = p0;
            A.action_store[gid * (V + 2) + V + 1] = p1;
        }
        const float m0 = (fminf(fmaxf(p0, -0.999f), 0.999f) + 1.0f) / 2.0f;
        const float m1 = (fminf(fmaxf(p1, -0.999f), 0.999f) + 1.0f) / 2.0f;
        if (A.action_env) {
            A.action_env[(e * 2 + 0) * V + v] = m0;
            A.action_env[(e * 2 + 1) * V + v] = fmaxf(m1, A.floor_eff);
        }
        if (A.p_off01) A.p_off01[gid] = m0;
    }
}

// ---------------------------------------------------------------------------------------------
// The two ends of PolicyNetwork.forward (SAC:62-78) that are NOT GEMM-shaped, hand-written so that
// only the fc2 product (fc1 x fc2 per agent) goes to rocBLAS:
//   k_policy_layer1  x[in] -> relu(LayerNorm(fc1 x + b1)) : in = 5, so the "GEMM" is 5 FMAs per output;
//                    one wavefront per (agent, env) row, the agent's fc1 weights / LayerNorm affine in LDS,
//                    the row's features strided over the lanes (coalesced 256-byte stores);
//   k_policy_heads   g[fc2] (= fc2 h1 + b2 from the GEMM) -> relu(LayerNorm(g)) -> the 4+V head outputs
//                    (mu, log_std, intent_logits): a GEMV per row with the head matrix in LDS and one
//                    wave reduction per output; the normalised hidden layer never goes back to HBM.
// LayerNorm: biased variance over the features, eps 1e-5, two passes over the registers (as torch).
// ---------------------------------------------------------------------------------------------
constexpr float kLnEps = 1e-5f;

template <int FPL>
__device__ __forceinline__ void layer_norm_relu(float (&x)[FPL], int F, int lane, const float* w, const float* b) {
    float s = 0.0f;
#pragma unroll
    for (int t = 0; t < FPL; ++t) s += (lane + t * kWave < F) ? x[t] : 0.0f;
    s = group_sum<kWave>(s);
    const float mean = s / (float)F;
    float q = 0.0f;
#pragma unroll
    for (int t = 0; t < FPL; ++t) { const float d = x[t] - mean; q += (lane + t * kWave < F) ? d * d : 0.0f; }
    q = group_sum<kWave>(q);
    const float rstd = 1.0f / sqrtf(q / (float)F + kLnEps);
#pragma unroll
    for (int t = 0; t < FPL; ++t) {
        const int j = lane + t * kWave;
        if (j < F) x[t] = fmaxf((x[t] - mean) * rstd * w[j] + b[j], 0.0f);
    }
}

template <int FPL>
__global__ void __launch_bounds__(kBlock)
k_policy_layer1(int E, int V, int IN, int F, const float* __restrict__ obs, const float* __restrict__ W1,
                const float* __restrict__ b1, const float* __restrict__ lw, const float* __restrict__ lb,
                float* __restrict__ out, int rows_per_wave) {
    extern __shared__ float s_par[];                   // [IN + 3][F]: W1 rows, b1, ln weight, ln bias
    const int v = blockIdx.y;
    for (int i = threadIdx.x; i < IN * F; i += kBlock) s_par[i] = W1[(long long)v * IN * F + i];
    for (int i = threadIdx.x; i < F; i += kBlock) {
        s_par[IN * F + i] = b1[(long long)v * F + i];
        s_par[(IN + 1) * F + i] = lw[(long long)v * F + i];
        s_par[(IN + 2) * F + i] = lb[(long long)v * F + i];
    }
    __syncthreads();
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const long long e0 = ((long long)blockIdx.x * (kBlock / kWave) + wave) * rows_per_wave;
    for (int r = 0; r < rows_per_wave; ++r) {
        const long long e = e0 + r;
        if (e >= E) break;
        const float* x = obs + (e * V + v) * IN;       // wave-uniform: broadcast loads
        float h[FPL];
#pragma unroll
        for (int t = 0; t < FPL; ++t) {
            const int j = lane + t * kWave;
            float acc = 0.0f;
            if (j < F) {
                acc = s_par[IN * F + j];
                for (int i = 0; i < IN; ++i) acc = fmaf(x[i], s_par[i * F + j], acc);
            }
            h[t] = acc;
        }
        layer_norm_relu<FPL>(h, F, lane, s_par + (IN + 1) * F, s_par + (IN + 2) * F);
        float* o = out + ((long long)v * E + e) * F;
#pragma unroll
        for (int t = 0; t < FPL; ++t) { const int j = lane + t * kWave; if (j < F) o[j] = h[t]; }
    }
}

template <int FPL>
__global__ void __launch_bounds__(kBlock)
k_policy_heads(int E, int V, int F, int H, const float* __restrict__ g, const float* __restrict__ b2,
               const float* __restrict__ lw, const float* __restrict__ lb, const float* __restrict__ Wh,
               const float* __restrict__ bh, float* __restrict__ heads, int rows_per_wave) {
    extern __shared__ float s_par[];                   // ln weight [F], ln bias [F], Wh [F][H], bh [H]
    const int v = blockIdx.y;
    float* s_lw = s_par; float* s_lb = s_par + F; float* s_wh = s_par + 2 * F; float* s_bh = s_wh + (long long)F * H;
    for (int i = threadIdx.x; i < F; i += kBlock) { s_lw[i] = lw[(long long)v * F + i]; s_lb[i] = lb[(long long)v * F + i]; }
    for (int i = threadIdx.x; i < F * H; i += kBlock) s_wh[i] = Wh[(long long)v * F * H + i];
    for (int i = threadIdx.x; i < H; i += kBlock) s_bh[i] = bh[(long long)v * H + i];
    __syncthreads();
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const long long e0 = ((long long)blockIdx.x * (kBlock / kWave) + wave) * rows_per_wave;
    for (int r = 0; r < rows_per_wave; ++r) {
        const long long e = e0 + r;
        if (e >= E) break;
        const float* row = g + ((long long)v * E + e) * F;
        float h[FPL];
#pragma unroll
        for (int t = 0; t < FPL; ++t) {
            const int j = lane + t * kWave;
            h[t] = j < F ? row[j] + (b2 ? b2[(long long)v * F + j] : 0.0f) : 0.0f;
        }
        layer_norm_relu<FPL>(h, F, lane, s_lw, s_lb);
        float* o = heads + ((long long)v * E + e) * H;
        for (int k = 0; k < H; ++k) {                  // one output at a time: partial dot, wave sum
            float acc = 0.0f;
#pragma unroll
            for (int t = 0; t < FPL; ++t) { const int j = lane + t * kWave; if (j < F) acc = fmaf(h[t], s_wh[j * H + k], acc); }
            acc = group_sum<kWave>(acc);
            if (lane == 0) o[k] = acc + s_bh[k];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// F % 4 == 0 (every size the reference uses): 16 lanes per row, 4 rows per wavefront.  A lane owns
// float4 chunks c, c+16, ... of its row, so a row is read / written as 256 contiguous bytes per
// instruction; the LayerNorm sums are 4-step DPP reductions inside the 16-lane group (plain VALU, no
// LDS crossbar); the head GEMV keeps 16 partial outputs per lane and collapses them with a TRANSPOSING
// reduction (8 + 4 + 2 + 1 exchanges: lane c ends up owning output c) instead of one full wave
// reduction per output.
// ---------------------------------------------------------------------------------------------
template <int K, int O>
__device__ __forceinline__ void treduce16(float (&val)[16], int c) {
    if constexpr (K > 1) {
        const bool hi = (c & O) != 0;                 // mirrors flip the lower (unused) bits too: go high -> low
#pragma unroll
        for (int j = 0; j < K / 2; ++j) {
            const float send = hi ? val[j] : val[j + K / 2];
            const float keep = hi ? val[j + K / 2] : val[j];
            val[j] = keep + xchg<O>(send);
        }
        treduce16<K / 2, O / 2>(val, c);
    }
}

template <int NT>
__device__ __forceinline__ void layer_norm_relu16(float4 (&x)[NT], int F, int c, const float* w, const float* b) {
    float s = 0.0f;
#pragma unroll
    for (int t = 0; t < NT; ++t) s += (x[t].x + x[t].y) + (x[t].z + x[t].w);     // chunks past F hold zeros
    s = gsum<16>(s);
    const float mean = s / (float)F;
    float q = 0.0f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
        if ((c + 16 * t) * 4 < F) {
            const float dx = x[t].x - mean, dy = x[t].y - mean, dz = x[t].z - mean, dw = x[t].w - mean;
            q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
    q = gsum<16>(q);
    const float rstd = 1.0f / sqrtf(q / (float)F + kLnEps);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int j = (c + 16 * t) * 4;
        if (j < F) {
            const float4 ww = *reinterpret_cast<const float4*>(w + j), bb = *reinterpret_cast<const float4*>(b + j);
            x[t] = make_float4(fmaxf((x[t].x - mean) * rstd * ww.x + bb.x, 0.0f), fmaxf((x[t].y - mean) * rstd * ww.y + bb.y, 0.0f),
                               fmaxf((x[t].z - mean) * rstd * ww.z + bb.z, 0.0f), fmaxf((x[t].w - mean) * rstd * ww.w + bb.w, 0.0f));
        }
    }
}

// SPLIT16: instead of the float32 row, write the row as the three float16 K-blocks of the split product
//     a.b ~= hi(a) hi(b) + hi(a) lo(b) + lo(a) hi(b),   hi = fp16(x), lo = x - hi  (|lo| <= 2^-11 |x|),
// [ hi(a) | hi(a) 2^-5 | lo(a) 2^6 ], to be multiplied by [ hi(b) ; lo(b) 2^5 ; hi(b) 2^-6 ] in ONE fp16 GEMM with
// float32 accumulation (K three times as long).  The power-of-two factors keep the low parts out of the
// float16 subnormal range and cancel exactly; what is dropped, lo(a) lo(b) and the rounding of the low
// parts, is 2^-22 relative per product -- float32-GEMM accuracy at the fp16 MFMA rate.
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));

template <int NT, int INMAX, bool SPLIT16>
__global__ void __launch_bounds__(kBlock)
k_policy_layer1_v4(int E, int V, int IN, int F, const float* __restrict__ obs, const float* __restrict__ W1,
                   const float* __restrict__ b1, const float* __restrict__ lw, const float* __restrict__ lb,
                   void* __restrict__ out_any, int iters) {
    extern __shared__ float s_par[];                   // [IN + 3][F]: W1 rows, b1, ln weight, ln bias
    const int v = blockIdx.y;
    for (int i = threadIdx.x; i < IN * F; i += kBlock) s_par[i] = W1[(long long)v * IN * F + i];
    for (int i = threadIdx.x; i < F; i += kBlock) {
        s_par[IN * F + i] = b1[(long long)v * F + i];
        s_par[(IN + 1) * F + i] = lw[(long long)v * F + i];
        s_par[(IN + 2) * F + i] = lb[(long long)v * F + i];
    }
    __syncthreads();
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6, r = lane >> 4, c = lane & 15;
    const long long e0 = ((long long)blockIdx.x * (kBlock / kWave) + wave) * iters * 4;
    for (int it = 0; it < iters; ++it) {
        const long long e = e0 + it * 4 + r;
        const bool live = e < E;                       // dead rows still take part in the DPP exchanges
        const float* xin = obs + ((live ? e : 0) * V + v) * IN;
        float xi[INMAX];
#pragma unroll
        for (int i = 0; i < INMAX; ++i) xi[i] = i < IN ? xin[i] : 0.0f;
        float4 h[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int j = (c + 16 * t) * 4;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j < F) {
                acc = *reinterpret_cast<const float4*>(s_par + IN * F + j);
#pragma unroll
                for (int i = 0; i < INMAX; ++i)
                    if (i < IN) {
                        const float4 ww = *reinterpret_cast<const float4*>(s_par + i * F + j);
                        acc.x = fmaf(xi[i], ww.x, acc.x); acc.y = fmaf(xi[i], ww.y, acc.y);
                        acc.z = fmaf(xi[i], ww.z, acc.z); acc.w = fmaf(xi[i], ww.w, acc.w);
                    }
            }
            h[t] = acc;
        }
        layer_norm_relu16<NT>(h, F, c, s_par + (IN + 1) * F, s_par + (IN + 2) * F);
        if (live) {
            if constexpr (SPLIT16) {
                _Float16* o = static_cast<_Float16*>(out_any) + ((long long)v * E + e) * 3 * F;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int j = (c + 16 * t) * 4;
                    if (j >= F) continue;
                    const float x[4] = {h[t].x, h[t].y, h[t].z, h[t].w};
                    half4_t hi, hs, lo;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        hi[k] = (_Float16)x[k];
                        hs[k] = (_Float16)((float)hi[k] * 0.03125f);
                        lo[k] = (_Float16)((x[k] - (float)hi[k]) * 64.0f);
                    }
                    *reinterpret_cast<half4_t*>(o + j) = hi;
                    *reinterpret_cast<half4_t*>(o + F + j) = hs;
                    *reinterpret_cast<half4_t*>(o + 2 * F + j) = lo;
                }
            } else {
                float* o = static_cast<float*>(out_any) + ((long long)v * E + e) * F;
#pragma unroll
                for (int t = 0; t < NT; ++t) { const int j = (c + 16 * t) * 4; if (j < F) *reinterpret_cast<float4*>(o + j) = h[t]; }
            }
        }
    }
}

template <int NT>
__global__ void __launch_bounds__(kBlock)
k_policy_heads_v4(int E, int V, int F, int H, const float* __restrict__ g, const float* __restrict__ b2,
                  const float* __restrict__ lw, const float* __restrict__ lb, const float* __restrict__ Wh,
                  const float* __restrict__ bh, float* __restrict__ heads, int iters) {
    extern __shared__ float s_par[];                   // fc2 bias [F], ln weight [F], ln bias [F], Wh [F][H], bh [H]
    const int v = blockIdx.y;
    float* s_b2 = s_par; float* s_lw = s_par + F; float* s_lb = s_par + 2 * F; float* s_wh = s_par + 3 * F;
    float* s_bh = s_wh + (long long)F * H;
    for (int i = threadIdx.x; i < F; i += kBlock) {
        s_b2[i] = b2 ? b2[(long long)v * F + i] : 0.0f;
        s_lw[i] = lw[(long long)v * F + i]; s_lb[i] = lb[(long long)v * F + i];
    }
    // head matrix TRANSPOSED in LDS, [H][F]: a lane's four features of one output are one 16-byte read and
    // the 16 lanes of a row read 256 contiguous bytes (the [F][H] layout put 8 lanes on the same bank)
    for (int i = threadIdx.x; i < F * H; i += kBlock) s_wh[(i % H) * F + i / H] = Wh[(long long)v * F * H + i];
    for (int i = threadIdx.x; i < H; i += kBlock) s_bh[i] = bh[(long long)v * H + i];
    __syncthreads();
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6, r = lane >> 4, c = lane & 15;
    const long long e0 = ((long long)blockIdx.x * (kBlock / kWave) + wave) * iters * 4;
    for (int it = 0; it < iters; ++it) {
        const long long e = e0 + it * 4 + r;
        const bool live = e < E;
        const float* row = g + ((long long)v * E + (live ? e : 0)) * F;
        float4 h[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int j = (c + 16 * t) * 4;
            h[t] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j < F) {
                const float4 x = *reinterpret_cast<const float4*>(row + j), bb = *reinterpret_cast<const float4*>(s_b2 + j);
                h[t] = make_float4(x.x + bb.x, x.y + bb.y, x.z + bb.z, x.w + bb.w);
            }
        }
        layer_norm_relu16<NT>(h, F, c, s_lw, s_lb);
        float* o = heads + ((long long)v * E + e) * H;
        for (int kb = 0; kb < H; kb += 16) {           // 16 outputs at a time
            float acc[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[k] = 0.0f;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (kb + k >= H) continue;
                const float* wr = s_wh + (long long)(kb + k) * F;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int j = (c + 16 * t) * 4;
                    if (j < F) {
                        const float4 w4 = *reinterpret_cast<const float4*>(wr + j);
                        acc[k] = fmaf(h[t].x, w4.x, fmaf(h[t].y, w4.y, fmaf(h[t].z, w4.z, fmaf(h[t].w, w4.w, acc[k]))));
                    }
                }
            }
            treduce16<16, 8>(acc, c);
            if (live && kb + c < H) o[kb + c] = acc[0] + s_bh[kb + c];
        }
    }
}

}  // namespace

static int policy_fpl(int F) {
    const int need = (F + kWave - 1) / kWave;
    for (int f : {1, 2, 4, 8, 16}) if (need <= f) return f;
    return 0;
}

static int policy_nt(int F) {                         // float4 chunks per lane in the 16-lanes-per-row kernels
    const int need = (F / 4 + 15) / 16;
    for (int f : {1, 2, 4, 8, 16}) if (need <= f) return f;
    return 0;
}

hipError_t launch_policy_layer1(int E, int V, int IN, int F, const float* obs, const float* W1, const float* b1,
                                const float* lw, const float* lb, float* out, hipStream_t st) {
    if (F % 4 == 0 && IN <= 8 && policy_nt(F) > 0) {
        const int iters = 4, rows_per_block = (kBlock / kWave) * 4 * iters;
        const dim3 grid((unsigned)((E + rows_per_block - 1) / rows_per_block), (unsigned)V);
        const size_t lds = (size_t)(IN + 3) * F * sizeof(float);
#define RISVEC_L1V(N_) hipLaunchKernelGGL((k_policy_layer1_v4<N_, 8, false>), grid, dim3(kBlock), lds, st, E, V, IN, F, obs, W1, b1, lw, lb, out, iters)
        switch (policy_nt(F)) {
            case 1: RISVEC_L1V(1); break;
            case 2: RISVEC_L1V(2); break;
            case 4: RISVEC_L1V(4); break;
            case 8: RISVEC_L1V(8); break;
            default: RISVEC_L1V(16); break;
        }
#undef RISVEC_L1V
        return hipGetLastError();
    }
    const int rpw = 8, wpb = kBlock / kWave;
    const dim3 grid((unsigned)((E + rpw * wpb - 1) / (rpw * wpb)), (unsigned)V);
    const size_t lds = (size_t)(IN + 3) * F * sizeof(float);
#define RISVEC_L1(FP) hipLaunchKernelGGL((k_policy_layer1<FP>), grid, dim3(kBlock), lds, st, E, V, IN, F, obs, W1, b1, lw, lb, out, rpw)
    switch (policy_fpl(F)) {
        case 1: RISVEC_L1(1); break;
        case 2: RISVEC_L1(2); break;
        case 4: RISVEC_L1(4); break;
        case 8: RISVEC_L1(8); break;
        case 16: RISVEC_L1(16); break;
        default: return hipErrorInvalidValue;
    }
#undef RISVEC_L1
    return hipGetLastError();
}

hipError_t launch_policy_layer1_split16(int E, int V, int IN, int F, const float* obs, const float* W1, const float* b1,
                                        const float* lw, const float* lb, void* out16, hipStream_t st) {
    if (!(F % 4 == 0 && IN <= 8 && policy_nt(F) > 0)) return hipErrorInvalidValue;
    const int iters = 4, rows_per_block = (kBlock / kWave) * 4 * iters;
    const dim3 grid((unsigned)((E + rows_per_block - 1) / rows_per_block), (unsigned)V);
    const size_t lds = (size_t)(IN + 3) * F * sizeof(float);
#define RISVEC_L1S(N_) hipLaunchKernelGGL((k_policy_layer1_v4<N_, 8, true>), grid, dim3(kBlock), lds, st, E, V, IN, F, obs, W1, b1, lw, lb, out16, iters)
    switch (policy_nt(F)) {
        case 1: RISVEC_L1S(1); break;
        case 2: RISVEC_L1S(2); break;
        case 4: RISVEC_L1S(4); break;
        case 8: RISVEC_L1S(8); break;
        default: RISVEC_L1S(16); break;
    }
#undef RISVEC_L1S
    return hipGetLastError();
}

hipError_t launch_policy_heads(int E, int V, int F, int H, const float* g, const float* b2, const float* lw,
                               const float* lb, const float* Wh, const float* bh, float* heads, hipStream_t st) {
    if (F % 4 == 0 && policy_nt(F) > 0) {
        const int iters = 4, rows_per_block = (kBlock / kWave) * 4 * iters;
        const dim3 grid((unsigned)((E + rows_per_block - 1) / rows_per_block), (unsigned)V);
        const size_t lds = ((size_t)3 * F + (size_t)F * H + H) * sizeof(float);
#define RISVEC_HDV(N_) hipLaunchKernelGGL((k_policy_heads_v4<N_>), grid, dim3(kBlock), lds, st, E, V, F, H, g, b2, lw, lb, Wh, bh, heads, iters)
        switch (policy_nt(F)) {
            case 1: RISVEC_HDV(1); break;
            case 2: RISVEC_HDV(2); break;
            case 4: RISVEC_HDV(4); break;
            case 8: RISVEC_HDV(8); break;
            default: RISVEC_HDV(16); break;
        }
#undef RISVEC_HDV
        return hipGetLastError();
    }
    const int rpw = 8, wpb = kBlock / kWave;
    const dim3 grid((unsigned)((E + rpw * wpb - 1) / (rpw * wpb)), (unsigned)V);
    const size_t lds = ((size_t)2 * F + (size_t)F * H + H) * sizeof(float);
#define RISVEC_HD(FP) hipLaunchKernelGGL((k_policy_heads<FP>), grid, dim3(kBlock), lds, st, E, V, F, H, g, b2, lw, lb, Wh, bh, heads, rpw)
    switch (policy_fpl(F)) {
        case 1: RISVEC_HD(1); break;
        case 2: RISVEC_HD(2); break;
        case 4: RISVEC_HD(4); break;
        case 8: RISVEC_HD(8); break;
        case 16: RISVEC_HD(16); break;
        default: return hipErrorInvalidValue;
    }
#undef RISVEC_HD
    return hipGetLastError();
}

hipError_t launch_policy_sample(int E, int V, long long env_offset, const float* heads, const uint8_t* mask,
                                const float* tau, const uint8_t* hard, const float* eps, const float* expo, uint64_t seed,
                                uint32_t counter, float floor_eff, float* power_raw, float* probs, float* onehot,
                                float* action_env, float* p_off01, float* action_store, hipStream_t st) {
    PolicyArgs a{E, V, env_offset, heads, mask, tau, hard, eps, expo, seed, counter, floor_eff, power_raw, probs, onehot,
                 action_env, p_off01, action_store};
    const int vp = pow2_ceil(V);
    const long long n = (long long)E * V * vp;
    const dim3 grid((unsigned)((n + kBlock - 1) / kBlock));
    switch (vp) {
        case 1: hipLaunchKernelGGL(k_policy_sample<1>, grid, dim3(kBlock), 0, st, a); break;
        case 2: hipLaunchKernelGGL(k_policy_sample<2>, grid, dim3(kBlock), 0, st, a); break;
        case 4: hipLaunchKernelGGL(k_policy_sample<4>, grid, dim3(kBlock), 0, st, a); break;
        case 8: hipLaunchKernelGGL(k_policy_sample<8>, grid, dim3(kBlock), 0, st, a); break;
        case 16: hipLaunchKernelGGL(k_policy_sample<16>, grid, dim3(kBlock), 0, st, a); break;
        case 32: hipLaunchKernelGGL(k_policy_sample<32>, grid, dim3(kBlock), 0, st, a); break;
        default: hipLaunchKernelGGL(k_policy_sample<64>, grid, dim3(kBlock), 0, st, a); break;
    }
    return hipGetLastError();
}

}  // namespace risvec
