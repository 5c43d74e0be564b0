// f3: the whole PolicyNetwork.forward (sac_agent.py:62-78) of every agent in ONE kernel --
// fc1 + LayerNorm + ReLU -> fc2 -> LayerNorm + ReLU -> {mu, log_std, intent_logits} -- with the fc2
// product on the fp16 matrix cores at float32 accuracy and neither hidden layer ever written to HBM
// (the three-launch form moves 805 + 805 + 268 + 268 MB per step at E = 32 768; this one reads the
// observations and writes the heads).
//
// Orientation.  D = A.B with A = weights (rows = output features), B = activations (columns = the 32 envs of
// this wavefront), so after the K loop a LANE holds, for its env (lane & 31), half of that env's output
// features in registers (C/D map of v_mfma_f32_32x32x16: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) +
// 4 (lane >> 5)); the other half sits in lane ^ 32.  LayerNorm over the features is therefore in-register sums
// plus ONE cross-lane exchange, and each layer's output is the next MFMA's B operand as it stands.
//
// B operand, generated on the fly -- by the matrix cores too.  LayerNorm needs the row statistics of the fc1
// pre-activation first; fc1 being linear they have a closed form: with the weight centred over the feature
// axis (bias as one more input row with x = 1), z_f - mean = sum_i x_i C[i][f] and var = x^T G x, G = C C^T / F1
// (a sum of squares: no cancellation).  So the normalised, scaled and shifted pre-activation of 32 features is
// ONE small product: rows of A = [C[0..IN-1][f] w_f, C[IN][f] w_f, b_f] (w, b the LayerNorm weight and bias),
// columns of B = [x_0 rstd, .., x_{IN-1} rstd, rstd, 1] per env, K padded to 16.  Its result lands in the C/D
// layout, i.e. with the features in registers and the env on the lane -- which is exactly the B-operand layout of
// the next MFMA when the k order of the fc2 weight is chosen to match (registers 8u .. 8u+7 are k-step u): ReLU and
// the float16 split are all the vector ALU has to do, and no fc1 weight is read by the vector path at all.
//
// Precision.  Both operands are split, x = hi + lo with hi = fp16(x), lo = fp16(x - hi), and the three
// significant partial products hi.hi + hi.lo + lo.hi taken as three MFMAs per 16-feature chunk with float32
// accumulation: 2^-22 relative per product.  The weight is pre-multiplied by a power of two (per agent, its
// largest entry lands in [64, 128)) so that its low part stays in the float16 normal range; the accumulator
// is multiplied back (exactly) when the fc2 bias is added.  The low part of a small activation (< 0.125) can be
// a float16 subnormal: absolute error <= 2^-25 per such product, below the float32 rounding of the sum.
//
// A operand.  The split fc2 weight is stored in FRAGMENT order (per group of 32 hidden features: the fc1 operand
// of the next group, then two chunks [hi|lo][m][lane][8 halfs]), one contiguous stream per agent that goes
// L2 -> LDS by LDS-direct loads (no registers) into a ring of three group slots, two groups ahead of the MFMAs:
// a group's MFMAs are shorter than an L2 round trip.  Counted s_waitcnt vmcnt + raw s_barrier keep the newest
// group in flight across the barrier.  Every wavefront reads its A fragments back conflict-free (lane-linear).
//
// Workgroup = 8 wavefronts (2 per SIMD, 220 registers each incl. 128 accumulators) sharing one weight stream;
// one workgroup per CU (123 KB of LDS).  The head weight rides in on the ring's free slot during the last group.
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
    const float* G;          // [V,6,6] Gram matrix of the centred fc1 rows / F1
    const uint4* W1F;        // [V,2,64] 16-byte fragments of the fc1 operand of group 0 (scaled): hi, lo
    const uint4* W2f;        // [V,F1/32,(8 + 4 MT) 64] the weight stream, per group of 32 hidden features: fc1 operand of the
                             // NEXT group (hi, lo, 6 KiB of padding), then two chunks [hi|lo][MT][64] of the scaled fc2 weight
    const float* gscale;     // [V] undoes the fc1 and fc2 weight scalings
    const float* b2; const float* ln2w; const float* ln2b;   // [V,F2]
    const uint4* WhF;        // [V,MT,2,2,64] 16-byte fragments of the scaled head weight (32 rows, zero padded): hi, lo
    const float* hscale;     // [V] undoes the head-weight scaling
    const float* bh;         // [V,H]
    float* heads;            // [V,E,H]
};

constexpr float kLnEps = 1e-5f;
constexpr int kIn1 = 6;      // input rows of G: up to 5 inputs + the bias row

__device__ __forceinline__ void split16(const f32x8_t& y, half8_t& hi, half8_t& lo) {
    hi = __builtin_convertvector(y, half8_t);
    lo = __builtin_convertvector(y - __builtin_convertvector(hi, f32x8_t), half8_t);
}

constexpr int kMlpBlock = 512;       // 8 wavefronts = 2 per SIMD sharing one weight stream
constexpr int kRing = 3;             // group slots in LDS: one being read, two in flight / landed

template <int MT>
__global__ void __launch_bounds__(kMlpBlock, 2)
k_policy_mlp(MlpArgs A) {
    constexpr int F2 = 32 * MT;
    constexpr int kChunkVec = 2 * MT * kWave;                 // uint4 per chunk of 16 hidden features
    constexpr int kGroupVec = 8 * kWave + 2 * kChunkVec;      // uint4 per group of 32: fc1 operand of the NEXT group (2 of 8 KiB used), 2 chunks
    constexpr int kStage = kGroupVec / kMlpBlock;             // LDS-direct loads per wavefront per group
    static_assert(kGroupVec % kMlpBlock == 0, "group must split evenly over the workgroup");
    constexpr int kHeadStage = 4 * MT * kWave / kMlpBlock;    // LDS-direct loads per wavefront for the head weight
    static_assert((4 * MT * kWave) % kMlpBlock == 0, "head weight must split evenly over the workgroup");
    extern __shared__ uint4 s_raw[];
    const int H = A.H, NG = A.F1 / 32;
    uint4* s_ring = s_raw;                                    // [kRing][kGroupVec]
    float* s_p2 = reinterpret_cast<float*>(s_ring + kRing * kGroupVec);   // [3][F2]: b2, ln2 weight, bias
    float* s_bh = s_p2 + 3 * F2;                              // [32]: head bias, zero padded
    float* s_g = s_bh + 32;                                   // [6][6]

    const int v = blockIdx.y, tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const uint4* wsrc = A.W2f + (size_t)v * NG * kGroupVec;
    auto stage = [&](int g) {                                 // group g of the weight stream, global -> LDS directly
        const uint4* src = wsrc + (size_t)g * kGroupVec;
        uint4* dst = s_ring + (g % kRing) * kGroupVec;
#pragma unroll
        for (int q = 0; q < kStage; ++q)
            __builtin_amdgcn_global_load_lds((const gvoid_t*)(src + q * kMlpBlock + tid), (lvoid_t*)(dst + q * kMlpBlock + tid), 16, 0, 0);
    };
    // The first two groups of the weight stream, this lane's observation and the parameter tables are all requested
    // before anything waits: one L2 round trip in front of the K loop instead of three in a row.
    stage(0);
    if (NG > 1) stage(1);
    const long long row0 = ((long long)blockIdx.x * (kMlpBlock / kWave) + wave) * 32;
    const long long e = row0 + r;
    float x[kIn1];
    {
        const float* xin = A.obs + ((e < A.E ? e : 0) * A.V + v) * A.IN;
#pragma unroll
        for (int i = 0; i < kIn1; ++i) x[i] = i < A.IN ? xin[i < A.IN ? i : 0] : (i == A.IN ? 1.0f : 0.0f);
    }
    for (int i = tid; i < F2; i += kMlpBlock) {
        s_p2[i] = A.b2[(size_t)v * F2 + i]; s_p2[F2 + i] = A.ln2w[(size_t)v * F2 + i]; s_p2[2 * F2 + i] = A.ln2b[(size_t)v * F2 + i];
    }
    for (int i = tid; i < 32; i += kMlpBlock) s_bh[i] = i < H ? A.bh[(size_t)v * H + i] : 0.0f;
    for (int i = tid; i < kIn1 * kIn1; i += kMlpBlock) s_g[i] = A.G[(size_t)v * kIn1 * kIn1 + i];
    // the fc1 operand of group 0 goes to the head of the last slot (free until group 2 is staged)
    if (tid < 2 * kWave) s_ring[(kRing - 1) * kGroupVec + tid] = A.W1F[(size_t)v * 2 * kWave + tid];
    __syncthreads();

    // this lane's env, its LayerNorm-1 scale, and the B operand of the fc1 product: [x rstd, rstd, 1, 0..] in the
    // k-slots of the low lane half, zeros in the high half
    half8_t xh, xl;
    {
        float var = 0.0f;
#pragma unroll
        for (int i = 0; i < kIn1; ++i) {
            float s = 0.0f;
#pragma unroll
            for (int k = 0; k < kIn1; ++k) s = fmaf(s_g[i * kIn1 + k], x[k], s);
            var = fmaf(x[i], s, var);
        }
        const float rstd = rsqrtf(fmaxf(var, 0.0f) + kLnEps);
        f32x8_t xb;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float val = j < kIn1 ? x[j < kIn1 ? j : 0] * rstd : 0.0f;      // the bias row (x = 1) becomes rstd
            xb[j] = h == 0 ? (j == A.IN + 1 ? 1.0f : val) : 0.0f;               // row IN+1 carries the LayerNorm bias
        }
        split16(xb, xh, xl);
    }

    f32x16_t acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[m][q] = 0.0f;

    // LayerNorm-1 output (before the ReLU) of 32 hidden features of this lane's env, C/D layout; w1 = its operand
    auto layer1 = [&](const uint4* w1) {
        const half8_t wh = __builtin_bit_cast(half8_t, w1[lane]);
        const half8_t wl = __builtin_bit_cast(half8_t, w1[kWave + lane]);
        f32x16_t d;
#pragma unroll
        for (int q = 0; q < 16; ++q) d[q] = 0.0f;
        d = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, d, 0, 0, 0);
        return d;
    };
    // registers 8u .. 8u+7 of it -> ReLU -> the split B fragments of k-step u
    auto make_b = [&](const f32x16_t& d, int u, half8_t (&bf)[2]) {
        f32x8_t y;
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] = fmaxf(d[8 * u + j], 0.0f);
        split16(y, bf[0], bf[1]);
    };
    // A fragments are read two feature tiles (six MFMAs) ahead of their use, by hand: hipcc sinks LDS reads next
    // to their first use and then waits lgkmcnt(0), which exposes the LDS latency in front of every third MFMA.
    // The reads are inline asm (issued in program order), and each counted wait names the fragments it releases,
    // so the MFMAs that consume them cannot be moved above it.  Compiler-generated LDS reads may interleave:
    // counters retire in order, so extra reads can only make either side wait longer, never too little.
    auto mfma_chunk = [&](const uint4* sa, const half8_t (&bf)[2]) {
        const uint32_t base = (uint32_t)(size_t)(lvoid_t*)(sa + lane);
        half8_t ah[3], al[3];
#define RISVEC_RD(M_) do {                                                                                          \
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ah[(M_) % 3]) : "v"(base), "n"((M_) * kWave * 16));          \
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(al[(M_) % 3]) : "v"(base), "n"((MT + (M_)) * kWave * 16));   \
        } while (0)
        RISVEC_RD(0);
        if constexpr (MT > 1) RISVEC_RD(1);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            if (m + 2 < MT) {
                if (m == 0) RISVEC_RD(2); else if (m == 1) RISVEC_RD(3); else if (m == 2) RISVEC_RD(4);
                else if (m == 3) RISVEC_RD(5); else if (m == 4) RISVEC_RD(6); else RISVEC_RD(7);
                asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(ah[m % 3]), "+v"(al[m % 3]));
            } else if (m + 1 < MT) {
                asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(ah[m % 3]), "+v"(al[m % 3]));
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah[m % 3]), "+v"(al[m % 3]));
            }
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m % 3], bf[0], acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[m % 3], bf[0], acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m % 3], bf[1], acc[m], 0, 0, 0);
        }
#undef RISVEC_RD
    };

    // The weight stream runs two groups ahead of the MFMAs (a chunk of MFMAs is shorter than an L2 round trip):
    // slot g % 3 is read while g+1 and g+2 are in flight; counted waits leave the newest group outstanding across
    // the barrier.  Raw s_barrier: __syncthreads() would drain the LDS-direct loads (vmcnt(0)).
    const uint4* hsrc = A.WhF + (size_t)v * (4 * MT * kWave);
    uint4* s_heads = s_ring + (NG % kRing) * kGroupVec;       // free from the last group on
    f32x16_t d = layer1(s_ring + (kRing - 1) * kGroupVec);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // every wavefront has read the group-0 fc1 operand ...
    __builtin_amdgcn_s_barrier();                                // ... before group 2 is staged over it
    half8_t b0[2], b1[2];
    for (int g = 0; g < NG; ++g) {
        const uint4* slot = s_ring + (g % kRing) * kGroupVec;
        make_b(d, 0, b0);
        if (g + 2 < NG) {
            stage(g + 2);
        } else if (g + 2 == NG || NG == 1) {                  // nothing left to stage: the head weight rides in on the
#pragma unroll                                                //  free slot, one group before it is needed
            for (int q = 0; q < kHeadStage; ++q)
                __builtin_amdgcn_global_load_lds((const gvoid_t*)(hsrc + q * kMlpBlock + tid), (lvoid_t*)(s_heads + q * kMlpBlock + tid), 16, 0, 0);
        }
        mfma_chunk(slot + 8 * kWave, b0);
        make_b(d, 1, b1);                                     // beside chunk 0's MFMAs
        if (g + 1 < NG) d = layer1(slot);                     // the next group's fc1 product: its operand came with this slot
        mfma_chunk(slot + 8 * kWave + kChunkVec, b1);
        if (g + 2 < NG) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(kStage) : "memory");
        else if (g + 2 == NG) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(kHeadStage) : "memory");   // last group landed
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    // ---- fc2 bias + LayerNorm + ReLU, in registers: this lane owns features 32m + (q & 3) + 8 (q >> 2) + 4h of its env
    // (whole-tile vector arithmetic: the compiler turns it into packed two-lane float32 instructions)
    const float inv_f2 = 1.0f / (float)F2, unscale = A.gscale[v];
    {
        auto tile_of = [&](const float* tab, int m) {            // 16 per-feature parameters in C/D register order
            f32x16_t t;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 q4 = *reinterpret_cast<const float4*>(tab + 32 * m + 8 * g + 4 * h);
                t[4 * g] = q4.x; t[4 * g + 1] = q4.y; t[4 * g + 2] = q4.z; t[4 * g + 3] = q4.w;
            }
            return t;
        };
        f32x16_t vs;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            acc[m] = acc[m] * unscale + tile_of(s_p2, m);
            vs = m == 0 ? acc[0] : vs + acc[m];
        }
        float s = 0.0f;
#pragma unroll
        for (int q = 0; q < 16; ++q) s += vs[q];
        s += __shfl_xor(s, 32, kWave);
        const float mean = s * inv_f2;
        f32x16_t v2;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            acc[m] = acc[m] - mean;                               // centred once, reused by the normalisation
            v2 = m == 0 ? acc[0] * acc[0] : v2 + acc[m] * acc[m];
        }
        float s2 = 0.0f;
#pragma unroll
        for (int q = 0; q < 16; ++q) s2 += v2[q];
        s2 += __shfl_xor(s2, 32, kWave);
        const float rs = rsqrtf(s2 * inv_f2 + kLnEps);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            acc[m] = (acc[m] * rs) * tile_of(s_p2 + F2, m) + tile_of(s_p2 + 2 * F2, m);
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[m][q] = fmaxf(acc[m][q], 0.0f);
        }
    }

    // ---- heads on the matrix cores: D[head][env] = Wh^T . y with the accumulator registers themselves as the B
    // operand (registers 8u .. 8u+7 of tile m are k-step (m, u); the head weight was laid out in that k order),
    // split hi + lo like the fc2 product
    f32x16_t hacc;
#pragma unroll
    for (int q = 0; q < 16; ++q) hacc[q] = 0.0f;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const half8_t wh = __builtin_bit_cast(half8_t, s_heads[((m * 2 + u) * 2 + 0) * kWave + lane]);
            const half8_t wl = __builtin_bit_cast(half8_t, s_heads[((m * 2 + u) * 2 + 1) * kWave + lane]);
            f32x8_t y;
#pragma unroll
            for (int j = 0; j < 8; ++j) y[j] = acc[m][8 * u + j];
            half8_t yh, yl;
            split16(y, yh, yl);
            hacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, yh, hacc, 0, 0, 0);
            hacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, yh, hacc, 0, 0, 0);
            hacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, yl, hacc, 0, 0, 0);
        }
    // head rows (q & 3) + 8 (q >> 2) + 4h: three groups of four consecutive heads per lane
    const float hs = A.hscale[v];
    float outv[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) outv[q] = fmaf(hacc[q], hs, s_bh[(q & 3) + 8 * (q >> 2) + 4 * h]);
    if (e < A.E) {
        float* o = A.heads + ((size_t)v * A.E + e) * H;
#pragma unroll
        for (int gq = 0; gq < 3; ++gq) {
            const int base = 8 * gq + 4 * h;
            if ((H & 3) == 0 && base + 3 < H) {
                *reinterpret_cast<float4*>(o + base) = make_float4(outv[4 * gq], outv[4 * gq + 1], outv[4 * gq + 2], outv[4 * gq + 3]);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (base + k < H) o[base + k] = outv[4 * gq + k];
            }
        }
    }
}

template <int MT>
hipError_t launch_mlp(const MlpArgs& a, hipStream_t st) {
    const int F2 = 32 * MT;
    const size_t lds = (size_t)kRing * (8 * kWave + 4 * MT * kWave) * sizeof(uint4) + ((size_t)3 * F2 + 32 + kIn1 * kIn1) * sizeof(float);
    auto kern = k_policy_mlp<MT>;
    if (lds > 64 * 1024) {
        hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (err != hipSuccess) return err;
    }
    const int rows = (kMlpBlock / kWave) * 32;
    hipLaunchKernelGGL(kern, dim3((unsigned)((a.E + rows - 1) / rows), (unsigned)a.V), dim3(kMlpBlock), lds, st, a);
    return hipGetLastError();
}

}  // namespace

// Instantiated for the reference's shapes: input_dims <= 5, fc2 = 256 or 128, up to 24 heads (4 + V, V <= 20).
// Anything else reports hipErrorInvalidValue and the caller uses the three-launch path.
bool policy_mlp_supported(int IN, int F1, int F2, int H) {
    return IN >= 1 && IN <= 5 && F1 >= 32 && F1 % 32 == 0 && F1 <= 1024 && (F2 == 256 || F2 == 128) && H >= 1 && H <= 24;
}

hipError_t launch_policy_mlp(int E, int V, int IN, int F1, int F2, int H, const float* obs, const float* G, const void* W1F,
                             const void* W2f, const float* gscale, const float* b2, const float* ln2w, const float* ln2b,
                             const void* WhF, const float* hscale, const float* bh, float* heads, hipStream_t st) {
    if (!policy_mlp_supported(IN, F1, F2, H)) return hipErrorInvalidValue;
    MlpArgs a{E, V, IN, F1, H, obs, G, static_cast<const uint4*>(W1F), static_cast<const uint4*>(W2f), gscale, b2, ln2w, ln2b,
              static_cast<const uint4*>(WhF), hscale, bh, heads};
    return F2 == 256 ? launch_mlp<8>(a, st) : launch_mlp<4>(a, st);
}

}  // namespace risvec
