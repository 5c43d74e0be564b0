// K5: one BCD (block-coordinate-descent) sweep over the discrete RIS phases.
//
// Reference: optimize_phase_shift ENV:208-220 with the objective of
// optimize_compute_objective_function ENV:222-231.  The reference objective sums
// the WHOLE [V,M] product (ENV:226 has no vehicle index), so it equals
//     Kc * | sum_m theta_m c_m |^2,   c_m = (sum_v h_r[v,m]) b[m],   Kc > 0,
// and for element m the candidate score is Kc |rest + cand_k c_m|^2 with
// rest = S - theta_m c_m.  Since
//     |rest + cand c|^2 = |rest|^2 + |c|^2 + 2 Re(cand * conj(rest) c),
// the arg-max over the 2^b candidates is the arg-max of Re(cand_k q), q = conj(rest) c_m:
// two FMAs per candidate.  That turns the reference's O(M^2 2^b V^2) sweep into
// O(V M + M 2^b).  `best < x` from best = 0 (ENV:210-218) = "first index wins ties, and the
// winner must score > 0", i.e. the new S = rest + cand c must be non-zero; otherwise the
// element becomes the integer 0 (ENV:211, 220).
//
// The sweep is a chain of M dependent discrete decisions, so it runs in float64 (inputs h_r,
// theta, b are the float32 tensors): a float32 sweep would flip near-tied decisions and
// drift away from the reference's theta.
//
// k_bcd_lane (even M): a 256-thread block stages c[] (f64) of `epb` envs in LDS - phase 1
// streams h_r with 16-byte loads, two lanes per element pair splitting the vehicle rows - then
// ONE LANE PER ENV walks the chain with no cross-lane traffic (theta prefetched from global
// 16 elements ahead, the new theta parked in the dead c slots), and the block writes theta
// back coalesced.  The chain is latency-bound, so throughput = envs in flight per CU, which
// LDS capacity bounds (16 B per element per env): two blocks per CU alternate streaming
// and sweeping.
// k_bcd_group (any M): the earlier form, 2^b lanes per env with a butterfly arg-max.
#include <cstdlib>

#include "risvec_step.hpp"

namespace risvec {

constexpr int kBcdLdsBudget = 79 * 1024;        // dynamic LDS per block; with the static candidate
                                                // table two blocks fit a CU's 160 KiB

__host__ __device__ constexpr size_t bcd_env_bytes(int M) {
    return (size_t)(M + 1) * sizeof(double2);
}

// arg-max_k Re(cand_k q) over the NC = 2^b unit phasors cand_k = exp(j 2 pi k / NC); the
// first index wins exact ties.  Returns k and writes the phasor.  Branch-free for NC = 8.
template <int NC>
__device__ __forceinline__ int pick_candidate(double qr, double qi, const double2* __restrict__ cand,
                                              double& nr, double& ni) {
    if constexpr (NC == 8) {
        // Re(cand_k q) = cand_k . w with w = (qr, -qi): the winner is the multiple of 45 deg
        // nearest to the direction of w -> an octant test instead of eight dot products.
        // (Boundaries sit at tan(22.5 deg), irrational: exact ties only at w = 0 -> k = 0.)
        const double fa = fabs(qr), fb = fabs(qi);
        const double t = 0.41421356237309503;               // tan(pi/8)
        const double r = 0.70710678118654757;               // cos(pi/4), as numpy rounds it
        const bool ax = fb <= t * fa;                        // along +-x (also w = 0 -> k = 0)
        const bool ay = !ax && (fa <= t * fb);               // along +-y
        const bool xn = qr < 0.0, yn = qi > 0.0;             // signs of w = (qr, -qi)
        const double mx = ax ? 1.0 : r, my = ay ? 1.0 : r;
        nr = ay ? 0.0 : (xn ? -mx : mx);
        ni = ax ? 0.0 : (yn ? -my : my);
        const int kd = yn ? (xn ? 5 : 7) : (xn ? 3 : 1);
        return ax ? (xn ? 4 : 0) : (ay ? (yn ? 6 : 2) : kd);
    } else {
        double best = cand[0].x * qr - cand[0].y * qi;
        int kb = 0;
        for (int k = 1; k < NC; ++k) {
            const double dk = cand[k].x * qr - cand[k].y * qi;
            if (dk > best) { best = dk; kb = k; }            // strict: first index wins ties
        }
        nr = cand[kb].x; ni = cand[kb].y;
        return kb;
    }
}

constexpr int kBcdRows = 8;     // h_r rows a lane keeps in flight per element pair

constexpr int kBcdTheta = 8;    // float4 (= 2 theta values) a sweep lane prefetches per block of 16 steps

template <int NC>
__global__ void __launch_bounds__(kBlock)
k_bcd_lane(Dims d, int epb, const float* __restrict__ h_r, float* __restrict__ theta,
           const float* __restrict__ b, int32_t* __restrict__ idx_out, int dbg) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int M = d.M, V = d.V, NPAIR = M >> 1;
    const int cstride = M + 1;                   // +16 B: lanes (envs) land on different banks
    double2* s_c = reinterpret_cast<double2*>(smem);                      // [epb][cstride]
    __shared__ double2 s_cand[NC];
    const int tid = threadIdx.x;
    const int e_blk = blockIdx.x * epb;
    const int n_env = min(epb, d.E - e_blk);
    const float4* __restrict__ h4 = reinterpret_cast<const float4*>(h_r);
    const float4* __restrict__ t4 = reinterpret_cast<const float4*>(theta);
    const float4* __restrict__ b4 = reinterpret_cast<const float4*>(b);

    if (tid < NC) {                              // candidate k: exp(j 2 pi k / NC)  (ENV:169, 213)
        double s, c;
        sincospi(2.0 * (double)tid / (double)NC, &s, &c);
        s_cand[tid] = make_double2(c, s);
    }

    // ---- phase 1: c[i][m] = (sum_v h_r[e,v,m]) b[m].  The (env, element-pair) slots of the
    // block are flattened over the threads, two lanes per slot: lane 2s takes the even vehicle
    // rows, lane 2s+1 the odd ones (their halves meet through one DPP exchange, then each
    // writes one element of the pair); each lane keeps up to kBcdRows 16-byte loads in flight.
    const int vh = tid & 1;
    const int n_slot = (dbg & 2) ? 0 : n_env * NPAIR;
    for (int s0 = 0; s0 < n_slot; s0 += kBlock / 2) {
        const int slot = s0 + (tid >> 1);
        const bool in = slot < n_slot;
        const int sl = in ? slot : 0;
        const int i = sl / NPAIR, p = sl - i * NPAIR;
        const long long e = e_blk + i;
        const float4* __restrict__ he = h4 + (e * V) * NPAIR + p;
        double s0r = 0.0, s0i = 0.0, s1r = 0.0, s1i = 0.0;
        for (int v0 = 0; v0 < V; v0 += 2 * kBcdRows) {
            float4 hb[kBcdRows];
#pragma unroll
            for (int k = 0; k < kBcdRows; ++k) {
                // unconditional load from a clamped (always valid) row: a predicated load would
                // become a branch with its own vmcnt(0) and serialise the eight requests
                const int v = v0 + vh + 2 * k;
                hb[k] = he[(long long)(v < V ? v : V - 1) * NPAIR];
            }
#pragma unroll
            for (int k = 0; k < kBcdRows; ++k) {
                const bool ok = in && (v0 + vh + 2 * k) < V;
                s0r += ok ? (double)hb[k].x : 0.0; s0i += ok ? (double)hb[k].y : 0.0;
                s1r += ok ? (double)hb[k].z : 0.0; s1i += ok ? (double)hb[k].w : 0.0;
            }
        }
        s0r += xchg<1>(s0r); s0i += xchg<1>(s0i);
        s1r += xchg<1>(s1r); s1i += xchg<1>(s1i);
        if (in) {
            const float4 bb = b4[p];
            const double sr = vh ? s1r : s0r, si = vh ? s1i : s0i;
            const double br = vh ? bb.z : bb.x, bi = vh ? bb.w : bb.y;
            s_c[(size_t)i * cstride + 2 * p + vh] = make_double2(sr * br - si * bi, sr * bi + si * br);
        }
    }
    __syncthreads();

    // ---- phases 2+3: one lane per env.  theta is read straight from global memory, a block
    // of 2*kBcdTheta elements ahead (the chain below takes microseconds per block); the new
    // theta is parked in the LDS slot of the element it replaces (c[m] is dead after step m).
    if (tid < n_env && !(dbg & 1)) {
        double2* c = s_c + (size_t)tid * cstride;
        const long long e = e_blk + tid;
        const float4* __restrict__ tg = t4 + e * NPAIR;
        const int n_blk = (NPAIR + kBcdTheta - 1) / kBcdTheta;
        float4 tcur[kBcdTheta], tnxt[kBcdTheta];

        // phase 2: S = sum_m theta_m c_m, fixed order (deterministic), two chains
        double Sr = 0.0, Si = 0.0, Tr = 0.0, Ti = 0.0;
#pragma unroll
        for (int j = 0; j < kBcdTheta; ++j) tcur[j] = tg[j < NPAIR ? j : NPAIR - 1];
        for (int kb = 0; kb < n_blk; ++kb) {
#pragma unroll
            for (int j = 0; j < kBcdTheta; ++j) {
                const int q = (kb + 1) * kBcdTheta + j;
                tnxt[j] = tg[q < NPAIR ? q : NPAIR - 1];
            }
#pragma unroll
            for (int j = 0; j < kBcdTheta; ++j) {
                const int q = kb * kBcdTheta + j;
                if (q < NPAIR) {
                    const double2 c0 = c[2 * q], c1 = c[2 * q + 1];
                    Sr += (double)tcur[j].x * c0.x - (double)tcur[j].y * c0.y;
                    Si += (double)tcur[j].x * c0.y + (double)tcur[j].y * c0.x;
                    Tr += (double)tcur[j].z * c1.x - (double)tcur[j].w * c1.y;
                    Ti += (double)tcur[j].z * c1.y + (double)tcur[j].w * c1.x;
                }
            }
#pragma unroll
            for (int j = 0; j < kBcdTheta; ++j) tcur[j] = tnxt[j];
        }
        Sr += Tr; Si += Ti;

        // phase 3: the chain
        auto chain_step = [&](int m, double tr, double ti) {
            const double2 cm = c[m];
            const double rr = Sr - (tr * cm.x - ti * cm.y);
            const double ri = Si - (tr * cm.y + ti * cm.x);
            const double qr = rr * cm.x + ri * cm.y;            // q = conj(rest) * c_m
            const double qi = rr * cm.y - ri * cm.x;
            double nr, ni;
            const int kb = pick_candidate<NC>(qr, qi, s_cand, nr, ni);
            const double nSr = rr + (nr * cm.x - ni * cm.y);
            const double nSi = ri + (nr * cm.y + ni * cm.x);
            const bool none = nSr == 0.0 && nSi == 0.0;         // no candidate scores above 0
            Sr = none ? rr : nSr;
            Si = none ? ri : nSi;
            *reinterpret_cast<float2*>(&c[m]) = make_float2(none ? 0.f : (float)nr, none ? 0.f : (float)ni);
            if (idx_out) idx_out[e * M + m] = none ? -1 : kb;
        };
#pragma unroll
        for (int j = 0; j < kBcdTheta; ++j) tcur[j] = tg[j < NPAIR ? j : NPAIR - 1];
        for (int kb = 0; kb < n_blk; ++kb) {
#pragma unroll
            for (int j = 0; j < kBcdTheta; ++j) {
                const int q = (kb + 1) * kBcdTheta + j;
                tnxt[j] = tg[q < NPAIR ? q : NPAIR - 1];
            }
#pragma unroll
            for (int j = 0; j < kBcdTheta; ++j) {
                const int q = kb * kBcdTheta + j;
                if (q < NPAIR) {
                    chain_step(2 * q, (double)tcur[j].x, (double)tcur[j].y);
                    chain_step(2 * q + 1, (double)tcur[j].z, (double)tcur[j].w);
                }
            }
#pragma unroll
            for (int j = 0; j < kBcdTheta; ++j) tcur[j] = tnxt[j];
        }
    }
    __syncthreads();

    // ---- write theta back, coalesced (it sits in the first 8 bytes of each c slot)
    float2* __restrict__ th_out = reinterpret_cast<float2*>(theta);
    for (int t = tid; t < n_env * M; t += kBlock) {
        const int i = t / M, m = t - i * M;
        th_out[(long long)(e_blk + i) * M + m] = *reinterpret_cast<const float2*>(&s_c[(size_t)i * cstride + m]);
    }
}

// ---------------------------------------------------------------------------
// generic form (any M): 2^b lanes per env, butterfly arg-max
// ---------------------------------------------------------------------------
template <int NC>
__global__ void __launch_bounds__(kBlock)
k_bcd_group(Dims d, int epb, const float* __restrict__ h_r, float* __restrict__ theta,
            const float* __restrict__ b, int32_t* __restrict__ idx_out) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int M = d.M, V = d.V;
    double2* s_c = reinterpret_cast<double2*>(smem);                 // [epb][M]
    float2* s_th = reinterpret_cast<float2*>(s_c + (size_t)epb * M); // [epb][M]
    const int e_blk = blockIdx.x * epb;
    const int n_env = min(epb, d.E - e_blk);

    for (int t = threadIdx.x; t < n_env * M; t += kBlock) {
        const int i = t / M, m = t % M;
        const long long e = e_blk + i;
        const float2* col = reinterpret_cast<const float2*>(h_r) + e * V * M + m;
        double sr = 0.0, si = 0.0;
        for (int v = 0; v < V; ++v) {
            const float2 h = col[(long long)v * M];
            sr += (double)h.x;
            si += (double)h.y;
        }
        const float2 bb = reinterpret_cast<const float2*>(b)[m];
        s_c[t] = make_double2(sr * bb.x - si * bb.y, sr * bb.y + si * bb.x);
        s_th[t] = reinterpret_cast<const float2*>(theta)[e * M + m];
    }
    __syncthreads();

    const int grp = threadIdx.x / NC, k = threadIdx.x % NC;
    const bool has_env = grp < n_env;          // uniform per group; NC | 64 so shuffles stay in-group
    const int gi = has_env ? grp : 0;
    const double2* c = s_c + (size_t)gi * M;
    const float2* th = s_th + (size_t)gi * M;
    const long long e = e_blk + gi;

    double Sr = 0.0, Si = 0.0;
    if (has_env) {
        for (int m = k; m < M; m += NC) {
            const double2 cm = c[m];
            const double tr = th[m].x, ti = th[m].y;
            Sr += tr * cm.x - ti * cm.y;
            Si += tr * cm.y + ti * cm.x;
        }
    }
    Sr = group_sum<NC>(Sr);
    Si = group_sum<NC>(Si);

    double cs, cc;
    sincospi(2.0 * (double)k / (double)NC, &cs, &cc);
    const int lane = threadIdx.x & (kWave - 1);
    const int base = lane - k;

    for (int m = 0; m < M; ++m) {
        const double2 cm = c[m];
        const double tr = th[m].x, ti = th[m].y;
        const double rr = Sr - (tr * cm.x - ti * cm.y);
        const double ri = Si - (tr * cm.y + ti * cm.x);
        double x = cc * (rr * cm.x + ri * cm.y) - cs * (rr * cm.y - ri * cm.x);   // Re(cand conj(rest) c)
        int kb = k;
#pragma unroll
        for (int o = NC / 2; o > 0; o >>= 1) {
            const double xo = __shfl_xor(x, o, kWave);
            const int ko = __shfl_xor(kb, o, kWave);
            if (xo > x || (xo == x && ko < kb)) { x = xo; kb = ko; }
        }
        double nr = __shfl(cc, base + kb, kWave), ni = __shfl(cs, base + kb, kWave);
        double nSr = rr + (nr * cm.x - ni * cm.y);
        double nSi = ri + (nr * cm.y + ni * cm.x);
        if (nSr == 0.0 && nSi == 0.0) { nr = 0.0; ni = 0.0; kb = -1; nSr = rr; nSi = ri; }
        Sr = nSr; Si = nSi;
        if (has_env && k == 0) {
            reinterpret_cast<float2*>(theta)[e * M + m] = make_float2((float)nr, (float)ni);
            if (idx_out) idx_out[e * M + m] = kb;
        }
    }
}

template <int NC>
static hipError_t launch_bcd_nc(const RisVecState& s, int32_t* idx_out, hipStream_t st) {
    const int M = s.n_ris;
    if ((M & 1) == 0) {
        const size_t per_env = bcd_env_bytes(M);
        int epb = (int)(kBcdLdsBudget / per_env);
        if (epb > kWave) epb = kWave;                       // one sweep lane per env, one wave of them
        if (epb >= 1) {
            static bool attr_set = false;                   // per instantiation
            if (!attr_set) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bcd_lane<NC>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, kBcdLdsBudget);
                if (e != hipSuccess) return e;
                attr_set = true;
            }
            // spread envs evenly over the blocks actually needed
            const unsigned grid = (unsigned)((s.n_envs + epb - 1) / epb);
            static const int dbg = [] { const char* e = std::getenv("RISVEC_BCD_DBG"); return e ? std::atoi(e) : 0; }();
            hipLaunchKernelGGL((k_bcd_lane<NC>), dim3(grid), dim3(kBlock), epb * per_env, st, dims_of(s), epb,
                               s.h_r, s.theta, s.b, idx_out, dbg);
            return hipGetLastError();
        }
    }
    const size_t per_env = (size_t)M * (sizeof(double2) + sizeof(float2));
    int epb = kBlock / NC;
    const int cap = (int)((64u * 1024u) / per_env);
    if (epb > cap) epb = cap;
    if (epb < 1) return hipErrorInvalidValue;
    const unsigned grid = (unsigned)((s.n_envs + epb - 1) / epb);
    hipLaunchKernelGGL((k_bcd_group<NC>), dim3(grid), dim3(kBlock), epb * per_env, st, dims_of(s), epb,
                       s.h_r, s.theta, s.b, idx_out);
    return hipGetLastError();
}

hipError_t launch_bcd(const RisVecState& s, const RisVecParams&, int32_t* idx_out, hipStream_t st) {
    switch (s.control_bit) {
        case 0: return launch_bcd_nc<1>(s, idx_out, st);
        case 1: return launch_bcd_nc<2>(s, idx_out, st);
        case 2: return launch_bcd_nc<4>(s, idx_out, st);
        case 3: return launch_bcd_nc<8>(s, idx_out, st);
        case 4: return launch_bcd_nc<16>(s, idx_out, st);
        case 5: return launch_bcd_nc<32>(s, idx_out, st);
        case 6: return launch_bcd_nc<64>(s, idx_out, st);
        default: return hipErrorInvalidValue;
    }
}

// BCD + gains + step: the sweep, then the fused gain+step kernel on the same stream.
hipError_t launch_step_fused_bcd(const RisVecState& s, const RisVecParams& p, const float* action,
                                 const int32_t* partner, const int32_t* n_groups,
                                 const int32_t* arrivals, uint64_t seed, uint32_t counter,
                                 uint32_t flags, hipStream_t st) {
    hipError_t err = launch_bcd(s, p, nullptr, st);
    if (err != hipSuccess) return err;
    return launch_step(s, p, action, partner, n_groups, arrivals, seed, counter, flags, true, st);
}

}  // namespace risvec
