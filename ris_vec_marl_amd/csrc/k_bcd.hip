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
// two FMAs per candidate, or for 2^b = 8 an octant test on q.  That turns the reference's
// O(M^2 2^b V^2) sweep into O(V M + M 2^b).  `best < x` from best = 0 (ENV:210-218) =
// "first index wins ties, and the winner must score > 0", i.e. the new S = rest + cand c
// must be non-zero; otherwise the element becomes the integer 0 (ENV:211, 220).
//
// The sweep is a chain of M dependent discrete decisions, so it runs in float64 (inputs h_r,
// theta, b are the float32 tensors): a float32 sweep would flip near-tied decisions and
// drift away from the reference's theta.
//
// Two kernels:
//   k_colsum      c_col[e,m] = (sum_v h_r[e,v,m]) b[m] in float64: one HBM pass over h_r, two
//                 lanes per element pair splitting the vehicle rows, eight 16-byte loads in
//                 flight per lane.  c is pure geometry (it changes only when h_r does), so it is
//                 cached in HBM like the path-loss factor: risvec_geometry rebuilds it, and a
//                 BCD call may reuse it.
//   k_bcd_sweep   ONE LANE PER ENV walks the chain; c streams from a lane-major cache, theta tiles
//                 are transposed through LDS (see the kernel).  An earlier form that staged c for
//                 whole envs in LDS was capped by LDS capacity at 38 envs per CU in flight.
#include <cstdlib>
#include <type_traits>

#include "risvec_pipe.hpp"

namespace risvec {

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

// theta lives in HBM as complex64, but the reference keeps it as complex128 holding the EXACT
// candidate phasors (ENV:213-220).  When a stored element is the float32 image of a candidate,
// the sweep uses that candidate in float64 - so S = sum theta.c stays what the reference's is,
// and the sum a sweep leaves behind is exactly the sum of what it stored.  Anything else
// (arbitrary phases from get_next_phase, the all-zero start) is used as stored.
template <int NC>
__device__ __forceinline__ void snap_theta(float2 t, const double2* __restrict__ cand, double& tr, double& ti) {
    if constexpr (NC == 8) {
        // only the four diagonal phasors are inexact in float32 (+-1, +-j and 0 are exact)
        const float r32 = 0.70710677f;
        const double r64 = 0.70710678118654757;
        const bool diag = fabsf(t.x) == r32 && fabsf(t.y) == r32;
        tr = diag ? copysign(r64, (double)t.x) : (double)t.x;
        ti = diag ? copysign(r64, (double)t.y) : (double)t.y;
    } else {
        double cr, ci;
        pick_candidate<NC>((double)t.x, -(double)t.y, cand, cr, ci);  // nearest candidate: max Re(conj(cand) t)
        const bool is_cand = (float)cr == t.x && (float)ci == t.y;
        tr = is_cand ? cr : (double)t.x;
        ti = is_cand ? ci : (double)t.y;
    }
}

// ---------------------------------------------------------------------------
// k_colsum
// ---------------------------------------------------------------------------
constexpr int kColRows = 8;     // h_r rows a lane keeps in flight

// VEC = 2: a slot is an element pair read with float4 (M even); lanes 2s / 2s+1 take the even /
// odd vehicle rows, meet through one DPP exchange, then each writes one element of the pair.
// VEC = 1: a slot is one element read with float2 (odd M), same row split.
template <int VEC>
__global__ void __launch_bounds__(kBlock)
k_colsum(Dims d, const float* __restrict__ h_r, const float* __restrict__ b, double* __restrict__ c_col) {
    const int M = d.M, V = d.V;
    const int spe = VEC == 2 ? (M >> 1) : M;               // slots per env
    const long long n_slot = (long long)d.E * spe;
    const long long slot = (long long)blockIdx.x * (kBlock / 2) + (threadIdx.x >> 1);
    const int vh = threadIdx.x & 1;
    const bool in = slot < n_slot;
    const long long sl = in ? slot : 0;
    const long long e = sl / spe;
    const int p = (int)(sl - e * spe);
    double s0r = 0.0, s0i = 0.0, s1r = 0.0, s1i = 0.0;
    if constexpr (VEC == 2) {
        const float4* __restrict__ he = reinterpret_cast<const float4*>(h_r) + (e * V) * spe + p;
        for (int v0 = 0; v0 < V; v0 += 2 * kColRows) {
            float4 hb[kColRows];
#pragma unroll
            for (int k = 0; k < kColRows; ++k) {
                // unconditional load from a clamped (always valid) row: a predicated load would
                // become a branch with its own vmcnt(0) and serialise the eight requests
                const int v = v0 + vh + 2 * k;
                hb[k] = he[(long long)(v < V ? v : V - 1) * spe];
            }
#pragma unroll
            for (int k = 0; k < kColRows; ++k) {
                const bool ok = (v0 + vh + 2 * k) < V;
                s0r += ok ? (double)hb[k].x : 0.0; s0i += ok ? (double)hb[k].y : 0.0;
                s1r += ok ? (double)hb[k].z : 0.0; s1i += ok ? (double)hb[k].w : 0.0;
            }
        }
    } else {
        const float2* __restrict__ he = reinterpret_cast<const float2*>(h_r) + (e * V) * spe + p;
        for (int v0 = 0; v0 < V; v0 += 2 * kColRows) {
            float2 hb[kColRows];
#pragma unroll
            for (int k = 0; k < kColRows; ++k) {
                const int v = v0 + vh + 2 * k;
                hb[k] = he[(long long)(v < V ? v : V - 1) * spe];
            }
#pragma unroll
            for (int k = 0; k < kColRows; ++k) {
                const bool ok = (v0 + vh + 2 * k) < V;
                s0r += ok ? (double)hb[k].x : 0.0; s0i += ok ? (double)hb[k].y : 0.0;
            }
        }
    }
    s0r += xchg<1>(s0r); s0i += xchg<1>(s0i);
    if constexpr (VEC == 2) { s1r += xchg<1>(s1r); s1i += xchg<1>(s1i); }
    if (!in) return;
    // c_col is private to the BCD kernels and stored LANE-MAJOR: element (e, m) lives at
    // ((e / 64) * M + m) * 64 + e % 64, so the sweep's 64 lanes (64 consecutive envs) read one
    // element each from 1 KiB of contiguous memory.  The price is paid here, once per geometry
    // refresh: these 16-byte stores are 1 KiB apart.
    double2* __restrict__ out = reinterpret_cast<double2*>(c_col) + ((e >> 6) * M) * 64 + (e & 63);
    if constexpr (VEC == 2) {
        const float4 bb = reinterpret_cast<const float4*>(b)[p];
        const double sr = vh ? s1r : s0r, si = vh ? s1i : s0i;
        const double br = vh ? bb.z : bb.x, bi = vh ? bb.w : bb.y;
        out[(long long)(2 * p + vh) * 64] = make_double2(sr * br - si * bi, sr * bi + si * br);
    } else if (vh == 0) {
        const float2 bb = reinterpret_cast<const float2*>(b)[p];
        out[(long long)p * 64] = make_double2(s0r * bb.x - s0i * bb.y, s0r * bb.y + s0i * bb.x);
    }
}

// k_colsum_slab (M a multiple of 16): the same sums with BOTH sides of the transpose coalesced.  A workgroup owns
// one slab of 64 envs x 16 elements: eight lanes read one 128-byte line of one vehicle row (two elements per lane,
// every row of the env in flight at once), the float64 column sums go through a [16][64] LDS tile, and the
// workgroup writes sixteen 1-KiB runs of the lane-major cache.  k_colsum<2> above reads just as well but scatters
// its 16-byte stores 2 KiB apart (343 us at 32 768 x 16 x 256 = 44 % of the HBM peak, profiles/r01p_*).
constexpr int kSlabM = 16;                                  // elements per tile
constexpr int kSlabRow = kWave + 1;                         // LDS row stride (double2 units): staggers the banks

constexpr int kSlabThreads = 8 * kWave;                    // 8 lanes per (env, 128-byte line), 64 envs

template <int VU, bool NT>                                  // VU: vehicle rows a lane keeps in flight (VU >= V); NT: non-temporal loads
__global__ void __launch_bounds__(kSlabThreads)
k_colsum_slab(Dims d, const float* __restrict__ h_r, const float* __restrict__ b, double* __restrict__ c_col) {
    __shared__ double2 s_c[kSlabM * kSlabRow];              // 16 640 B
    const int M = d.M, V = d.V, NP = M >> 1;
    const int tiles = M / kSlabM;
    const int slab = blockIdx.x;                            // a workgroup walks the 16-element tiles of its slab in order:
                                                            // the rows it touches stay open / cached from tile to tile
    const int l8 = threadIdx.x & 7, el = threadIdx.x >> 3; // element pair inside the tile, env inside the slab
    long long e = (long long)slab * kWave + el;
    e = e < d.E ? e : d.E - 1;                              // tail slab: re-read the last env (never used by a live lane)
    const float4* __restrict__ he = reinterpret_cast<const float4*>(h_r) + e * V * NP + l8;
    const float4* __restrict__ b4 = reinterpret_cast<const float4*>(b) + l8;
    double2* __restrict__ out = reinterpret_cast<double2*>(c_col) + (long long)slab * M * kWave;
    struct Rows { float4 h[VU]; };
    auto fetch = [&](Rows& r, int tile) {
        const int tc = tile < tiles ? tile : tiles - 1;
#pragma unroll
        for (int k = 0; k < VU; ++k) {
            const float4* src = he + ((long long)(k < V ? k : V - 1) * NP + tc * (kSlabM / 2));
            if (NT) {
                typedef float v4f __attribute__((ext_vector_type(4)));
                const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(src));
                r.h[k] = make_float4(t.x, t.y, t.z, t.w);
            } else {
                r.h[k] = *src;
            }
        }
    };
    auto do_tile = [&](int tile, const Rows& cur, Rows& nxt) {
        fetch(nxt, tile + 1);                               // the next tile's rows are in flight during this tile's stores
        const float4 bb = b4[tile * (kSlabM / 2)];
        double s0r = 0.0, s0i = 0.0, s1r = 0.0, s1i = 0.0;
#pragma unroll
        for (int k = 0; k < VU; ++k) {
            const bool ok = k < V;
            s0r += ok ? (double)cur.h[k].x : 0.0; s0i += ok ? (double)cur.h[k].y : 0.0;
            s1r += ok ? (double)cur.h[k].z : 0.0; s1i += ok ? (double)cur.h[k].w : 0.0;
        }
        s_c[(2 * l8) * kSlabRow + el] = make_double2(s0r * bb.x - s0i * bb.y, s0r * bb.y + s0i * bb.x);
        s_c[(2 * l8 + 1) * kSlabRow + el] = make_double2(s1r * bb.z - s1i * bb.w, s1r * bb.w + s1i * bb.z);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < kSlabM * kWave / kSlabThreads; ++i) {
            const int idx = threadIdx.x + i * kSlabThreads; // m_local * 64 + env: consecutive lanes, consecutive envs
            out[(long long)tile * kSlabM * kWave + idx] = s_c[(idx >> 6) * kSlabRow + (idx & 63)];
        }
        __syncthreads();
    };
    Rows ra, rb;
    fetch(ra, 0);
    for (int tile = 0; tile < tiles; tile += 2) {
        do_tile(tile, ra, rb);
        if (tile + 1 < tiles) do_tile(tile + 1, rb, ra);
    }
}

// k_colsum_rows256 (M = 256, the C5 shape): the same sums with every READ a whole 2-KiB vehicle row.  The slab kernel
// above reads 128-byte pieces 2 KiB apart (one tile of 16 elements of every row of 64 envs) and reaches 4.5 TB/s on the
// 1.07 GB of h_r at 32 768 x 16 x 256 -- the fused step kernel, which walks each env's 32 KiB front to back, reads the
// same bytes at 6.5.  Here a wavefront does that walk: a row is 64 lanes x 32 bytes (four elements per lane), all rows
// of the env in flight at once, float64 sums in registers; a workgroup covers 16 consecutive envs (8 wavefronts x 2)
// and turns them into the lane-major cache through a [256][16] LDS tile, 256 bytes of every 1-KiB run.
constexpr int kRowsM = 256, kRowsEnvs = 16, kRowsThreads = 8 * kWave, kRowsPad = kRowsEnvs + 1;

template <int VU, bool NT>                                  // VU: vehicle rows a lane keeps in flight (VU >= V); NT: non-temporal loads
__global__ void __launch_bounds__(kRowsThreads)
k_colsum_rows256(Dims d, const float* __restrict__ h_r, const float* __restrict__ b, double* __restrict__ c_col) {
    __shared__ double2 s_c[kRowsM * kRowsPad];              // 69 632 B
    const int V = d.V, lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const long long e_base = (long long)blockIdx.x * kRowsEnvs;
    const float4* __restrict__ b4 = reinterpret_cast<const float4*>(b) + 2 * lane;
    const float4 bb0 = b4[0], bb1 = b4[1];                  // b of elements 4 lane .. 4 lane + 3
    auto ld = [](const float4* src) {
        if (NT) {
            typedef float v4f __attribute__((ext_vector_type(4)));
            const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(src));
            return make_float4(t.x, t.y, t.z, t.w);
        }
        return *src;
    };
#pragma unroll
    for (int q = 0; q < kRowsEnvs / 8; ++q) {
        const int el = wave * (kRowsEnvs / 8) + q;          // env inside the workgroup
        long long e = e_base + el;
        e = e < d.E ? e : d.E - 1;                          // tail: re-read the last env (never written)
        const float4* __restrict__ row = reinterpret_cast<const float4*>(h_r) + e * V * (kRowsM / 2) + 2 * lane;
        float4 h0[VU], h1[VU];
#pragma unroll
        for (int k = 0; k < VU; ++k) {
            const float4* src = row + (long long)(k < V ? k : V - 1) * (kRowsM / 2);
            h0[k] = ld(src);
            h1[k] = ld(src + 1);
        }
        double s0r = 0.0, s0i = 0.0, s1r = 0.0, s1i = 0.0, s2r = 0.0, s2i = 0.0, s3r = 0.0, s3i = 0.0;
#pragma unroll
        for (int k = 0; k < VU; ++k) {
            const bool ok = k < V;
            s0r += ok ? (double)h0[k].x : 0.0; s0i += ok ? (double)h0[k].y : 0.0;
            s1r += ok ? (double)h0[k].z : 0.0; s1i += ok ? (double)h0[k].w : 0.0;
            s2r += ok ? (double)h1[k].x : 0.0; s2i += ok ? (double)h1[k].y : 0.0;
            s3r += ok ? (double)h1[k].z : 0.0; s3i += ok ? (double)h1[k].w : 0.0;
        }
        double2* dst = s_c + (4 * lane) * kRowsPad + el;
        dst[0 * kRowsPad] = make_double2(s0r * bb0.x - s0i * bb0.y, s0r * bb0.y + s0i * bb0.x);
        dst[1 * kRowsPad] = make_double2(s1r * bb0.z - s1i * bb0.w, s1r * bb0.w + s1i * bb0.z);
        dst[2 * kRowsPad] = make_double2(s2r * bb1.x - s2i * bb1.y, s2r * bb1.y + s2i * bb1.x);
        dst[3 * kRowsPad] = make_double2(s3r * bb1.z - s3i * bb1.w, s3r * bb1.w + s3i * bb1.z);
    }
    __syncthreads();
    // element (e, m) of the lane-major cache: ((e / 64) * M + m) * 64 + e % 64; the 16 envs of a workgroup share e / 64
    double2* __restrict__ out = reinterpret_cast<double2*>(c_col) + (e_base >> 6) * kRowsM * kWave + (e_base & 63);
#pragma unroll
    for (int i = 0; i < kRowsM * kRowsEnvs / kRowsThreads; ++i) {
        const int idx = threadIdx.x + i * kRowsThreads, m = idx / kRowsEnvs, el = idx % kRowsEnvs;
        if (e_base + el < d.E) out[(long long)m * kWave + el] = s_c[m * kRowsPad + el];
    }
}

// ---------------------------------------------------------------------------
// k_bcd_sweep
// ---------------------------------------------------------------------------
// A wave owns 64 consecutive envs, one lane each.
//   c      is read straight from the lane-major c_col: one element per lane, 1 KiB contiguous
//          per instruction, 8 elements ahead of the chain.
//   theta  keeps its public row-major [E,M] layout (the gain kernels read it along m), where the
//          rows of neighbouring lanes are 8M bytes apart.  A per-lane access would touch 64
//          cache lines per instruction (measured: that, not the chain, bounded the kernel), so
//          theta tiles move COOPERATIVELY - 4 lanes read one env's 64 bytes, 16 envs per
//          instruction - and are transposed through LDS; the new theta goes back the same way.
// Measured on the way here (C5 shape, 32 768 envs): neither more waves per SIMD (8/16/32 envs per
// wave) nor longer tiles changed the time, i.e. the kernel is bound by how efficiently HBM
// serves its accesses, which is why c moved to a layout that streams.
constexpr int kSweepBlk = 8;                                // elements per tile
constexpr int kTRow = kSweepBlk + 2;                        // LDS row stride of the theta tile (elements)

// theta_idx: the candidate index of every theta element as the last 2^b = 8 sweep left it (8 = the integer 0 of
// ENV:211, 220), one byte per element, ROW-MAJOR [E][8 ceil(M / 8)] (round 3: the fused step kernel reads it along m
// when theta is kept by index).  Rows are padded to a multiple of 32 bytes: the pair sweep moves the indices of
// FOUR tiles per request.

template <int NC>
__global__ void __launch_bounds__(kWave)
k_bcd_sweep(Dims d, const double* __restrict__ c_col, float* __restrict__ theta,
            int32_t* __restrict__ idx_out, double* __restrict__ s_sum, int reuse_s,
            uint8_t* __restrict__ theta_idx) {
    __shared__ double2 s_cand[NC];
    __shared__ float2 s_t[kWave * kTRow];                   // 5 120 B
    const int M = d.M;
    const int lane = threadIdx.x;
    if (lane < NC) {                             // candidate k: exp(j 2 pi k / NC)  (ENV:169, 213)
        double s, c;
        sincospi(2.0 * (double)lane / (double)NC, &s, &c);
        s_cand[lane] = make_double2(c, s);
    }
    __syncthreads();
    const long long e0 = (long long)blockIdx.x * kWave;
    const long long e_last = d.E - 1;
    const bool live = e0 + lane < d.E;
    const long long e = live ? e0 + lane : e_last;
    // this wave's slab of c_col: [M][64] double2, lane-major (padded to whole slabs by the host)
    const double2* __restrict__ cg = reinterpret_cast<const double2*>(c_col) + (long long)blockIdx.x * M * kWave + lane;
    float2* __restrict__ tg = reinterpret_cast<float2*>(theta);
    const int n_blk = (M + kSweepBlk - 1) / kSweepBlk;
    const int sub4 = lane >> 2, q4 = lane & 3;              // theta tile roles: env sub-index, 16-byte piece

    double2 pc[kSweepBlk];
    float4 pt[4];
    auto fetch = [&](int kb) {                              // clamped indices keep every address valid
#pragma unroll
        for (int j = 0; j < kSweepBlk; ++j) pc[j] = cg[(long long)min(kb * kSweepBlk + j, M - 1) * kWave];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long long ee = min(e0 + i * 16 + sub4, e_last);
            const int m0 = kb * kSweepBlk + 2 * q4;
            const float2 a = tg[ee * M + min(m0, M - 1)], b2 = tg[ee * M + min(m0 + 1, M - 1)];
            pt[i] = make_float4(a.x, a.y, b2.x, b2.y);
        }
    };
    // c: registers as they are; theta: registers -> LDS tile -> each lane takes its own row.
    // Elements past M become c = theta = 0, for which a chain step leaves S unchanged.
    auto stage = [&](int kb, double2 (&cc)[kSweepBlk], float2 (&tc)[kSweepBlk]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            s_t[(i * 16 + sub4) * kTRow + 2 * q4] = make_float2(pt[i].x, pt[i].y);
            s_t[(i * 16 + sub4) * kTRow + 2 * q4 + 1] = make_float2(pt[i].z, pt[i].w);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < kSweepBlk; ++j) {
            const bool ok = kb * kSweepBlk + j < M;
            const float2 tv = s_t[lane * kTRow + j];
            cc[j] = make_double2(ok ? pc[j].x : 0.0, ok ? pc[j].y : 0.0);
            tc[j] = make_float2(ok ? tv.x : 0.f, ok ? tv.y : 0.f);
        }
        __builtin_amdgcn_wave_barrier();
    };

    double2 cc[kSweepBlk];
    float2 tc[kSweepBlk];

    // ---- pass 1: S = sum_m theta_m c_m in a fixed order (deterministic), two chains.  A sweep
    // ends knowing exactly this sum for the theta it wrote, and leaves it in s_sum[e]: the next
    // sweep of an unchanged (theta, c) starts from there and skips the pass.
    double Sr = 0.0, Si = 0.0, Tr = 0.0, Ti = 0.0;
    if (reuse_s) {
        const double2 s0 = reinterpret_cast<const double2*>(s_sum)[e];
        Sr = s0.x; Si = s0.y;
    } else {
        fetch(0);
        for (int kb = 0; kb < n_blk; ++kb) {
            stage(kb, cc, tc);
            fetch(kb + 1);                                  // clamped past the end: harmless re-read
#pragma unroll
            for (int j = 0; j < kSweepBlk; j += 2) {
                double ar, ai, br, bi;
                snap_theta<NC>(tc[j], s_cand, ar, ai);
                snap_theta<NC>(tc[j + 1], s_cand, br, bi);
                Sr += ar * cc[j].x - ai * cc[j].y;
                Si += ar * cc[j].y + ai * cc[j].x;
                Tr += br * cc[j + 1].x - bi * cc[j + 1].y;
                Ti += br * cc[j + 1].y + bi * cc[j + 1].x;
            }
        }
        Sr += Tr; Si += Ti;
    }

    // ---- pass 2: the chain.  The "no candidate scores above 0" case (new S exactly 0; ENV:211,
    // 220) is kept OFF the dependent chain: a tile first runs without it while OR-ing a flag, and
    // is re-run exactly from its saved start state in the (practically never taken) case that
    // some lane raised the flag.
    fetch(0);
    for (int kb = 0; kb < n_blk; ++kb) {
        stage(kb, cc, tc);
        fetch(kb + 1);
        float2 out[kSweepBlk];
        int ko[kSweepBlk];
        const double Sr0 = Sr, Si0 = Si;
        bool flagged = false;
#pragma unroll
        for (int j = 0; j < kSweepBlk; ++j) {
            const double2 cm = cc[j];
            double tr, ti;
            snap_theta<NC>(tc[j], s_cand, tr, ti);              // off the dependent chain
            const double rr = Sr - (tr * cm.x - ti * cm.y);
            const double ri = Si - (tr * cm.y + ti * cm.x);
            const double qr = rr * cm.x + ri * cm.y;            // q = conj(rest) * c_m
            const double qi = rr * cm.y - ri * cm.x;
            double nr, ni;
            ko[j] = pick_candidate<NC>(qr, qi, s_cand, nr, ni);
            Sr = rr + (nr * cm.x - ni * cm.y);
            Si = ri + (nr * cm.y + ni * cm.x);
            flagged |= (Sr == 0.0 && Si == 0.0);
            out[j] = make_float2((float)nr, (float)ni);
        }
        if (__any(flagged)) {
            Sr = Sr0; Si = Si0;
#pragma unroll
            for (int j = 0; j < kSweepBlk; ++j) {        // unrolled: cc/out/ko must stay in registers
                const double2 cm = cc[j];
                double tr, ti;
                snap_theta<NC>(tc[j], s_cand, tr, ti);
                const double rr = Sr - (tr * cm.x - ti * cm.y);
                const double ri = Si - (tr * cm.y + ti * cm.x);
                double nr, ni;
                const int kk = pick_candidate<NC>(rr * cm.x + ri * cm.y, rr * cm.y - ri * cm.x, s_cand, nr, ni);
                const double nSr = rr + (nr * cm.x - ni * cm.y);
                const double nSi = ri + (nr * cm.y + ni * cm.x);
                const bool none = nSr == 0.0 && nSi == 0.0;     // no candidate scores above 0
                Sr = none ? rr : nSr;
                Si = none ? ri : nSi;
                out[j] = make_float2(none ? 0.f : (float)nr, none ? 0.f : (float)ni);
                ko[j] = none ? -1 : kk;
            }
        }
        // new theta: lane rows -> LDS -> cooperative stores (16 envs x 64 contiguous bytes each)
#pragma unroll
        for (int j = 0; j < kSweepBlk; ++j) s_t[lane * kTRow + j] = out[j];
        if (idx_out && live) {
#pragma unroll
            for (int j = 0; j < kSweepBlk; ++j) {
                const int m = kb * kSweepBlk + j;
                if (m < M) idx_out[e * M + m] = ko[j];
            }
        }
        if (theta_idx && live) {                          // candidate index per element (8 = the integer 0) for the indexed sweep
            unsigned lo = 0, hi = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                lo |= (unsigned)(ko[j] < 0 ? 8 : ko[j]) << (8 * j);
                hi |= (unsigned)(ko[j + 4] < 0 ? 8 : ko[j + 4]) << (8 * j);
            }
            *reinterpret_cast<uint2*>(theta_idx + e * theta_idx_stride(M) + kb * 8) = make_uint2(lo, hi);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long long ee = e0 + i * 16 + sub4;
            const int m0 = kb * kSweepBlk + 2 * q4;
            const float2 a = s_t[(i * 16 + sub4) * kTRow + 2 * q4], b2 = s_t[(i * 16 + sub4) * kTRow + 2 * q4 + 1];
            if (ee < d.E) {
                if (m0 < M) tg[ee * M + m0] = a;
                if (m0 + 1 < M) tg[ee * M + m0 + 1] = b2;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (live && s_sum) reinterpret_cast<double2*>(s_sum)[e] = make_double2(Sr, Si);
}

// ---------------------------------------------------------------------------
// k_bcd_sweep8_pair: the sweep for 2^b = 8 when the candidate INDEX of every element is known (theta_idx, left by
// the previous sweep), TWO LANES PER ENV (round 3).
//
// The generic kernel above spends ~75 vector instructions per coordinate, most of them not on the dependent chain:
// reading theta back through an LDS transpose, re-snapping its float32 image to the exact float64 phasor, 64-bit
// selects for the octant winner.  With the indices known the old phasor is a table lookup hoisted out of the chain
// and theta is not read at all.  Round 2 ran that with one lane per env (k_bcd_sweep8_idx: 512 wavefronts for 32 768
// envs -- half of the chip's 1 024 SIMDs had none -- 348 cycles per coordinate, 243 of them in the chain); a lone
// wavefront issues one instruction per 4 (float32 / integer) or 8 (float64) cycles whatever its dependencies, so
// the chain time is (instructions per coordinate) x M.  Here the even lane of a pair owns the real part of the
// running sum S, the odd lane the imaginary part: each of the three complex products of a coordinate
// (rest = S - cand_old c, w = conj(rest) c, S' = rest + cand_new c) costs a lane 2 float64 instructions instead
// of 4, the halves meet through DPP quad permutes (lane ^ 1: plain VALU, no LDS), each lane evaluates ONE side
// of the octant test, and the tile epilogue (float32 image + index of the winner, theta / index stores) is split
// between the two lanes.  32 envs per wavefront: 1 024 wavefronts at BASELINE configs[4], one per SIMD.
//
// Per-lane operands that make both lanes run the SAME instructions (hb = lane & 1; c = (cx, cy)):
//   cys          = cy (hb = 0) / -cy (hb = 1)                      one XOR on the high word per coordinate
//   (p1, p2)     = (px, -py) / (py, -px)  of the old phasor         from a per-lane LDS table
//   rest_self    = S_self - p1 cx - p2 cys
//   w_self       = rest_self cx + rest_other cys                    = Re q / -Im q  (q = conj(rest) c): the winner
//                                                                     is the multiple of 45 deg nearest to (w_0, w_1)
//   along_self   = |w_other| <= tan(pi/8) |w_self|                  the winner lies on this lane's axis
//   n_self       = +-1 (along_self) | 0 (along_other only) | +-r (neither), sign of w_self
//   S'_self      = rest_self + n_self cx - n_other cys
// q = 0 (every candidate ties -> the first wins, ENV:210-218) and S' = 0 (no candidate scores above 0 -> the
// integer 0, ENV:211, 220) are detected off the chain (one v_min3 on high words per coordinate) and the tile is
// replayed exactly from its saved start state.  Decisions equal the generic kernel's wherever best and second-best
// candidate differ by more than float64 rounding (the products are fused differently).
// WTH = false ("theta by index", RISVEC_BCD_NO_THETA / RISVEC_STEP_THETA_BY_INDEX): the complex64 theta row is NOT
// written -- the indices are the state, k_theta_from_index materialises theta when somebody asks for it, and the
// fused step reads the indices.  Measured in the BASELINE configs[4] loop the sweep is bound by its memory traffic
// (134 MB of c_col in, 67 MB of theta out, 17 MB of indices: 44 us with either sweep kernel), not by the chain.
// ---------------------------------------------------------------------------
struct PairOld { double p1, p2; };                          // 16 bytes
struct PairOut { float cr, ci; int k, pad; };               // 16 bytes

__device__ __forceinline__ int dpp_x1(int x) { return __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true); }

template <bool PAD, bool WTH = true, bool STAMP = false>
__global__ void __launch_bounds__(kWave)
k_bcd_sweep8_pair(Dims d, const double* __restrict__ c_col, float* __restrict__ theta,
                  int32_t* __restrict__ idx_out, double* __restrict__ s_sum, int reuse_s,
                  uint8_t* __restrict__ theta_idx) {
    __shared__ PairOld s_old[2][16];                        // [hb][k]; k = 8 (and up): the integer 0
    __shared__ PairOut s_out[2][16];                        // [hb][code]; code = piece_self | piece_other << 2, piece = along << 1 | negative
    constexpr int HALF = kWave / 2;                         // envs per wavefront
    constexpr unsigned ONE_HI = 0x3FF00000u, R_HI = 0x3FE6A09Eu, R_LO = 0x667F3BCDu;     // 1.0, 0.70710678118654757
    const int M = d.M;
    const int lane = threadIdx.x, hb = lane & 1, ei = lane >> 1;
    if (lane < 32) {
        const double r = 0.70710678118654757;              // cos(pi/4) as numpy rounds it
        const double cr[9] = {1.0, r, 0.0, -r, -1.0, -r, 0.0, r, 0.0};
        const double ci[9] = {0.0, r, 1.0, r, 0.0, -r, -1.0, -r, 0.0};
        const int t_hb = lane >> 4, id = lane & 15;
        const int kc = id < 9 ? id : 8;
        s_old[t_hb][id] = t_hb ? PairOld{ci[kc], -cr[kc]} : PairOld{cr[kc], -ci[kc]};
        const int ps = id & 3, po = id >> 2;
        const int xp = t_hb ? po : ps, yp = t_hb ? ps : po;
        const bool ax = xp >> 1, sx = xp & 1, ay = yp >> 1, sy = yp & 1;   // sx / sy: w_0 / w_1 negative
        int k;
        if (ax && ay) k = 0;                                // q = 0: replayed; the first index
        else if (ax) k = sx ? 4 : 0;
        else if (ay) k = sy ? 6 : 2;
        else k = sx ? (sy ? 5 : 3) : (sy ? 7 : 1);
        s_out[t_hb][id] = PairOut{(float)cr[k], (float)ci[k], k, 0};
    }
    __syncthreads();
    const long long e0 = (long long)blockIdx.x * HALF;
    const bool live = e0 + ei < d.E;
    const long long e = live ? e0 + ei : d.E - 1;
    const long long slab = blockIdx.x >> 1;
    const int sl = (blockIdx.x & 1) * HALF + ei;            // this env's lane in its 64-env slab
    // c_col is addressed as (wave-uniform slab base + coordinate row, in SGPRs) + (this lane's 32-bit byte offset):
    // the loads take the scalar-base form and a tile's eight addresses cost two scalar adds, not eight 64-bit vector adds
    const char* __restrict__ cgs = reinterpret_cast<const char*>(c_col) + slab * M * (kWave * 16LL);
    const unsigned cgl = (unsigned)sl * 16u;
    const int n_blk = (M + kSweepBlk - 1) / kSweepBlk;
    uint8_t* __restrict__ ig = theta_idx + e * (long long)theta_idx_stride(M);     // this env's row of candidate indices
    constexpr int RING = 4;                                 // tile images in flight = tiles per index group
    const unsigned sgn = (unsigned)hb << 31;
    const PairOld* __restrict__ told = s_old[hb];
    const PairOut* __restrict__ tout = s_out[hb];

    struct Tile {
        double2 c[kSweepBlk];
        uint2 k;
    };
    auto fetch = [&](Tile& t, int kb, bool with_idx = false) __attribute__((always_inline)) {
        const int kbc = kb < n_blk ? kb : n_blk - 1;           // past the end: harmless re-read of the last tile
        const char* __restrict__ row = cgs + (long long)kbc * (kSweepBlk * kWave * 16);     // uniform
#pragma unroll
        for (int j = 0; j < kSweepBlk; ++j) {
            const int jj = PAD ? min(kbc * kSweepBlk + j, M - 1) - kbc * kSweepBlk : j;
            t.c[j] = *reinterpret_cast<const double2*>(row + jj * (kWave * 16) + cgl);
        }
        if (with_idx) t.k = *reinterpret_cast<const uint2*>(ig + kbc * 8);      // (the chain takes its indices by groups of 4 tiles)
    };
    auto old_of = [&](const Tile& t, int j) -> PairOld { return told[((j < 4 ? t.k.x : t.k.y) >> (8 * (j & 3))) & 15u]; };
    auto flip = [&](double cy) { return __hiloint2double(__double2hiint(cy) ^ (int)sgn, __double2loint(cy)); };

    // ---- pass 1 (only when the cached sum is not current): S_self = sum_m p1 cx + p2 cys
    double S = 0.0;
    if (reuse_s) {
        S = s_sum[e * 2 + hb];
    } else {
        double T = 0.0;
        Tile ta, tb;
        auto sum_tile = [&](int kb, const Tile& cur, Tile& nxt) {
            fetch(nxt, kb + 1, true);
#pragma unroll
            for (int j = 0; j < kSweepBlk; j += 2) {
                const bool ok0 = !PAD || kb * kSweepBlk + j < M, ok1 = !PAD || kb * kSweepBlk + j + 1 < M;
                const PairOld a = ok0 ? old_of(cur, j) : told[8], b2 = ok1 ? old_of(cur, j + 1) : told[8];
                S = fma(a.p1, cur.c[j].x, S); S = fma(a.p2, flip(cur.c[j].y), S);
                T = fma(b2.p1, cur.c[j + 1].x, T); T = fma(b2.p2, flip(cur.c[j + 1].y), T);
            }
        };
        fetch(ta, 0, true);
        for (int kb = 0; kb < n_blk; kb += 2) {
            sum_tile(kb, ta, tb);
            if (kb + 1 < n_blk) sum_tile(kb + 1, tb, ta);
        }
        S += T;
    }

    // ---- pass 2: the chain
    long long t_chain = 0, t_epi = 0, t_all = 0;
    if constexpr (STAMP) t_all = -(long long)__builtin_amdgcn_s_memtime();
    // this lane's half of its env's 64-byte theta tile (coordinates 4 hb .. 4 hb + 3) and of the index word
    float4* __restrict__ trow4 = reinterpret_cast<float4*>(theta + e * (long long)M * 2) + 2 * hb;
    float2* __restrict__ trow2 = reinterpret_cast<float2*>(theta + e * (long long)M * 2) + 4 * hb;
    const double tn = 0.41421356237309503;                  // tan(pi/8)
    auto chain_tile = [&](int kb, Tile& cur, Tile& nxt, int ahead) __attribute__((always_inline)) -> unsigned {
        fetch(nxt, kb + ahead);
        if constexpr (STAMP) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            t_chain -= (long long)__builtin_amdgcn_s_memtime();
        }
        // the tile image becomes (cx, cys) in place: one XOR per coordinate, no copy of the low word
#pragma unroll
        for (int j = 0; j < kSweepBlk; ++j) cur.c[j].y = flip(cur.c[j].y);
        const double S0 = S;
        PairOld po[kSweepBlk];
#pragma unroll
        for (int j = 0; j < kSweepBlk; ++j) po[j] = old_of(cur, j);
        unsigned acc = 0;                                   // (along_self, w_self < 0) of the 8 coordinates, coordinate 0 on top
        float zmin = 1.0f;                                  // reaches 0 when some w_self or S'_self is (as good as) zero
#pragma unroll
        for (int j = 0; j < kSweepBlk; ++j) {
            const bool ok = !PAD || kb * kSweepBlk + j < M;  // wave-uniform; elements past M leave S alone
            const double cx = ok ? cur.c[j].x : 0.0, cys = ok ? cur.c[j].y : 0.0;   // (the tile holds cys: flipped below)
            double r = fma(-po[j].p1, cx, S);
            r = fma(-po[j].p2, cys, r);
            const double r_o = xchg<1>(r);
            const double w = fma(r_o, cys, r * cx);
            const double w_o = xchg<1>(w);
            // masks instead of booleans: every select below is one v_bfi / v_and_or, no scalar-mask logic on the chain
            // |w_other| <= tan(pi/8) |w_self|  <=>  |w_other| - tan(pi/8) |w_self| < 0 (equality only at w = 0: replayed):
            // ONE fused float64 instruction and an arithmetic shift of its sign instead of multiply + compare + select
            unsigned am = (unsigned)(__double2hiint(fma(-tn, fabs(w), fabs(w_o))) >> 31);   // winner along this lane's axis
            asm("" : "+v"(am));      // opaque: or the compiler turns the bit operations below back into a 64-bit compare + selects
            const unsigned bm = (unsigned)dpp_x1((int)am);                      // ... along the other lane's axis
            const unsigned whi = (unsigned)__double2hiint(w);
            const unsigned nlo = ~(am | bm) & R_LO;
            // 1 | 0 | r as three-input bit operations (both masks set: q = 0, replayed)
            const unsigned mag = ((am & (ONE_HI ^ R_HI)) ^ R_HI) & ~(bm & ~am);
            const unsigned nhi = (whi & 0x80000000u) | mag;                     // a signed zero is as good as zero here
            const double n_s = __hiloint2double((int)nhi, (int)nlo);
            const double n_o = __hiloint2double(dpp_x1((int)nhi), (int)nlo);
            S = fma(n_s, cx, r);
            S = fma(-n_o, cys, S);
            float z3;
            asm("v_min3_f32 %0, %1, |%2|, |%3|" : "=v"(z3) : "v"(zmin), "v"(whi), "v"(__double2hiint(S)));
            zmin = ok ? z3 : zmin;
            acc = __builtin_amdgcn_alignbit(acc, am, 31);    // (acc << 1) | along
            acc = __builtin_amdgcn_alignbit(acc, whi, 31);   // (acc << 1) | sign bit of w
        }
        // epilogue: this lane's four coordinates (4 hb + s): code -> float32 image and index of the winner
        float2 out[4];
        unsigned kw = 0;
        if (__builtin_expect(!__any(zmin == 0.f), 1)) {       // (the fall-through: the replay below practically never runs)
            const unsigned acc_o = (unsigned)dpp_x1((int)acc);
            const unsigned fs = (acc >> (hb ? 0 : 8)) & 0xFFu, fo = (acc_o >> (hb ? 0 : 8)) & 0xFFu;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const unsigned code = ((fs >> (2 * (3 - s))) & 3u) | (((fo >> (2 * (3 - s))) & 3u) << 2);
                const PairOut en = tout[code];
                out[s] = make_float2(en.cr, en.ci);
                kw |= (unsigned)en.k << (8 * s);
            }
        } else {
            // q = 0 (all candidates tie: the first wins) or S' = 0 (no candidate scores above 0: the integer 0,
            // ENV:211, 220) somewhere in this tile: replay it exactly from the saved start state
            S = S0;
            float2 o8[kSweepBlk];
            unsigned k8[kSweepBlk];
#pragma unroll
            for (int j = 0; j < kSweepBlk; ++j) {
                const bool ok = !PAD || kb * kSweepBlk + j < M;
                const PairOld p = old_of(cur, j);
                const double cx = ok ? cur.c[j].x : 0.0, cys = ok ? cur.c[j].y : 0.0;
                double r = fma(-p.p1, cx, S);
                r = fma(-p.p2, cys, r);
                const double r_o = xchg<1>(r);
                const double w = fma(r_o, cys, r * cx);
                const double w_o = xchg<1>(w);
                const bool along = fabs(w_o) <= tn * fabs(w);
                const bool along_o = dpp_x1(along ? 1 : 0) != 0;
                const bool tie = along && along_o;           // w = 0: k = 0, the phasor (1, 0)
                const int neg = (int)((unsigned)__double2hiint(w) >> 31), neg_o = dpp_x1(neg);
                // (both lanes "along" -> the table's k = 0 entry whatever the signs of the zeros)
                const unsigned code = ((along ? 2u : 0u) | (unsigned)neg) | (((along_o ? 2u : 0u) | (unsigned)neg_o) << 2);
                const PairOut en = tout[code];
                // this lane's component of the winner, exact float64
                const double rr = 0.70710678118654757;
                double n_se = along ? (neg ? -1.0 : 1.0) : (along_o ? 0.0 : (neg ? -rr : rr));
                n_se = tie ? (hb ? 0.0 : 1.0) : n_se;
                const double n_oe = xchg<1>(n_se);
                double nS = fma(n_se, cx, r);
                nS = fma(-n_oe, cys, nS);
                const bool z_s = nS == 0.0, z_o = dpp_x1(z_s ? 1 : 0) != 0;
                const bool none = ok && z_s && z_o;          // no candidate scores above 0
                S = none ? r : nS;
                o8[j] = none ? make_float2(0.f, 0.f) : make_float2(en.cr, en.ci);
                k8[j] = none ? 8u : (unsigned)en.k;
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                out[s] = hb ? o8[s + 4] : o8[s];
                kw |= (hb ? k8[s + 4] : k8[s]) << (8 * s);
            }
        }
        if constexpr (STAMP) {
            asm volatile("" ::"v"(S), "v"(acc));
            const long long now = (long long)__builtin_amdgcn_s_memtime();
            t_chain += now;
            t_epi -= now;
        }
        if (live) {
            if constexpr (WTH) {
                if (!PAD) {
                    trow4[kb * 4] = make_float4(out[0].x, out[0].y, out[1].x, out[1].y);
                    trow4[kb * 4 + 1] = make_float4(out[2].x, out[2].y, out[3].x, out[3].y);
                } else {
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        if (kb * kSweepBlk + 4 * hb + s < M) trow2[kb * kSweepBlk + s] = out[s];
                }
            }
        }
        // (tests and `return_idx` only: one wave-uniform, not-taken branch on the production path -- as per-lane
        // conditions these four stores were four taken branches, i.e. four instruction-fetch restarts, per tile)
        if (!STAMP && __builtin_expect(idx_out != nullptr, 0)) {
            if (live) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int m = kb * kSweepBlk + 4 * hb + s;
                    const unsigned kn = (kw >> (8 * s)) & 15u;
                    if (!PAD || m < M) idx_out[e * M + m] = kn == 8u ? -1 : (int)kn;
                }
            }
        }
        if constexpr (STAMP) t_epi += (long long)__builtin_amdgcn_s_memtime();
        return kw;
    };
    {
        // a ring of RING tile images, RING - 1 tiles ahead of the chain: the requests a wavefront keeps in flight
        // (RING - 1) x 4 KiB; every slot has its own copy of the tile code, so the images never move between registers
        Tile ring[RING];
#pragma unroll
        for (int r = 0; r + 1 < RING; ++r) fetch(ring[r], r);
        // The candidate indices move by GROUPS of four tiles (32 bytes of the env's row): one 2 x 16-byte read a group
        // ahead, and ONE 16-byte store per lane per group -- the even lane the first two tiles' words, the odd lane the
        // last two, halves swapped through DPP.  (A 4-byte store per lane and tile touched 32 cache lines per
        // instruction; loads and stores retire in order, so the tiles' c requests queued behind those stores:
        // +30 cycles per coordinate in the s_memtime stamps.)
        const int n_grp = (n_blk + 3) / 4;
        uint4 ic[2], inx[2];
        auto fetch_idx = [&](uint4 (&dst)[2], int g) __attribute__((always_inline)) {
            const uint4* __restrict__ p = reinterpret_cast<const uint4*>(ig + (g < n_grp ? g : n_grp - 1) * 32);
            dst[0] = p[0]; dst[1] = p[1];
        };
        fetch_idx(ic, 0);
        // FULL groups run straight through (no per-tile test: each was a taken branch out and back); a last partial group
        // (n_blk % 4 != 0) tests every tile
        auto do_group = [&](int kb, auto full_tag) __attribute__((always_inline)) {
            constexpr bool FULL = decltype(full_tag)::value;
            fetch_idx(inx, kb / 4 + 1);
            unsigned kwq[4] = {0x08080808u, 0x08080808u, 0x08080808u, 0x08080808u};
#pragma unroll
            for (int r = 0; r < RING; ++r) {
                if (FULL || kb + r < n_blk) {
                    const uint4 w = ic[r >> 1];
                    ring[r].k = (r & 1) ? make_uint2(w.z, w.w) : make_uint2(w.x, w.y);
                    kwq[r] = chain_tile(kb + r, ring[r], ring[(r + RING - 1) % RING], RING - 1);
                }
            }
            const unsigned r0 = (unsigned)dpp_x1((int)(hb ? kwq[0] : kwq[2])), r1 = (unsigned)dpp_x1((int)(hb ? kwq[1] : kwq[3]));
            const uint4 wq = hb ? make_uint4(r0, kwq[2], r1, kwq[3]) : make_uint4(kwq[0], r0, kwq[1], r1);
            if (live) *reinterpret_cast<uint4*>(ig + (kb / 4) * 32 + 16 * hb) = wq;
            ic[0] = inx[0]; ic[1] = inx[1];
        };
        int kb = 0;
        for (; kb + RING <= n_blk; kb += RING) do_group(kb, std::true_type{});
        if (kb < n_blk) do_group(kb, std::false_type{});
    }
    if (live && s_sum) s_sum[e * 2 + hb] = S;
    if constexpr (STAMP) {
        t_all += (long long)__builtin_amdgcn_s_memtime();
        if (lane == 0 && idx_out) {
            idx_out[blockIdx.x * 4 + 0] = (int)t_chain;
            idx_out[blockIdx.x * 4 + 1] = (int)t_epi;
            idx_out[blockIdx.x * 4 + 2] = (int)t_all;
            idx_out[blockIdx.x * 4 + 3] = n_blk;
        }
    }
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
hipError_t launch_colsum(const RisVecState& s, hipStream_t st) {
    static const bool no_slab = std::getenv("RISVEC_NO_COLSUM_SLAB") != nullptr;       // A/B switches for experiments
    static const bool no_rows = std::getenv("RISVEC_NO_COLSUM_ROWS") != nullptr;
    if (s.n_ris == kRowsM && s.n_veh <= 16 && !no_rows && !no_slab) {
        static const char* nt_env = std::getenv("RISVEC_COLSUM_NT");              // "0" / "1" force it off / on (tests, A/B)
        const bool nt = nt_env ? nt_env[0] == '1' : (long long)s.n_envs * s.n_veh * s.n_ris * 8 > tuning().colsum_nt_from;
        const dim3 g((unsigned)(((long long)s.n_envs + kRowsEnvs - 1) / kRowsEnvs)), b(kRowsThreads);
        if (s.n_veh <= 8) {
            if (nt) hipLaunchKernelGGL((k_colsum_rows256<8, true>), g, b, 0, st, dims_of(s), s.h_r, s.b, s.c_col);
            else hipLaunchKernelGGL((k_colsum_rows256<8, false>), g, b, 0, st, dims_of(s), s.h_r, s.b, s.c_col);
        } else {
            if (nt) hipLaunchKernelGGL((k_colsum_rows256<16, true>), g, b, 0, st, dims_of(s), s.h_r, s.b, s.c_col);
            else hipLaunchKernelGGL((k_colsum_rows256<16, false>), g, b, 0, st, dims_of(s), s.h_r, s.b, s.c_col);
        }
        return hipGetLastError();
    }
    if (s.n_ris % kSlabM == 0 && s.n_veh <= 16 && !no_slab) {
        const long long blocks = ((long long)s.n_envs + kWave - 1) / kWave;
        if (blocks < (1LL << 31)) {
            // h_r streams larger than the Infinity Cache are read with the non-temporal hint (see launch_pipe)
            static const char* nt_env = std::getenv("RISVEC_COLSUM_NT");          // "0" / "1" force it off / on (tests, A/B)
            const bool nt = nt_env ? nt_env[0] == '1' : (long long)s.n_envs * s.n_veh * s.n_ris * 8 > tuning().colsum_nt_from;
            const dim3 g((unsigned)blocks), b(kSlabThreads);
            if (s.n_veh <= 8) {
                if (nt) hipLaunchKernelGGL((k_colsum_slab<8, true>), g, b, 0, st, dims_of(s), s.h_r, s.b, s.c_col);
                else hipLaunchKernelGGL((k_colsum_slab<8, false>), g, b, 0, st, dims_of(s), s.h_r, s.b, s.c_col);
            } else {
                if (nt) hipLaunchKernelGGL((k_colsum_slab<16, true>), g, b, 0, st, dims_of(s), s.h_r, s.b, s.c_col);
                else hipLaunchKernelGGL((k_colsum_slab<16, false>), g, b, 0, st, dims_of(s), s.h_r, s.b, s.c_col);
            }
            return hipGetLastError();
        }
    }
    const bool even = (s.n_ris & 1) == 0;
    const long long n_slot = (long long)s.n_envs * (even ? s.n_ris / 2 : s.n_ris);
    const unsigned grid = (unsigned)((n_slot + kBlock / 2 - 1) / (kBlock / 2));
    if (even) hipLaunchKernelGGL((k_colsum<2>), dim3(grid), dim3(kBlock), 0, st, dims_of(s), s.h_r, s.b, s.c_col);
    else hipLaunchKernelGGL((k_colsum<1>), dim3(grid), dim3(kBlock), 0, st, dims_of(s), s.h_r, s.b, s.c_col);
    return hipGetLastError();
}

template <int NC>
static hipError_t launch_sweep_nc(const RisVecState& s, int32_t* idx_out, bool reuse_s, hipStream_t st) {
    const unsigned grid = (unsigned)((s.n_envs + kWave - 1) / kWave);
    hipLaunchKernelGGL((k_bcd_sweep<NC>), dim3(grid), dim3(kWave), 0, st, dims_of(s), s.c_col, s.theta, idx_out,
                       s.s_sum, (reuse_s && s.s_sum) ? 1 : 0, NC == 8 ? s.theta_idx : nullptr);
    note_kernel("k_bcd_sweep<%d>", NC);
    return hipGetLastError();
}

// theta[e, m] = the float32 image of candidate theta_idx[e, m] (ENV:169, 213; 8 = the integer 0): materialises the
// complex64 tensor after sweeps that kept theta by index.  One lane per 4 elements: 4 bytes in, 32 bytes out.
__global__ void __launch_bounds__(kBlock)
k_theta_from_index(Dims d, const uint8_t* __restrict__ theta_idx, float* __restrict__ theta) {
    const int MK = theta_idx_stride(d.M), q = MK / 4;
    const long long t = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (t >= (long long)d.E * q) return;
    const long long e = t / q;
    const int m0 = (int)(t - e * q) * 4;
    const unsigned w = *reinterpret_cast<const unsigned*>(theta_idx + e * MK + m0);
    const float r = 0.70710677f;
    float2 o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned k = (w >> (8 * j)) & 15u;
        float c = (k & 3u) == 2u ? 0.f : ((k & 1u) ? r : 1.f);
        float sn = (k & 3u) == 0u ? 0.f : ((k & 1u) ? r : 1.f);
        if (k >= 3u && k <= 5u) c = -c;
        if (k >= 5u) sn = -sn;
        o[j] = k < 8u ? make_float2(c, sn) : make_float2(0.f, 0.f);
    }
    float2* __restrict__ row = reinterpret_cast<float2*>(theta) + e * d.M;
    if (m0 + 3 < d.M && (d.M & 1) == 0) {
        float4* r4 = reinterpret_cast<float4*>(row + m0);
        r4[0] = make_float4(o[0].x, o[0].y, o[1].x, o[1].y);
        r4[1] = make_float4(o[2].x, o[2].y, o[3].x, o[3].y);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (m0 + j < d.M) row[m0 + j] = o[j];
    }
}

hipError_t launch_theta_from_index(const RisVecState& s, hipStream_t st) {
    const long long n = (long long)s.n_envs * (theta_idx_stride(s.n_ris) / 4);
    hipLaunchKernelGGL(k_theta_from_index, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, dims_of(s),
                       s.theta_idx, s.theta);
    return hipGetLastError();
}

hipError_t launch_bcd(const RisVecState& s, const RisVecParams&, int32_t* idx_out, bool reuse_colsum,
                      bool reuse_s, bool reuse_idx, bool write_theta, hipStream_t st) {
    if (!reuse_colsum) {
        const hipError_t err = launch_colsum(s, st);
        if (err != hipSuccess) return err;
    }
    static const bool no_idx = std::getenv("RISVEC_NO_IDX_SWEEP") != nullptr;     // A/B switch: always the generic sweep
    if (s.control_bit == 3 && reuse_idx && s.theta_idx && (!no_idx || !write_theta)) {
        const unsigned gp = (unsigned)((s.n_envs + kWave / 2 - 1) / (kWave / 2));
        const int rs = (reuse_s && s.s_sum) ? 1 : 0;
        const bool pad = s.n_ris % kSweepBlk != 0;
#ifdef RISVEC_DIAG
        // diagnostic library only (make diag -> librisvec_diag.so, tools/sweep_stamps.py): the s_memtime build of the sweep
        static const char* stamps = std::getenv("RISVEC_SWEEP_STAMPS");
        if (stamps && idx_out && !pad) {
            hipLaunchKernelGGL((k_bcd_sweep8_pair<false, true, true>), dim3(gp), dim3(kWave), 0, st, dims_of(s), s.c_col, s.theta,
                               idx_out, s.s_sum, rs, s.theta_idx);
            return hipGetLastError();
        }
#endif
#define RISVEC_PAIR(PAD, WTH) hipLaunchKernelGGL((k_bcd_sweep8_pair<PAD, WTH>), dim3(gp), dim3(kWave), 0, st, dims_of(s), s.c_col, \
                                                 s.theta, idx_out, s.s_sum, rs, s.theta_idx)
        if (pad) { if (write_theta) RISVEC_PAIR(true, true); else RISVEC_PAIR(true, false); }
        else { if (write_theta) RISVEC_PAIR(false, true); else RISVEC_PAIR(false, false); }
#undef RISVEC_PAIR
        note_kernel("k_bcd_sweep8_pair<%s,%s>", pad ? "PAD" : "M%8=0", write_theta ? "theta written" : "theta by index");
        return hipGetLastError();
    }
    if (!write_theta) return hipErrorNotSupported;            // theta by index needs current indices and 2^b = 8
    switch (s.control_bit) {
        case 0: return launch_sweep_nc<1>(s, idx_out, reuse_s, st);
        case 1: return launch_sweep_nc<2>(s, idx_out, reuse_s, st);
        case 2: return launch_sweep_nc<4>(s, idx_out, reuse_s, st);
        case 3: return launch_sweep_nc<8>(s, idx_out, reuse_s, st);
        case 4: return launch_sweep_nc<16>(s, idx_out, reuse_s, st);
        case 5: return launch_sweep_nc<32>(s, idx_out, reuse_s, st);
        case 6: return launch_sweep_nc<64>(s, idx_out, reuse_s, st);
        default: return hipErrorInvalidValue;
    }
}

// BCD + gains + step: the sweep, then the fused gain+step kernel on the same stream.
hipError_t launch_step_fused_bcd(const RisVecState& s, const RisVecParams& p, const float* action,
                                 const int32_t* partner, const int32_t* n_groups,
                                 const int32_t* arrivals, uint64_t seed, uint32_t counter,
                                 uint32_t flags, hipStream_t st) {
    const bool by_index = (flags & RISVEC_STEP_THETA_BY_INDEX) != 0;
    hipError_t err = launch_bcd(s, p, nullptr, (flags & RISVEC_STEP_REUSE_COLSUM) != 0,
                                (flags & RISVEC_STEP_REUSE_SSUM) != 0, (flags & RISVEC_STEP_REUSE_IDX) != 0, !by_index, st);
    if (err != hipSuccess) return err;
    return launch_step(s, p, action, partner, n_groups, arrivals, seed, counter,
                       flags & ~(uint32_t)(RISVEC_STEP_REUSE_COLSUM | RISVEC_STEP_REUSE_SSUM | RISVEC_STEP_REUSE_IDX), true, st);
}

}  // namespace risvec
