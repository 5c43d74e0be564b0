// Slow-cadence kernels of the RIS-VEC environment: reset, mobility, geometry,
// 3GPP gains, phase setters.  One thread per (env, vehicle) [or per RIS element];
// positions and everything derived from them are advanced in float64 so that the
// steering-vector phases (|arg| up to ~800 rad at m = 255) keep the 1e-5 budget.
//
// Reference: Simulation-MARL-BCD/Environment.py (ENV): make_new_game ENV:733-737,
// add_new_vehicles_by_number ENV:381-410, renew_positions ENV:412-542,
// compute_parms ENV:241-253, update_channel_gains (3GPP) ENV:275-327,
// get_next_phase ENV:233-239, Random_phase ENV:203-206.
#include "risvec_launch.hpp"

namespace risvec {

// ENV:29-42
__device__ constexpr double kRisX = 220.0, kRisY = 220.0, kRisZ = 25.0;
__device__ constexpr double kBsX = 0.0, kBsY = 0.0, kBsZ = 25.0;
__device__ constexpr double kVehZ = 1.5;
__device__ constexpr double kRo = 1e-2, kAlpha1 = 2.2, kAlpha2 = 2.5;

// ---------------------------------------------------------------------------
// K1 reset
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
k_reset(Dims d, RisVecParams P, double* __restrict__ pos, int32_t* __restrict__ dir,
        float* __restrict__ vel, float* __restrict__ data_buf,
        const int32_t* __restrict__ spawn_ints, const int32_t* __restrict__ buf0,
        uint64_t seed, uint32_t counter) {
    const long long idx = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (idx >= (long long)d.E * d.V) return;
    const int e = (int)(idx / d.V), v = (int)(idx % d.V);
    const uint32_t genv = (uint32_t)(d.env_offset + e);
    const int n_round4 = 4 * (d.V / 4);
    int aux, coord, velo, b0;
    if (spawn_ints) {
        aux = spawn_ints[idx * 3 + 0];
        coord = spawn_ints[idx * 3 + 1];
        velo = spawn_ints[idx * 3 + 2];
        b0 = buf0[e];
    } else {
        const uint4 r = philox4x32_10(genv, (uint32_t)v, counter, kSiteSpawn, seed);
        const int lane = randint_u32(r.x, 0, P.n_lanes);                 // ENV:384, 403
        if (v < n_round4) {
            const int slot = v & 3;
            aux = lane;
            coord = (slot == 0 || slot == 2) ? randint_u32(r.y, 220, 230)   // ENV:386, 394
                                             : randint_u32(r.y, 170, 180);  // ENV:390, 398
            velo = randint_u32(r.z, 10, 15);                                // ENV:388-400
        } else {
            aux = lane + 4 * randint_u32(r.w, 0, 4);                        // ENV:404
            coord = randint_u32(r.y, 0, (int)P.height);                     // ENV:405
            velo = randint_u32(r.z, 15, 20);                                // ENV:407
        }
        const uint4 rb = philox4x32_10(genv, 0u, counter, kSiteBuf0, seed);
        b0 = randint_u32(rb.x, 5, 9);                                       // ENV:737
    }
    double x, y;
    int dr;
    if (v < n_round4) {
        switch (v & 3) {
            case 0: x = P.lanes_down[aux]; y = (double)coord; dr = RISVEC_DIR_D; break;
            case 1: x = P.lanes_up[0]; y = (double)coord; dr = RISVEC_DIR_U; break;
            case 2: x = (double)coord; y = P.lanes_left[0]; dr = RISVEC_DIR_L; break;
            default: x = (double)coord; y = P.lanes_right[0]; dr = RISVEC_DIR_R; break;
        }
    } else {
        const int choice = aux >> 2;                                        // index into 'dulr'
        x = P.lanes_down[aux & 3];
        y = (double)coord;
        dr = choice == 0 ? RISVEC_DIR_D : choice == 1 ? RISVEC_DIR_U
           : choice == 2 ? RISVEC_DIR_L : RISVEC_DIR_R;
    }
    pos[idx * 2 + 0] = x;
    pos[idx * 2 + 1] = y;
    dir[idx] = dr;
    vel[idx] = (float)velo;
    data_buf[idx] = (float)b0 * 0.5f;                                       // ENV:737
}

// ---------------------------------------------------------------------------
// K2a mobility
// ---------------------------------------------------------------------------
struct TurnDraws {
    const float* inj;      // injected draws for this vehicle, or nullptr
    uint32_t genv, v, counter;
    uint64_t seed;
    int nd;
    uint4 blk;
    __device__ double next() {
        float u;
        if (inj) {
            u = inj[nd];
        } else {
            const int k = nd & 3;
            if (k == 0) blk = philox4x32_10(genv, v, counter, nd < 4 ? kSiteTurnA : kSiteTurnB, seed);
            u = u01(k == 0 ? blk.x : k == 1 ? blk.y : k == 2 ? blk.z : blk.w);
        }
        ++nd;
        return (double)u;
    }
};

__global__ void __launch_bounds__(kBlock)
k_mobility(Dims d, RisVecParams P, double* __restrict__ pos, int32_t* __restrict__ dir,
           const float* __restrict__ vel, const float* __restrict__ u_turn,
           int32_t* __restrict__ n_used, uint64_t seed, uint32_t counter) {
#pragma clang fp contract(off)
    const long long idx = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (idx >= (long long)d.E * d.V) return;
    const int e = (int)(idx / d.V), v = (int)(idx % d.V);
    double x = pos[idx * 2], y = pos[idx * 2 + 1];
    int dr = dir[idx];
    const double dd = (double)vel[idx] * P.time_slow;                       // ENV:419
    TurnDraws draw{u_turn ? u_turn + idx * 8 : nullptr, (uint32_t)(d.env_offset + e), (uint32_t)v,
                   counter, seed, 0, make_uint4(0, 0, 0, 0)};
    const int nl = P.n_lanes;
    bool turned = false;
    if (dr == RISVEC_DIR_U) {                                               // ENV:421-446
        for (int j = 0; j < nl && !turned; ++j) {
            const double ln = P.lanes_left[j];
            if (y <= ln && (y + dd) >= ln && draw.next() < 0.4) {
                x = x - (dd - (ln - y)); y = ln; dr = RISVEC_DIR_L; turned = true;
            }
        }
        for (int j = 0; j < nl && !turned; ++j) {
            const double ln = P.lanes_right[j];
            if (y <= ln && (y + dd) >= ln && draw.next() < 0.4) {           // '+': ENV:439-440
                x = x + (dd + (ln - y)); y = ln; dr = RISVEC_DIR_R; turned = true;
            }
        }
        if (!turned) y += dd;
    }
    if (dr == RISVEC_DIR_D && !turned) {                                    // ENV:447-473
        for (int j = 0; j < nl && !turned; ++j) {
            const double ln = P.lanes_left[j];
            if (y >= ln && (y - dd) <= ln && draw.next() < 0.4) {
                x = x - (dd - (y - ln)); y = ln; dr = RISVEC_DIR_L; turned = true;
            }
        }
        for (int j = 0; j < nl && !turned; ++j) {
            const double ln = P.lanes_right[j];
            if (y >= ln && (y - dd) <= ln && draw.next() < 0.4) {           // '+': ENV:465-466
                x = x + (dd + (y - ln)); y = ln; dr = RISVEC_DIR_R; turned = true;
            }
        }
        if (!turned) y -= dd;
    }
    if (dr == RISVEC_DIR_R && !turned) {                                    // ENV:474-496
        for (int j = 0; j < nl && !turned; ++j) {
            const double ln = P.lanes_up[j];
            if (x <= ln && (x + dd) >= ln && draw.next() < 0.4) {
                y = y + (dd - (ln - x)); x = ln; dr = RISVEC_DIR_U; turned = true;
            }
        }
        for (int j = 0; j < nl && !turned; ++j) {
            const double ln = P.lanes_down[j];
            if (x <= ln && (x + dd) >= ln && draw.next() < 0.4) {
                y = y - (dd - (ln - x)); x = ln; dr = RISVEC_DIR_D; turned = true;
            }
        }
        if (!turned) x += dd;
    }
    if (dr == RISVEC_DIR_L && !turned) {                                    // ENV:497-519
        for (int j = 0; j < nl && !turned; ++j) {
            const double ln = P.lanes_up[j];
            if (x >= ln && (x - dd) <= ln && draw.next() < 0.4) {
                y = y + (dd - (x - ln)); x = ln; dr = RISVEC_DIR_U; turned = true;
            }
        }
        for (int j = 0; j < nl && !turned; ++j) {
            const double ln = P.lanes_down[j];
            if (x >= ln && (x - dd) <= ln && draw.next() < 0.4) {
                y = y - (dd - (x - ln)); x = ln; dr = RISVEC_DIR_D; turned = true;
            }
        }
        if (!turned) x -= dd;
    }
    if (x < 0.0 || y < 0.0 || x > P.width || y > P.height) {                // ENV:522-540
        if (dr == RISVEC_DIR_U) { dr = RISVEC_DIR_R; y = P.lanes_right[nl - 1]; }
        else if (dr == RISVEC_DIR_D) { dr = RISVEC_DIR_L; y = P.lanes_left[0]; }
        else if (dr == RISVEC_DIR_L) { dr = RISVEC_DIR_U; x = P.lanes_up[0]; }
        else { dr = RISVEC_DIR_D; x = P.lanes_down[nl - 1]; }
    }
    pos[idx * 2] = x;
    pos[idx * 2 + 1] = y;
    dir[idx] = dr;
    if (n_used) n_used[idx] = draw.nd;
}

// ---------------------------------------------------------------------------
// K2b geometry: one thread per (env, vehicle, chunk of kGeoChunk RIS elements)
// ---------------------------------------------------------------------------
// phases_R_i[v][m] = exp(-j 2 pi/lamb d ang m) = exp(-j pi ang m) (lamb = 1, d = 0.5; ENV:253).
// |arg| reaches ~800 rad at m = 255, so everything is float64 until the final rounding.  A thread
// evaluates two sincospi - the chunk's first element and the per-element rotation w =
// exp(-j pi ang) - and walks the chunk with z <- z w (16 rotations: ~2e-15 accumulated error),
// instead of one sincospi per element.  A chunk is 128 contiguous bytes of the row.
constexpr int kGeoChunk = 16;

template <bool STAGED>
__global__ void __launch_bounds__(kBlock)
k_geometry(Dims d, const double* __restrict__ pos, float* __restrict__ dist_r,
           float* __restrict__ ang_r, float* __restrict__ pl, float* __restrict__ h_r, double* __restrict__ z_r) {
    const int nchunk = (d.M + kGeoChunk - 1) / kGeoChunk;
    long long idx = (long long)blockIdx.x * kBlock + threadIdx.x;
    const bool in_range = idx < (long long)d.E * d.V * nchunk;
    if (!in_range) {
        if constexpr (!STAGED) return;
        idx = (long long)d.E * d.V * nchunk - 1;   // staged form: stay for the barrier, recompute the last chunk
    }
    const long long ev = idx / nchunk;
    const int m0 = (int)(idx % nchunk) * kGeoChunk;
    const double x = pos[ev * 2], y = pos[ev * 2 + 1];
    const double dx = x - kRisX, dy = y - kRisY, dz = kVehZ - kRisZ;
    const double dist = sqrt(dx * dx + dy * dy + dz * dz);                  // ENV:245-246
    const double ang = dx / dist;                                           // ENV:248
    if (m0 == 0 && in_range) {
        dist_r[ev] = (float)dist;
        ang_r[ev] = (float)ang;
        const double d_br = sqrt((kBsX - kRisX) * (kBsX - kRisX) + (kBsY - kRisY) * (kBsY - kRisY)
                                 + (kBsZ - kRisZ) * (kBsZ - kRisZ));        // ENV:175-176
        pl[ev] = (float)((kRo * kRo) / (pow(dist, kAlpha1) * pow(d_br, kAlpha2)));   // ENV:270-272
    }
    double zs, zc, ws, wc;
    sincospi(ang * (double)m0, &zs, &zc);          // z = exp(-j pi ang m0) = (zc, -zs)
    sincospi(ang, &ws, &wc);                       // w = exp(-j pi ang)    = (wc, -ws)
    double zr = zc, zi = -zs;
    const double wr = wc, wi = -ws;
    if (z_r && m0 == 0 && in_range) reinterpret_cast<double2*>(z_r)[ev] = make_double2(wr, wi);   // h_r[e,v,m] = w^m
    float* out = h_r + (ev * d.M + m0) * 2;
    if constexpr (STAGED) {
        // M % 16 == 0: every lane owns one full 128-byte chunk and the block's 256 chunks are contiguous
        // in h_r.  Park the chunks in LDS (rows padded to 144 B against bank conflicts) and write the
        // block's 32 KB back with consecutive lanes on consecutive 16-byte pieces (1 KB per instruction)
        // instead of 64 partial lines per instruction.
        __shared__ float4 s_stage[kBlock * 9];
        float4* row = s_stage + threadIdx.x * 9;
#pragma unroll
        for (int k = 0; k < kGeoChunk; k += 2) {
            const double ar = zr, ai = zi;
            const double br = ar * wr - ai * wi, bi = ar * wi + ai * wr;        // next element
            zr = br * wr - bi * wi; zi = br * wi + bi * wr;                     // the one after
            row[k / 2] = make_float4((float)ar, (float)ai, (float)br, (float)bi);
        }
        __syncthreads();
        const long long block_first = (long long)blockIdx.x * kBlock;          // first chunk of this block
        const long long n_chunks = (long long)d.E * d.V * nchunk;
        float4* gout = reinterpret_cast<float4*>(h_r) + block_first * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int f = k * kBlock + threadIdx.x;                             // 16-byte piece within the block
            if (block_first + f / 8 < n_chunks) gout[f] = s_stage[(f / 8) * 9 + (f % 8)];
        }
        return;
    }
    const bool vec = (d.M & 1) == 0;               // even M: rows are 16-byte aligned, pairs never straddle
#pragma unroll
    for (int k = 0; k < kGeoChunk; k += 2) {
        const double ar = zr, ai = zi;
        const double br = ar * wr - ai * wi, bi = ar * wi + ai * wr;        // next element
        zr = br * wr - bi * wi; zi = br * wi + bi * wr;                     // the one after
        if (m0 + k + 1 < d.M) {
            if (vec) *reinterpret_cast<float4*>(out + 2 * k) = make_float4((float)ar, (float)ai, (float)br, (float)bi);
            else { out[2 * k] = (float)ar; out[2 * k + 1] = (float)ai; out[2 * k + 2] = (float)br; out[2 * k + 3] = (float)bi; }
        } else if (m0 + k < d.M) {
            out[2 * k] = (float)ar; out[2 * k + 1] = (float)ai;
        }
    }
}

// ---------------------------------------------------------------------------
// K3b gains, 3GPP TR 38.901-style models (RIS ignored; ENV:275-327)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
k_gain_3gpp(Dims d, RisVecParams P, int model, const double* __restrict__ pos,
            float* __restrict__ gain, const float* __restrict__ u_los,
            const float* __restrict__ z_shadow, const float* __restrict__ small_in,
            uint64_t seed, uint32_t counter) {
    const long long idx = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (idx >= (long long)d.E * d.V) return;
    const int e = (int)(idx / d.V), v = (int)(idx % d.V);
    double u, z, sm;
    if (u_los) {
        u = u_los[idx]; z = z_shadow[idx]; sm = small_in[idx];
    } else {
        const uint32_t genv = (uint32_t)(d.env_offset + e);
        const uint4 r = philox4x32_10(genv, (uint32_t)v, counter, kSite3gpp, seed);
        u = u01(r.x);                                                       // ENV:299
        const float2 n = normal2(r.y, r.z);
        z = n.x;                                                            // ENV:10
        if (P.rician_k_db <= 1e-6f) {
            sm = -log(((double)(r.w >> 8) + 1.0) * 0x1p-24);               // Exp(1), ENV:17
        } else {                                                            // ENV:19-25
            const uint4 r2 = philox4x32_10(genv, (uint32_t)v, counter, kSite3gpp + 0x100u, seed);
            const float2 n2 = normal2(r2.x, r2.y);
            const double K = pow(10.0, (double)P.rician_k_db / 10.0);
            const double s = sqrt(K / (K + 1.0)), sg = 1.0 / sqrt(2.0 * (K + 1.0));
            const double hr = s + sg * n2.x, hi = sg * n2.y;
            sm = hr * hr + hi * hi;
        }
    }
    const double dx = fabs(pos[idx * 2] - kBsX), dy = fabs(pos[idx * 2 + 1] - kBsY);
    const double dz = fabs(kBsZ - kVehZ);
    const double d2d = hypot(dx, dy);
    const double d3d = sqrt(d2d * d2d + dz * dz);
    const bool los = u < 0.7 * exp(-d2d / 200.0);                           // ENV:298-299
    const double fc = (double)P.fc_ghz;
    const double ld = log10(fmax(d3d, 1.0)), lf = log10(fc);
    double pl_db = 0.0;                                                     // ENV:315-317
    if (model == RISVEC_CH_3GPP_UMI)
        pl_db = los ? 32.4 + 21.0 * lf + 20.0 * ld : 36.7 + 22.7 * lf + 26.0 * ld;       // ENV:281,285
    else if (model == RISVEC_CH_3GPP_UMA)
        pl_db = los ? 28.0 + 22.0 * lf + 20.0 * ld
                    : 13.54 + 39.08 * ld + 20.0 * lf - 0.6 * (double)P.veh_ant_gain;     // ENV:289,293
    const double large = pow(10.0, -pl_db / 10.0);
    const double sd = los ? (double)P.shadow_std_los : (double)P.shadow_std_nlos;
    const double shadow = pow(10.0, (z * sd) / 10.0);                       // ENV:10-11
    gain[idx] = (float)(large * shadow * sm);                               // ENV:327
}

// ---------------------------------------------------------------------------
// phase setters
// ---------------------------------------------------------------------------
// exp(j x) in float64 for |x| <= 1e5: Cody-Waite reduction by pi/2 (two-term split, exact product for |k| < 2^17) and the
// fdlibm kernel polynomials on [-pi/4, pi/4] -- error < 1e-15, i.e. the float32 image is the correctly rounded one except
// within 1e-8 ulp of a rounding boundary.  ~30 float64 instructions against the ~150 of the general sincos(), whose
// range reduction for huge arguments and special cases made k_set_phase VALU-bound (9 us for 25 MB at E = 32 768, M = 64:
// the SARL step's theta = exp(j action_phase), every step).
__device__ __forceinline__ void sincos_small(double x, double& s, double& c) {
    const double kd = rint(x * 0.63661977236758138);                        // round(x * 2/pi)
    double r = fma(-kd, 1.5707963267341256, x);                              // pi/2, high 33 bits
    r = fma(-kd, 6.0771005065061922e-11, r);                                 // pi/2, next 53 bits
    const double z = r * r;
    const double ps = fma(z, fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08),
                                                  2.75573137070700676789e-06), -1.98412698298579493134e-04),
                                 8.33333333332248946124e-03), -1.66666666666666324348e-01);
    const double pc = fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09),
                                                  -2.75573143513906633035e-07), 2.48015872894767294178e-05),
                                 -1.38888888888741095749e-03), 4.16666666666666019037e-02);
    const double sr = fma(r * z, ps, r);
    const double cr = fma(z * z, pc, fma(-0.5, z, 1.0));
    const int q = (int)kd & 3;
    const double a = (q & 1) ? cr : sr, b = (q & 1) ? sr : cr;               // sin, cos before the signs
    s = (q & 2) ? -a : a;
    c = (q == 1 || q == 2) ? -b : b;
}

__global__ void __launch_bounds__(kBlock)
k_set_phase(long long n, const float* __restrict__ angle, float* __restrict__ theta) {
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const double x = (double)angle[i];
    double s, c;
    if (fabs(x) <= 1.0e5) sincos_small(x, s, c);
    else sincos(x, &s, &c);                                                  // ENV:239
    *reinterpret_cast<float2*>(theta + 2 * i) = make_float2((float)c, (float)s);
}

__global__ void __launch_bounds__(kBlock)
k_random_phase(Dims d, const int32_t* __restrict__ idx, float* __restrict__ theta, uint64_t seed,
               uint32_t counter) {
    __shared__ float2 s_cand[64];                                           // exp(j possible_angles[k]), ENV:169
    const int nc = 1 << d.cbit;
    if ((int)threadIdx.x < nc) {
        double s, c;
        sincospi(2.0 * (double)threadIdx.x / (double)nc, &s, &c);
        s_cand[threadIdx.x] = make_float2((float)c, (float)s);
    }
    __syncthreads();
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= (long long)d.E * d.M) return;
    const int e = (int)(i / d.M), m = (int)(i % d.M);
    int k;
    if (idx) {
        k = idx[i] & (nc - 1);
    } else {
        const uint4 r = philox4x32_10((uint32_t)(d.env_offset + e), (uint32_t)m, counter, kSitePhase, seed);
        k = randint_u32(r.x, 0, nc);                                        // ENV:204
    }
    *reinterpret_cast<float2*>(theta + 2 * i) = s_cand[k];                  // ENV:206
}

// ---------------------------------------------------------------------------
static inline unsigned blocks_for(long long n) { return (unsigned)((n + kBlock - 1) / kBlock); }

hipError_t launch_reset(const RisVecState& s, const RisVecParams& p, const int32_t* spawn_ints,
                        const int32_t* buf0, uint64_t seed, uint32_t counter, hipStream_t st) {
    const long long n = (long long)s.n_envs * s.n_veh;
    hipLaunchKernelGGL(k_reset, dim3(blocks_for(n)), dim3(kBlock), 0, st, dims_of(s), p, s.pos, s.dir,
                       s.vel, s.data_buf, spawn_ints, buf0, seed, counter);
    return hipGetLastError();
}

hipError_t launch_mobility(const RisVecState& s, const RisVecParams& p, const float* u_turn,
                           int32_t* n_used, uint64_t seed, uint32_t counter, hipStream_t st) {
    const long long n = (long long)s.n_envs * s.n_veh;
    hipLaunchKernelGGL(k_mobility, dim3(blocks_for(n)), dim3(kBlock), 0, st, dims_of(s), p, s.pos,
                       s.dir, s.vel, u_turn, n_used, seed, counter);
    return hipGetLastError();
}

hipError_t launch_geometry(const RisVecState& s, const RisVecParams&, hipStream_t st) {
    const long long n = (long long)s.n_envs * s.n_veh * ((s.n_ris + kGeoChunk - 1) / kGeoChunk);
    if (s.n_ris % kGeoChunk == 0)
        hipLaunchKernelGGL(k_geometry<true>, dim3(blocks_for(n)), dim3(kBlock), 0, st, dims_of(s), s.pos,
                           s.dist_r, s.ang_r, s.pl, s.h_r, s.z_r);
    else
        hipLaunchKernelGGL(k_geometry<false>, dim3(blocks_for(n)), dim3(kBlock), 0, st, dims_of(s), s.pos,
                           s.dist_r, s.ang_r, s.pl, s.h_r, s.z_r);
    return hipGetLastError();
}

hipError_t launch_gain_3gpp(const RisVecState& s, const RisVecParams& p, int32_t model,
                            const float* u_los, const float* z_shadow, const float* small,
                            uint64_t seed, uint32_t counter, hipStream_t st) {
    const long long n = (long long)s.n_envs * s.n_veh;
    hipLaunchKernelGGL(k_gain_3gpp, dim3(blocks_for(n)), dim3(kBlock), 0, st, dims_of(s), p, model,
                       s.pos, s.gain, u_los, z_shadow, small, seed, counter);
    return hipGetLastError();
}

hipError_t launch_set_phase(const RisVecState& s, const float* angle, hipStream_t st) {
    const long long n = (long long)s.n_envs * s.n_ris;
    hipLaunchKernelGGL(k_set_phase, dim3(blocks_for(n)), dim3(kBlock), 0, st, n, angle, s.theta);
    return hipGetLastError();
}

hipError_t launch_random_phase(const RisVecState& s, const int32_t* idx, uint64_t seed,
                               uint32_t counter, hipStream_t st) {
    const long long n = (long long)s.n_envs * s.n_ris;
    hipLaunchKernelGGL(k_random_phase, dim3(blocks_for(n)), dim3(kBlock), 0, st, dims_of(s), idx,
                       s.theta, seed, counter);
    return hipGetLastError();
}

}  // namespace risvec
