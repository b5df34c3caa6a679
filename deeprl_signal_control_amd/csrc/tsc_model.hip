// tsc_model.hip -- per-intersection actor-critic nets + on-policy A2C update on gfx950.
//
// Replaces, for all agents of all env instances at once:
//   K7 policy forward ........ IA2C/MA2C.forward (agents/models.py:185-200) -> LstmACPolicy /
//                              FPLstmACPolicy._build_net (agents/policies.py:99-118,191-211),
//                              fc / lstm (agents/utils.py:66-74,88-116)
//   sampling ................. np.random.choice per agent (utils.py:155-157)
//   K8 returns / advantages .. OnPolicyBuffer (agents/utils.py:182-228), reward norm/clip
//                              (agents/models.py:222-229)
//   K9 loss + BPTT + update .. ACPolicy.prepare_loss (agents/policies.py:41-61): A2C loss,
//                              per-agent clip_by_global_norm, TF1 RMSProp
//
// Layout: agent-tower group g = 2*agent + tower (0 = pi, 1 = v).  All parameters live in one
// flat fp32 buffer [G][stride] (W1 | b1 | Wx | Wh | bl | Wo | bo per group); gradients and the
// RMSProp accumulator use the same layout, so the gradient buffer is one contiguous RCCL
// all-reduce.  The three input FCs (wave / fingerprint / wait) are one block-diagonal
// [SMAX x H] matrix with structural zeros (kept zero by a row-range mask on its gradient).
// Kernels (all fp32, v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32 for every contraction):
//   rollout ......... policy_fwd_ws_kernel: one launch per control step, [Wx ; Wh] stationary in registers, tiles
//                     software-pipelined over two barrier intervals, also fills the activation cache the update reads
//                     (policy_fwd_fused_kernel: tile-per-workgroup variant; policy_fwd_fc_mfma_kernel: FcACPolicy;
//                     grouped GEMMs + lstm_fwd + head_fwd: training-shape re-forward)
//   update .......... head_bwd2 (persistent, loss gradient + dH + dWo | dbo) -> lstm_bwd2 (Wh^T stationary in registers,
//                     dc in registers / dh through LDS over the n_step time steps, next step's inputs in flight under the
//                     MFMAs) -> dwxh (dWx | dWh | dbl, whole tower output in accumulators) -> dx1w1_kernel2 (dX1 in
//                     registers, chained into dW1 | db1) -> grad_norm -> rmsprop; the FcACPolicy update: head_bwd2 ->
//                     fc_bwd_kernel (dWfc | dbfc | dW1 | db1 in one pass over the X1 / Hh rows its rollout forward cached;
//                     the grouped split-K GEMMs of tsc_gemm.h behind TSC_UNFUSED_DX and for other shapes)
#include "tsc_common.h"
#include "tsc_gemm.h"
#include "../../include/tsc.h"

#include <cstdlib>
#include <type_traits>
#include <vector>

namespace {

using tsc::f32x16;
using tsc::f32x4;
using tsc::GemmArgs;

constexpr int kOut = 8;          // padded head width (n_a <= 8; v uses column 0)
constexpr int kL = 64;           // num_lstm (fixed by the kernels' tiling)
constexpr int kG4 = 4 * kL;      // gate columns i|f|o|u

struct Layout {
    int G, A, SMAX, AMAX, H;
    int NZ, fc;      // second-layer width: 4L gate columns (LSTM) or L units (FcACPolicy, agents/policies.py:214-256)
    long long stride, oW1, ob1, oWx, oWh, obl, oWo, obo;
};

// Gate non-linearities on the hardware exp2 / rcp units (v_exp_f32, v_rcp_f32: ~1 ulp each): 4-5 instructions
// instead of the ~30-50 of the IEEE expf / tanhf / division sequences.  Absolute error < 3e-7, far inside the
// fp32-vs-float64 tolerance the parity tests state; the cell update was 26 % of the fused forward before.
__device__ __forceinline__ float sigmoidf_(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float tanhf_(float x) {
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
}

// The activation cache is written once per control step and not read before the update: streaming stores keep its 160 MB
// per launch from evicting the simulator's vehicle state out of L2 / the infinity cache between two tsc_env_step launches.
// Round 4, measured and rejected: `sc1` instead of `nt` on the 16-byte stores (__builtin_amdgcn_raw_buffer_store_b128 with aux
// bit 4 over a per-tower descriptor: the lines then leave the XCD's L2 at once instead of staying dirty until the kernel ends)
// -- forward 99.2 -> 98.5 us, simulator step behind it 88.0 -> 89.9 us, iteration 42.25 -> 42.5 ms at equal traffic (two runs
// each).  (An inline-asm `global_store_dwordx4 ... sc1` is NOT an option: the hazard recogniser does not see a wide store in it
// and lets the next VALU instruction overwrite the data registers -- 7 % of the cached rows came out as garbage.)
__device__ __forceinline__ void st_stream(float *p, float v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void st_stream4(float *p, const float4 &v) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<f4 *>(p));
}

// ------------------------------------------------------------------------------------------------
// LSTM forward (agents/utils.py:88-116), T steps, one workgroup per (group, 64-env tile).
//   Z     [G][T*E][256]  in: x*Wx + b   out (if store): post-activation gates i|f|o|u
//   state [G][E][128]    c | h   (read; written back if write_state)
//   Hh,Cc [G][T*E][64]   h_t, c_t            (if store)
//   Hp    [G][T*E][64]   masked h_{t-1} fed to step t   (if store)
//   done  [T][E] u8      pre-step done (resets c, h)
// Wave w owns env rows 32*(w&1).. and hidden units 32*(w>>1)..; its four 32x32 MFMA tiles are
// the four gates of the same (env, unit) pairs, so the cell update needs no cross-lane traffic.
// ------------------------------------------------------------------------------------------------
constexpr int kWhLd = kG4 + 4;
constexpr int kHsLd = 64 + 4;

template <bool PF>   // PF: prefetch the next step's Z under the MFMAs (training, T > 1); the 1-step rollout
                      // variant keeps the register budget for 2 workgroups per CU
__global__ void __launch_bounds__(256, PF ? 1 : 2) lstm_fwd_kernel(const float *__restrict__ params, Layout lay, float *Z,
                                                      const float *state_in, float *state_out,
                                                      float *Hh, float *Cc, float *Hp, const uint8_t *done,
                                                      int T, int E, int store) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float *Whs = (float *)smem_raw;                 // [64][kWhLd]
    float *hs = Whs + 64 * kWhLd;                   // [64 k = unit][kHsLd m = env]
    const int g = blockIdx.x, e0 = blockIdx.y * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = 32 * (wave & 1), j0 = 32 * (wave >> 1), li = lane & 31, kh = lane >> 5;
    const float *Wh = params + (long long)g * lay.stride + lay.oWh;
    for (int i = tid; i < 64 * kG4; i += 256) Whs[(i / kG4) * kWhLd + (i % kG4)] = Wh[i];
    const long long N = (long long)T * E;
    const int j = j0 + li;
    float c[16];
    int erow[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        erow[r] = r0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        const int e = e0 + erow[r];
        float c0 = 0.f, h0 = 0.f;
        if (e < E) {
            const float *s = state_in + ((long long)g * E + e) * 2 * kL;
            c0 = s[j]; h0 = s[kL + j];
            const float keep = 1.0f - (float)done[e];
            c0 *= keep; h0 *= keep;
        }
        c[r] = c0;
        hs[j * kHsLd + erow[r]] = h0;
    }
    __syncthreads();
    f32x16 zn[4];                                   // x*Wx+b of the NEXT step, prefetched under the MFMAs
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int e = e0 + erow[r];
            zn[q][r] = e < E ? Z[(((long long)g * N + e) * kG4) + 64 * q + j] : 0.f;
        }
    for (int t = 0; t < T; ++t) {
        if (!PF && t > 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int e = e0 + erow[r];
                    zn[q][r] = e < E ? Z[(((long long)g * N + (long long)t * E + e) * kG4) + 64 * q + j] : 0.f;
                }
        }
        f32x16 acc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = zn[q];
        if (PF && t + 1 < T) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int e = e0 + erow[r];
                    zn[q][r] = e < E ? Z[(((long long)g * N + (long long)(t + 1) * E + e) * kG4) + 64 * q + j] : 0.f;
                }
        }
        if (store) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int e = e0 + erow[r];
                if (e < E) Hp[((long long)g * N + (long long)t * E + e) * kL + j] = hs[j * kHsLd + erow[r]];
            }
        }
#pragma unroll 4
        for (int kk = 0; kk < 64; kk += 2) {
            const float a = hs[(kk + kh) * kHsLd + r0 + li];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float b = Whs[(kk + kh) * kWhLd + 64 * q + j];
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q], 0, 0, 0);
            }
        }
        __syncthreads();                             // everyone is done reading hs
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int e = e0 + erow[r];
            const float ig = sigmoidf_(acc[0][r]), fg = sigmoidf_(acc[1][r]);
            const float og = sigmoidf_(acc[2][r]), ug = tanhf_(acc[3][r]);
            const float cn = fg * c[r] + ig * ug;
            const float hn = og * tanhf_(cn);
            float keep = 1.0f;
            if (e < E) {
                const long long n = (long long)g * N + (long long)t * E + e;
                if (store) {
                    Z[n * kG4 + j] = ig; Z[n * kG4 + 64 + j] = fg; Z[n * kG4 + 128 + j] = og; Z[n * kG4 + 192 + j] = ug;
                    Cc[n * kL + j] = cn;
                }
                Hh[n * kL + j] = hn;
                if (t + 1 < T) keep = 1.0f - (float)done[(long long)(t + 1) * E + e];
                if (t + 1 == T && state_out) {
                    float *s = state_out + ((long long)g * E + e) * 2 * kL;
                    s[j] = cn; s[kL + j] = hn;
                }
            }
            c[r] = cn * keep;
            hs[j * kHsLd + erow[r]] = hn * keep;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// LSTM backward through time (tf.gradients through the n_step-unrolled agents/utils.py:88-116).  One workgroup per
// (group, 32-env tile); dc and the recurrent dh of every (env, unit) pair stay on chip over the whole sequence.
//   Z   [G][N][256] in: gates i|f|o|u (post-activation)   out: dz (pre-activation gradients)
//   Cc  [G][N][64]  c_t;  c_{t-1} = Cc[t-1] or state_bw, masked by done[t]
//   dH  [G][N][64]  gradient arriving at h_t from the head
// per step: dz -> LDS (row-major) -> dh_{t-1} = dz * Wh^T on the MFMA (K = 256).  Wh^T is the STATIONARY operand: a wave
// owns 16 unit columns (v_mfma_f32_16x16x4_f32, two 16-instance row tiles = two independent accumulators), its slice of
// Wh^T is 64 registers for all T steps -- which leaves room for ALL of the next step's inputs: they are requested before
// the MFMA phase and land underneath it (round 2's 32-column strips held 128 weight registers and paid every step's
// load latency twice: MFMA busy 35 %, 4.2 ms).
// The gate math is elementwise, so it runs on its own thread mapping -- a thread owns ONE instance and EIGHT consecutive
// units (16-byte accesses, eight threads = one 256-byte row; on the MFMA output layout the same data were 48 dword loads per
// lane and step, 64-byte runs over four rows) -- and only the recurrent dh crosses mappings, through 8 KB of LDS between the
// two barriers the step has anyway.  c_t of a step is the c_{t-1} the step processed before it asked for: one cache array is
// read once, not twice.
// A step: gate math from the registers -> dz to HBM and LDS (read back as 16-byte A-operand quads) -> barrier
// -> request the inputs of step t - 1 -> 64 MFMAs (the loads land underneath) -> dh to LDS -> barrier.
// Measured (tools/bench_update.py, E = 1024, T = 120): 4.13 ms (round 2) -> 4.02 (16-column waves, dword accesses)
// -> 3.60 (c_t from registers) -> 3.39 ms: 2.56 KB per sample and tower (gates in, dz out, c_prev, dH) = 15.7 GB per update
// at 4.6 TB/s.  Three workgroups per CU (168 VGPRs, 4 spilled) measured the same (3.35 ms); 64-instance tiles spill.
// Round 6 (under rule 10's batches the kernel had drifted to 3.8 ms): the inputs are requested TWO steps ahead (two register sets):
// 3.82 -> 3.57 - 3.70 ms.  An eight-wavefront variant (512 threads, a thread owns four units, a wavefront half of the contraction: 64
// MFMAs per step, 16 wavefronts per CU) measured the same 3.62 ms and was removed: at 15.5 GB of mixed read / write traffic per
// launch the kernel sits at 4.3 TB/s whatever the per-step latency chain looks like.
// ------------------------------------------------------------------------------------------------
constexpr int kDz2Ld = kG4 + 4;   // dz tile [32 env][256 k + 4]
constexpr int kDh3Ld = kL + 4;    // recurrent dh tile [32 env][64 units + 4]

__global__ void __launch_bounds__(256, 2) lstm_bwd2_kernel(const float *__restrict__ params, Layout lay, float *Z,
                                                            const float *Cc, const float *state_bw, const float *dH,
                                                            const uint8_t *done, int T, int E) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float *dzs = (float *)smem_raw;                 // [32 env][kDz2Ld]
    float *dhs = dzs + 32 * kDz2Ld;                 // [32 env][kDh3Ld]
    const int g = blockIdx.x, e0 = blockIdx.y * 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, kq = lane >> 4;
    const int j = 16 * wave + n;                    // MFMA phase: this lane's hidden unit
    const int er = tid >> 3, u0 = (tid & 7) * 8;    // elementwise phase: instance row, first of 8 units
    const long long N = (long long)T * E;
    float bw[64];                                   // bw[4 q + c] = Wh[j][16 q + 4 kq + c] = B[k = 16 q + 4 kq + c][n]
    {
        const float4 *src = reinterpret_cast<const float4 *>(params + (long long)g * lay.stride + lay.oWh + (long long)j * kG4 + 4 * kq);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float4 w4 = src[4 * q];
            bw[4 * q] = w4.x; bw[4 * q + 1] = w4.y; bw[4 * q + 2] = w4.z; bw[4 * q + 3] = w4.w;
        }
    }
    float dc_rec[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) dc_rec[r] = 0.f;
    const float *zg = Z + (long long)g * N * kG4, *cg = Cc + (long long)g * N * kL, *hg = dH + (long long)g * N * kL;
    const float *sg = state_bw + (long long)g * E * 2 * kL;
    const int e = e0 + er < E ? e0 + er : E - 1;
    const bool live = e0 + er < E;
    // The inputs of a step are requested TWO steps ahead (round 6; one step ahead before: the request went out behind the step's
    // first barrier and had the 64 MFMAs of one step -- ~ 2 us with the co-resident workgroup's -- to land, less than a loaded HBM
    // round trip): two register sets, alternating by step parity.
    struct In { float4 gi[2], gf[2], go[2], gu[2], dhi[2], cpv[2]; float keep; };
    In in0, in1;
    float4 cc[2];                                   // c_t of the step about to be processed = the c_{t-1} the step after it asked for
    auto request = [&](int t, In &o) {
        const long long nt = (long long)t * E;
        const float *zt = zg + nt * kG4, *ct = cg + nt * kL, *ht = hg + nt * kL;
        const float *cp_base = t > 0 ? ct - (long long)E * kL : sg;         // c_{t-1}: Cc[t-1] ([e][64]) or the state ([e][128])
        const unsigned cp_ld = t > 0 ? kL : 2 * kL;
        int zq = 0;
        asm volatile("" : "+v"(zq));                // lane offsets are formed per step, not hoisted out of the t loop
        const unsigned oz = (unsigned)((e + zq) * kG4 + u0) * 4u, oc = (unsigned)((e + zq) * kL + u0) * 4u;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            o.gi[a] = ldg((const float4 *)zt, oz + 16u * a); o.gf[a] = ldg((const float4 *)zt, oz + 256u + 16u * a);
            o.go[a] = ldg((const float4 *)zt, oz + 512u + 16u * a); o.gu[a] = ldg((const float4 *)zt, oz + 768u + 16u * a);
            o.dhi[a] = ldg((const float4 *)ht, oc + 16u * a);
            o.cpv[a] = ldg((const float4 *)cp_base, (unsigned)((e + zq) * cp_ld + u0) * 4u + 16u * a);
        }
        o.keep = done[nt + e] == 0 ? 1.0f : 0.0f;
    };
    {   // c_{T-1} itself: nobody asked for it as a c_{t-1}
        const float *ct = cg + (long long)(T - 1) * E * kL;
        cc[0] = ldg((const float4 *)ct, (unsigned)(e * kL + u0) * 4u); cc[1] = ldg((const float4 *)ct, (unsigned)(e * kL + u0) * 4u + 16u);
    }
    request(T - 1, in0);
    if (T > 1) request(T - 2, in1);
    float keep_next = 0.f;                          // keep of the step processed before (t + 1)
    auto step = [&](int t, In &cur) {
        const long long nt = (long long)t * E;
        float *zw = Z + ((long long)g * N + nt) * kG4;
        const float keep = cur.keep;
        // recurrent dh of this thread's pairs: what the MFMA phase of step t + 1 left in LDS, cut where step t + 1 began an episode
        float4 dhr[2];
        if (t == T - 1) { dhr[0] = dhr[1] = make_float4(0.f, 0.f, 0.f, 0.f); }
        else {
            dhr[0] = *reinterpret_cast<const float4 *>(dhs + er * kDh3Ld + u0);
            dhr[1] = *reinterpret_cast<const float4 *>(dhs + er * kDh3Ld + u0 + 4);
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const float igv[4] = {cur.gi[a].x, cur.gi[a].y, cur.gi[a].z, cur.gi[a].w}, fgv[4] = {cur.gf[a].x, cur.gf[a].y, cur.gf[a].z, cur.gf[a].w};
            const float ogv[4] = {cur.go[a].x, cur.go[a].y, cur.go[a].z, cur.go[a].w}, ugv[4] = {cur.gu[a].x, cur.gu[a].y, cur.gu[a].z, cur.gu[a].w};
            const float ccv[4] = {cc[a].x, cc[a].y, cc[a].z, cc[a].w}, cpw[4] = {cur.cpv[a].x, cur.cpv[a].y, cur.cpv[a].z, cur.cpv[a].w};
            const float dhv[4] = {cur.dhi[a].x, cur.dhi[a].y, cur.dhi[a].z, cur.dhi[a].w}, drv[4] = {dhr[a].x, dhr[a].y, dhr[a].z, dhr[a].w};
            float di[4], df[4], dg[4], du[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int r = 4 * a + c;
                const float ig = igv[c], fg = fgv[c], og = ogv[c], ug = ugv[c];
                const float cp = cpw[c] * keep;
                const float dh = dhv[c] + (keep_next != 0.f ? drv[c] : 0.f);
                const float tc = tanhf_(ccv[c]);
                dg[c] = dh * tc * og * (1.0f - og);
                const float dc = dh * og * (1.0f - tc * tc) + dc_rec[r];
                di[c] = dc * ug * ig * (1.0f - ig);
                df[c] = dc * cp * fg * (1.0f - fg);
                du[c] = dc * ig * (1.0f - ug * ug);
                dc_rec[r] = dc * fg * keep;
            }
            const float4 o_di = make_float4(di[0], di[1], di[2], di[3]), o_df = make_float4(df[0], df[1], df[2], df[3]);
            const float4 o_do = make_float4(dg[0], dg[1], dg[2], dg[3]), o_du = make_float4(du[0], du[1], du[2], du[3]);
            if (live) {
                const unsigned oz = (unsigned)((e0 + er) * kG4 + u0 + 4 * a) * 4u;
                stg((float4 *)zw, oz, o_di); stg((float4 *)zw, oz + 256u, o_df);
                stg((float4 *)zw, oz + 512u, o_do); stg((float4 *)zw, oz + 768u, o_du);
            }
            float *row = dzs + er * kDz2Ld + u0 + 4 * a;
            *reinterpret_cast<float4 *>(row) = o_di; *reinterpret_cast<float4 *>(row + 64) = o_df;
            *reinterpret_cast<float4 *>(row + 128) = o_do; *reinterpret_cast<float4 *>(row + 192) = o_du;
            __builtin_amdgcn_sched_barrier(0);       // the two halves one after the other (registers)
        }
        cc[0] = cur.cpv[0]; cc[1] = cur.cpv[1];     // this step's c_{t-1} is the next step's c_t
        keep_next = keep;
        __syncthreads();
        if (t > 1) request(t - 2, cur);             // two steps ahead, into the set this step just freed
        f32x4 acc[2];
        acc[0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float4 *Aa = reinterpret_cast<const float4 *>(dzs + n * kDz2Ld + 4 * kq), *Ab = Aa + (16 * kDz2Ld) / 4;
        // two row tiles at a time (two independent accumulators cover the 40-cycle dependent latency); the next quads are
        // requested before this step's MFMAs and the iterations are kept apart, or all the LDS reads get hoisted
        float4 pa = Aa[0], pb = Ab[0];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float4 na = Aa[4 * (q + 1 < 16 ? q + 1 : q)], nb = Ab[4 * (q + 1 < 16 ? q + 1 : q)];
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa.x, bw[4 * q], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(pb.x, bw[4 * q], acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa.y, bw[4 * q + 1], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(pb.y, bw[4 * q + 1], acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa.z, bw[4 * q + 2], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(pb.z, bw[4 * q + 2], acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa.w, bw[4 * q + 3], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(pb.w, bw[4 * q + 3], acc[1], 0, 0, 0);
            pa = na; pb = nb;
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int i = 0; i < 4; ++i) dhs[(16 * rt + 4 * kq + i) * kDh3Ld + j] = acc[rt][i];
        __syncthreads();
    };
    for (int t = T - 1; t >= 0; t -= 2) {
        step(t, in0);
        if (t >= 1) step(t - 1, in1);
    }
}

// ------------------------------------------------------------------------------------------------
// Heads.  pi = softmax(h_pi Wo + bo) over the agent's n_a actions, v = h_v Wv + bv
// (agents/policies.py:20-26).  One thread per (sample, agent).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void head_eval(const float *__restrict__ params, const Layout &lay, int a, int na,
                                          const float *hp, const float *hv, float *pi, float &v) {
    const float *Wo = params + (long long)(2 * a) * lay.stride + lay.oWo;
    const float *bo = params + (long long)(2 * a) * lay.stride + lay.obo;
    const float *Wv = params + (long long)(2 * a + 1) * lay.stride + lay.oWo;
    const float *bv = params + (long long)(2 * a + 1) * lay.stride + lay.obo;
    float lg[kOut];
#pragma unroll
    for (int k = 0; k < kOut; ++k) lg[k] = 0.f;
    float vv = 0.f;
    for (int jj = 0; jj < kL; ++jj) {
        const float h = hp[jj];
#pragma unroll
        for (int k = 0; k < kOut; ++k) lg[k] += h * Wo[jj * kOut + k];
        vv += hv[jj] * Wv[jj * kOut];
    }
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < kOut; ++k) { lg[k] += bo[k]; if (k < na && lg[k] > mx) mx = lg[k]; }
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < kOut; ++k) { pi[k] = k < na ? expf(lg[k] - mx) : 0.f; sum += pi[k]; }
#pragma unroll
    for (int k = 0; k < kOut; ++k) pi[k] = pi[k] / sum;
    v = vv + bv[0];
}

// Tile helpers: one wave per (agent, 64 consecutive samples).  The [64 samples][64 units] slab of a
// tower is contiguous in HBM, so it is streamed with coalesced float4 accesses and transposed through
// LDS (row stride 65: conflict-free per-sample reads).
constexpr int kTLd = 65;
__device__ __forceinline__ void tile_load(const float *src, long long rows_left, float (*t)[kTLd], int lane) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int q = i * 64 + lane, r = q >> 4, c = (q & 15) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < rows_left) v = *reinterpret_cast<const float4 *>(src + (long long)r * kL + c);
        t[r][c] = v.x; t[r][c + 1] = v.y; t[r][c + 2] = v.z; t[r][c + 3] = v.w;
    }
}
__device__ __forceinline__ void tile_store(float *dst, long long rows_left, float (*t)[kTLd], int lane) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int q = i * 64 + lane, r = q >> 4, c = (q & 15) * 4;
        if (r < rows_left)
            *reinterpret_cast<float4 *>(dst + (long long)r * kL + c) = make_float4(t[r][c], t[r][c + 1], t[r][c + 2], t[r][c + 3]);
    }
}

__global__ void __launch_bounds__(64) head_fwd_kernel(const float *__restrict__ params, Layout lay, const int *n_act,
                                                      const float *Hh, int E, float *pi_out, float *v_out) {
    __shared__ float sp[64][kTLd], sv[64][kTLd];
    const int a = blockIdx.y, lane = threadIdx.x;
    const long long n0 = (long long)blockIdx.x * 64, left = E - n0;
    tile_load(Hh + ((long long)(2 * a) * E + n0) * kL, left, sp, lane);
    tile_load(Hh + ((long long)(2 * a + 1) * E + n0) * kL, left, sv, lane);
    __syncthreads();
    if (lane >= left) return;
    float pi[kOut], v;
    head_eval(params, lay, a, n_act[a], sp[lane], sv[lane], pi, v);
    const long long idx = (n0 + lane) * lay.A + a;
    for (int k = 0; k < lay.AMAX; ++k) pi_out[idx * lay.AMAX + k] = k < kOut ? pi[k] : 0.f;
    v_out[idx] = v;
}

// Loss (agents/policies.py:41-52) and its gradient w.r.t. logits / v, then back through the head:
//   L = -mean(log_pi[a] Adv) + 0.5 v_coef mean((R - v)^2) - beta mean(entropy), mean over the
//   T*E samples of one agent.  Writes dL [G][N][8] (pi tower: dlogits, v tower: dv in col 0)
//   and dH [G][N][64].


// ------------------------------------------------------------------------------------------------
// head_bwd, second form: loss gradient + dH + dWo / dbo of BOTH towers of an agent in one persistent pass.
// Thread = (sample, quad of 4 hidden units): the 16 lanes of a sample read its h row as one coalesced 16-byte load per
// lane (no LDS transpose, so nothing limits the waves per SIMD), reduce their partial logits with four xor-shuffles, all
// evaluate the softmax / loss gradient of the sample, write their quad of dH, and accumulate h^T dL (= dWo, dWv) and
// colsum(dL) (= dbo, dbv) in registers over the workgroup's fixed slice of the samples.  The slices' partial sums are
// added in slice order by head_bwd2_reduce_kernel (deterministic), which replaces the separate split-K dWo GEMM and
// its second pass over h and dL.  grid = (slices, agents), 256 threads.
// ------------------------------------------------------------------------------------------------
constexpr int kHbPer = kL * kOut + kOut + kL + 1;      // partial record of a slice: dWo [64][8] | dbo [8] | dWv [64] | dbv

__global__ void __launch_bounds__(256, 3)
head_bwd2_kernel(const float *__restrict__ params, Layout lay, const int *__restrict__ n_act, const float *__restrict__ Hh,
                 const int *__restrict__ act, const float *__restrict__ Rs, const float *__restrict__ Advs, long long N,
                 long long rows_per_slice, float v_coef, float beta, float *__restrict__ dH, float *__restrict__ part,
                 double *stats) {
    __shared__ float red[4][16][36];
    const int a = blockIdx.y, sp = blockIdx.x, na = n_act[a];
    const int tid = threadIdx.x, c = tid & 15, grp = tid >> 4;          // 16 sample groups per workgroup
    const float *Pp = params + (long long)(2 * a) * lay.stride, *Pv = params + (long long)(2 * a + 1) * lay.stride;
    float wo[4][kOut], wv[4], bo[kOut];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int k = 0; k < kOut; ++k) wo[u][k] = Pp[lay.oWo + (4 * c + u) * kOut + k];
        wv[u] = Pv[lay.oWo + (4 * c + u) * kOut];
    }
#pragma unroll
    for (int k = 0; k < kOut; ++k) bo[k] = Pp[lay.obo + k];
    const float bv = Pv[lay.obo];
    float aw[4][kOut], awv[4], ab[kOut], abv = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        awv[u] = 0.f;
#pragma unroll
        for (int k = 0; k < kOut; ++k) aw[u][k] = 0.f;
    }
#pragma unroll
    for (int k = 0; k < kOut; ++k) ab[k] = 0.f;
    float lp = 0.f, lv = 0.f, le = 0.f;
    const long long n0 = (long long)sp * rows_per_slice;
    long long n1 = n0 + rows_per_slice;
    if (n1 > N) n1 = N;
    const float *hp = Hh + (long long)(2 * a) * N * kL + 4 * c, *hq = Hh + (long long)(2 * a + 1) * N * kL + 4 * c;
    float *dp = dH + (long long)(2 * a) * N * kL + 4 * c, *dq = dH + (long long)(2 * a + 1) * N * kL + 4 * c;
    const float invN = 1.0f / (float)N;
    for (long long nb = n0; nb < n1; nb += 16) {
        const long long n = nb + grp;
        const bool in = n < n1;
        const long long nc = in ? n : n1 - 1;                           // clamped: unconditional loads
        const float4 h4 = *reinterpret_cast<const float4 *>(hp + nc * kL);
        const float4 g4 = *reinterpret_cast<const float4 *>(hq + nc * kL);
        const long long idx = nc * lay.A + a;
        const int ac = act[idx];
        const float adv = Advs[idx], R = Rs[idx];
        const float hh[4] = {h4.x, h4.y, h4.z, h4.w}, gg[4] = {g4.x, g4.y, g4.z, g4.w};
        float lg[kOut], vv = 0.f;
#pragma unroll
        for (int k = 0; k < kOut; ++k) lg[k] = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int k = 0; k < kOut; ++k) lg[k] += hh[u] * wo[u][k];
            vv += gg[u] * wv[u];
        }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {                              // the 16 lanes of a sample are one shuffle row
#pragma unroll
            for (int k = 0; k < kOut; ++k) lg[k] += __shfl_xor(lg[k], o, 64);
            vv += __shfl_xor(vv, o, 64);
        }
        const float v = vv + bv;
        // softmax / loss gradient: lane c of the sample's 16 handles action k = c & 7 (one exp, one log, one division per
        // lane instead of eight of each), sums go over the 8-lane row by xor-shuffles, then every lane collects all dl[k]
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < kOut; ++k) { lg[k] += bo[k]; if (k < na && lg[k] > mx) mx = lg[k]; }
        const int km = c & 7;
        float lgm = lg[0];
#pragma unroll
        for (int k = 1; k < kOut; ++k) lgm = km == k ? lg[k] : lgm;
        const bool valid = km < na;
        const float pk = valid ? expf(lgm - mx) : 0.f;
        float sum = pk;
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) sum += __shfl_xor(sum, o, 64);
        const float pim = pk / sum;
        const bool inr = pim >= 1e-10f;                                  // tf.clip_by_value(pi, 1e-10, 1)
        const float logpm = valid ? logf(fminf(fmaxf(pim, 1e-10f), 1.0f)) : 0.f;
        float gpi = 0.f;
        if (valid) {
            if (km == ac && inr) gpi += -adv * invN / fmaxf(pim, 1e-10f);
            gpi += beta * invN * (logpm + (inr ? 1.0f : 0.f));
        }
        float dot = pim * gpi, ent = valid ? -pim * logpm : 0.f, lpa = (km == (ac < na ? ac : 0)) ? logpm : 0.f;
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            dot += __shfl_xor(dot, o, 64); ent += __shfl_xor(ent, o, 64); lpa += __shfl_xor(lpa, o, 64);
        }
        const float dlm = (in && valid) ? pim * (gpi - dot) : 0.f;
        float dl[kOut];
        const int row0 = (tid & 63) & ~15;
#pragma unroll
        for (int k = 0; k < kOut; ++k) dl[k] = __shfl(dlm, row0 + k, 64);
        const float dv = in ? v_coef * (v - R) * invN : 0.f;
        // dH of my quad (FC policy: the head input is relu(.), fold its derivative in), dWo / dWv accumulation
        float o4[4], q4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float sacc = 0.f;
#pragma unroll
            for (int k = 0; k < kOut; ++k) { sacc += dl[k] * wo[u][k]; aw[u][k] += hh[u] * dl[k]; }
            o4[u] = (lay.fc && !(hh[u] > 0.f)) ? 0.f : sacc;
            q4[u] = (lay.fc && !(gg[u] > 0.f)) ? 0.f : dv * wv[u];
            awv[u] += gg[u] * dv;
        }
        if (in) {
            *reinterpret_cast<float4 *>(dp + n * kL) = make_float4(o4[0], o4[1], o4[2], o4[3]);
            *reinterpret_cast<float4 *>(dq + n * kL) = make_float4(q4[0], q4[1], q4[2], q4[3]);
        }
        if (c == 0) {
#pragma unroll
            for (int k = 0; k < kOut; ++k) ab[k] += dl[k];
            abv += dv;
            if (in) {
                lp += -lpa * adv * invN;
                lv += 0.5f * v_coef * (R - v) * (R - v) * invN;
                le += -beta * ent * invN;
            }
        }
    }
    // fold the 16 sample groups: across the wave's four groups by shuffles, across the four waves through LDS
    const int wave = tid >> 6;
#pragma unroll
    for (int o = 16; o < 64; o <<= 1) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int k = 0; k < kOut; ++k) aw[u][k] += __shfl_xor(aw[u][k], o, 64);
            awv[u] += __shfl_xor(awv[u], o, 64);
        }
#pragma unroll
        for (int k = 0; k < kOut; ++k) ab[k] += __shfl_xor(ab[k], o, 64);
        abv += __shfl_xor(abv, o, 64);
        lp += __shfl_xor(lp, o, 64); lv += __shfl_xor(lv, o, 64); le += __shfl_xor(le, o, 64);
    }
    if ((tid & 63) < 16) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int k = 0; k < kOut; ++k) red[wave][c][u * kOut + k] = aw[u][k];
            red[wave][c][32 + u] = awv[u];
        }
    }
    __shared__ float redb[4][kOut + 1];
    __shared__ float redl[4][3];
    if ((tid & 63) == 0) {
#pragma unroll
        for (int k = 0; k < kOut; ++k) redb[wave][k] = ab[k];
        redb[wave][kOut] = abv;
        redl[wave][0] = lp; redl[wave][1] = lv; redl[wave][2] = le;
    }
    __syncthreads();
    float *out = part + ((long long)sp * lay.A + a) * kHbPer;
    for (int j = tid; j < kHbPer; j += 256) {
        float acc = 0.f;
        if (j < kL * kOut) {                       // dWo[jj][k]: jj = 4 c + u
            const int jj = j / kOut, k = j % kOut;
            for (int w = 0; w < 4; ++w) acc += red[w][jj >> 2][(jj & 3) * kOut + k];
        } else if (j < kL * kOut + kOut) {
            for (int w = 0; w < 4; ++w) acc += redb[w][j - kL * kOut];
        } else if (j < kL * kOut + kOut + kL) {
            const int jj = j - (kL * kOut + kOut);
            for (int w = 0; w < 4; ++w) acc += red[w][jj >> 2][32 + (jj & 3)];
        } else {
            for (int w = 0; w < 4; ++w) acc += redb[w][kOut];
        }
        out[j] = acc;
    }
    if (stats && tid == 0) {     // logging only (policies.py:63-72)
        double a0 = 0, a1 = 0, a2 = 0;
        for (int w = 0; w < 4; ++w) { a0 += redl[w][0]; a1 += redl[w][1]; a2 += redl[w][2]; }
        atomicAdd(&stats[a * 4 + 0], a0); atomicAdd(&stats[a * 4 + 1], a1); atomicAdd(&stats[a * 4 + 2], a2);
    }
}

// grads[g][oWo .. obo + 8) of both towers of every agent = sum over slices, in slice order
__global__ void head_bwd2_reduce_kernel(const float *__restrict__ part, int A, int S, float *__restrict__ grads, Layout lay) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int per = kL * kOut + kOut;                 // entries per tower: Wo [64][8] | bo [8]
    if (i >= A * 2 * per) return;
    const int a = i / (2 * per), r = i % (2 * per), tower = r / per, j = r % per;
    int src = -1;                                      // index into the slice record, -1: structurally zero
    if (tower == 0) src = j;                           // dWo | dbo
    else if (j < kL * kOut) { if (j % kOut == 0) src = kL * kOut + kOut + j / kOut; }     // dWv in column 0
    else if (j == kL * kOut) src = kL * kOut + kOut + kL;                                 // dbv
    float acc = 0.f;
    if (src >= 0)
        for (int s2 = 0; s2 < S; ++s2) acc += part[((long long)s2 * A + a) * kHbPer + src];
    grads[(long long)(2 * a + tower) * lay.stride + lay.oWo + j] = acc;
}

// np.random.choice(n, p=pi): cdf = cumsum(p); cdf /= cdf[-1]; searchsorted(cdf, u, 'right')
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ int sample_action(const float *pi, int na, unsigned long long seed, unsigned long long step, long long idx) {
    const unsigned long long h = splitmix64(splitmix64(seed ^ (step * 0xD1B54A32D192ED03ull)) + (unsigned long long)idx);
    const double u = (double)(h >> 11) * (1.0 / 9007199254740992.0);
    double cdf[kOut], sacc = 0.0;
    for (int k = 0; k < na; ++k) { sacc += (double)pi[k]; cdf[k] = sacc; }
    int ans = na - 1;
    for (int k = 0; k < na; ++k)
        if (u < cdf[k] / sacc) { ans = k; break; }
    return ans;
}

// ------------------------------------------------------------------------------------------------
// FcACPolicy rollout forward in ONE launch (agents/policies.py:214-240; BASELINE configs[1]: IA2C FC, 256 instances):
//   obs -> relu(obs W1 + b1) -> relu(. Wfc + bfc) -> softmax head / value head -> sampled action.
// One workgroup per (agent, 64 instances): waves 0-3 evaluate the pi tower, waves 4-7 the V tower; a thread is an
// instance, a wave a block of output columns, so every weight is wave-uniform (scalar loads, four / eight at a time)
// and an activation is read from LDS once per four / eight multiply-adds.  The stateless policy needs no activation
// cache: its update re-evaluates the forward at training shape.  51 us per launch at E = 256 (round 2's four launches per
// control step -- two grouped GEMMs, heads, sampling -- took 85 us); the 100 workgroups are a latency chain each (quad LDS
// reads with all 16 units per pass measured slower: 57 us).  Since the MFMA formulation below (22.7 us) this kernel serves first
// layers that are not a multiple of 32 columns wide, and TSC_FC_MFMA=0.  The two kernels are NOT bit-identical: this one sums the 64
// hidden units of a head sequentially (head_eval), the MFMA kernel as 16 four-unit partial sums folded by an xor-shuffle tree; logits,
// pi and v agree to float32 rounding and a sampled action can differ where the uniform sits within that rounding of a boundary of
// the cumulative distribution (tests/test_model_gpu.py::test_fc_policy_forward_kernels_agree_with_each_other).
// ------------------------------------------------------------------------------------------------
constexpr int kFcLdo = 64 + 1;
__global__ void __launch_bounds__(512) policy_fwd_fc_kernel(const float *__restrict__ params, Layout lay, const int *__restrict__ n_act,
                                                           const float *__restrict__ obs, int E, float *__restrict__ pi_out,
                                                           float *__restrict__ v_out, int *action_out, unsigned long long seed,
                                                           unsigned long long step) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int H = lay.H, SMAX = lay.SMAX, ldx = H + 1;
    float *ob = (float *)smem_raw;                      // [64][kFcLdo]
    float *x1 = ob + 64 * kFcLdo;                       // [2 towers][64][H + 1]
    float *x2 = x1 + 2 * 64 * ldx;                      // [2 towers][64][kFcLdo]
    const int a = blockIdx.y, e0 = blockIdx.x * 64, tid = threadIdx.x;
    const int tower = __builtin_amdgcn_readfirstlane(tid >> 8), e = tid & 63;      // wave-uniform: the weight pointers live in SGPRs
    const int wv = __builtin_amdgcn_readfirstlane((tid >> 6) & 3);
    for (int i = tid; i < 64 * SMAX; i += 512) {
        const int r = i / SMAX, c = i % SMAX;
        const int er = e0 + r < E ? e0 + r : E - 1;
        ob[r * kFcLdo + c] = obs[((long long)er * lay.A + a) * SMAX + c];
    }
    __syncthreads();
    const float *P = params + (long long)(2 * a + tower) * lay.stride;
    {   // layer 1: this wave's H / 4 columns, four at a time
        const float *W1 = P + lay.oW1, *b1 = P + lay.ob1;
        const float *orow = ob + e * kFcLdo;
        float *xrow = x1 + (tower * 64 + e) * ldx;
        const int c0 = wv * (H / 4), c1 = c0 + H / 4;
        for (int c = c0; c < c1; c += 4) {
            float acc0 = b1[c], acc1 = b1[c + 1], acc2 = b1[c + 2], acc3 = b1[c + 3];
            // the weights are wave-uniform (scalar loads): twelve rows are requested before the first is used
            for (int s0 = 0; s0 < SMAX; s0 += 12) {
                float w[12][4];
#pragma unroll
                for (int q = 0; q < 12; ++q) {
                    const float *wp = W1 + (long long)(s0 + q < SMAX ? s0 + q : SMAX - 1) * H + c;
                    w[q][0] = wp[0]; w[q][1] = wp[1]; w[q][2] = wp[2]; w[q][3] = wp[3];
                }
#pragma unroll
                for (int q = 0; q < 12; ++q) {          // (-ffp-contract=off: the multiply-adds are explicit)
                    const float o = s0 + q < SMAX ? orow[s0 + q] : 0.f;
                    acc0 = fmaf(o, w[q][0], acc0); acc1 = fmaf(o, w[q][1], acc1); acc2 = fmaf(o, w[q][2], acc2); acc3 = fmaf(o, w[q][3], acc3);
                }
            }
            xrow[c] = fmaxf(acc0, 0.f); xrow[c + 1] = fmaxf(acc1, 0.f); xrow[c + 2] = fmaxf(acc2, 0.f); xrow[c + 3] = fmaxf(acc3, 0.f);
        }
    }
    __syncthreads();
    {   // layer 2: this wave's 16 of the 64 units, eight at a time
        const float *W2 = P + lay.oWx, *b2 = P + lay.obl;
        const float *xrow = x1 + (tower * 64 + e) * ldx;
        float *yrow = x2 + (tower * 64 + e) * kFcLdo;
#pragma unroll 1
        for (int j = 16 * wv; j < 16 * wv + 16; j += 8) {
            float acc[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q] = b2[j + q];
            for (int c0 = 0; c0 < H; c0 += 8) {                  // H % 8 == 0; eight rows of eight weights in flight
                float w[8][8];
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const float *wp = W2 + (long long)(c0 + r) * kL + j;
#pragma unroll
                    for (int q = 0; q < 8; ++q) w[r][q] = wp[q];
                }
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const float x = xrow[c0 + r];
#pragma unroll
                    for (int q = 0; q < 8; ++q) acc[q] = fmaf(x, w[r][q], acc[q]);
                }
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) yrow[j + q] = fmaxf(acc[q], 0.f);
        }
    }
    __syncthreads();
    if (tid < 64 && e0 + tid < E) {                     // heads + sampling, one instance per thread
        float pi[kOut], v;
        const int na = n_act[a];
        head_eval(params, lay, a, na, x2 + tid * kFcLdo, x2 + (64 + tid) * kFcLdo, pi, v);
        const long long idx = (long long)(e0 + tid) * lay.A + a;
        for (int k = 0; k < lay.AMAX; ++k) pi_out[idx * lay.AMAX + k] = k < kOut ? pi[k] : 0.f;
        v_out[idx] = v;
        if (action_out) action_out[idx] = sample_action(pi, na, seed, step, idx);
    }
}

// ------------------------------------------------------------------------------------------------
// The same forward on the matrix cores (round 3).  policy_fwd_fc_kernel above is a latency chain of ~2 x 10^4 dependent
// multiply-adds per thread: 51 us per launch whatever E is.  Here one workgroup takes (agent, 32 instances, both
// towers): layer 1 is 2 H/32 column tiles of [32 x SMAX] x [SMAX x 32] (v_mfma_f32_32x32x2_f32, SMAX / 2 steps each,
// obs k-major in LDS, W1 straight from L2 into the B operand), layer 2 is 4 column tiles of [32 x H] x [H x 32] with
// the contraction cut in two halves (eight waves, H / 4 dependent MFMAs each; the halves meet in LDS).  Every weight a
// wave needs is requested before the first barrier, so the launch costs two L2 round trips plus ~4k cycles of MFMA chains.
// ------------------------------------------------------------------------------------------------
constexpr int kFmLd = 33;        // [k or row][32 instances / columns + 1]

template <int NCT>               // H / 32
__global__ void __launch_bounds__(512, 1)
policy_fwd_fc_mfma_kernel(const float *__restrict__ params, Layout lay, const int *__restrict__ n_act,
                          const float *__restrict__ obs, int E, float *__restrict__ pi_out, float *__restrict__ v_out,
                          int *action_out, unsigned long long seed, unsigned long long step, int tslot, long long Ntot,
                          float *__restrict__ X1c, float *__restrict__ Hhc) {
    constexpr int H = 32 * NCT, LDX = H + 1, NS2 = H / 4, NU = 2 * NCT;
    static_assert(NU <= 16, "two layer-1 units per wave at most");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float *Os = (float *)smem_raw;                      // [64 k][kFmLd]: obs tile, k-major
    float *X1s = Os + 64 * kFmLd;                       // [2 towers][32][LDX]
    float *P2 = X1s + 2 * 32 * LDX;                     // [8 waves][32][kFmLd]: layer-2 partial tiles (one K half each)
    float *X2s = P2 + 8 * 32 * kFmLd;                   // [2 towers][32][kFcLdo]
    const int a = blockIdx.y, e0 = blockIdx.x * 32, tid = threadIdx.x, lane = tid & 63, li = lane & 31, kh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int SMAX = lay.SMAX, KS1 = (SMAX + 1) >> 1;
    const float *Pa = params + (long long)(2 * a) * lay.stride;
    // ---- every global operand is requested up front
    // layer 2: wave -> (tower, column tile, K half); B[k][n] = W2[k][32 ct + n]
    const int t2 = wave >> 2, ct2 = (wave >> 1) & 1, kh2 = wave & 1;
    float w2[NS2];
    {
        const float *W2 = Pa + (long long)t2 * lay.stride + lay.oWx + (long long)(kh2 * (H / 2) + kh) * kL + 32 * ct2 + li;
#pragma unroll
        for (int s = 0; s < NS2; ++s) w2[s] = W2[(long long)(2 * s) * kL];
    }
    // layer 1: unit u = tower * NCT + column tile; wave w owns units w and w + 8
    float w1[2][32], b1v[2];
#pragma unroll
    for (int uu = 0; uu < 2; ++uu) {
        const int u = wave + 8 * uu < NU ? wave + 8 * uu : NU - 1;
        const float *P = Pa + (long long)(u / NCT) * lay.stride;
        const int col = 32 * (u % NCT) + li;
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            const int k = 2 * s + kh < SMAX ? 2 * s + kh : SMAX - 1;
            w1[uu][s] = P[lay.oW1 + (long long)k * H + col];
        }
        b1v[uu] = P[lay.ob1 + col];
    }
    // heads: thread = (instance tid >> 4, units 4 (tid & 15) .. + 3) -- its slice of Wo (pi tower) and of Wv, requested with the rest
    const int hc = tid & 15;
    float hwo[4][kOut], hwv[4], hbo[kOut];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int k = 0; k < kOut; ++k) hwo[u][k] = Pa[lay.oWo + (4 * hc + u) * kOut + k];
        hwv[u] = Pa[lay.stride + lay.oWo + (4 * hc + u) * kOut];
    }
#pragma unroll
    for (int k = 0; k < kOut; ++k) hbo[k] = Pa[lay.obo + k];
    const float hbv = Pa[lay.stride + lay.obo];
    const float b2p = Pa[lay.obl + (tid & 63)], b2v = Pa[lay.stride + lay.obl + (tid & 63)];   // second-layer biases of this thread's column
    {   // obs tile -> LDS, k-major: all (<= 4, SMAX <= 64) loads of a thread first, then the stores (a load inside the store loop
        // costs one L2 round trip per iteration)
        float ovv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = tid + 512 * q, ic = i < 32 * SMAX ? i : 32 * SMAX - 1;
            const int r = ic / SMAX, c = ic - r * SMAX;
            const int er = e0 + r < E ? e0 + r : E - 1;
            ovv[q] = obs[((long long)er * lay.A + a) * SMAX + c];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = tid + 512 * q;
            const int r = i / SMAX, c = i - r * SMAX;
            if (i < 32 * SMAX) Os[c * kFmLd + r] = ovv[q];
        }
    }
    if (SMAX & 1) { if (tid < 32) Os[SMAX * kFmLd + tid] = 0.f; }      // the odd last step multiplies a zero row
    __syncthreads();
#pragma unroll
    for (int uu = 0; uu < 2; ++uu) {
        const int u = wave + 8 * uu;
        if (u < NU) {                                                   // wave-uniform
            const int tower = u / NCT, ct = u % NCT;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            // (one range test per step: every MFMA of this dependent 64-cycle chain sits in a block of its own behind its LDS
            //  operand, whose latency the previous MFMA covers -- grouping eight steps behind one test was measured slower, round 5:
            //  19.3 -> 20.4 us, six wasted steps at SMAX = 36)
#pragma unroll
            for (int s = 0; s < 32; ++s)
                if (s < KS1) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Os[(2 * s + kh) * kFmLd + li], w1[uu][s], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;
                X1s[(tower * 32 + row) * LDX + 32 * ct + li] = fmaxf(acc[r] + b1v[uu], 0.f);
            }
        }
    }
    __syncthreads();
    {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float *xr = X1s + (t2 * 32 + li) * LDX + kh2 * (H / 2) + kh;
#pragma unroll
        for (int s = 0; s < NS2; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xr[2 * s], w2[s], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) P2[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh) * kFmLd + li] = acc[r];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 2 * 32 * kL / 512; ++q) {                       // the two K halves + bias, relu (j = tid & 63 in every iteration)
        const int i = tid + 512 * q;
        const int t = i >> 11, row = (i >> 6) & 31, j = i & 63, w0 = t * 4 + (j >> 5) * 2, c = j & 31;
        const float v = (P2[(w0 * 32 + row) * kFmLd + c] + P2[((w0 + 1) * 32 + row) * kFmLd + c]) + (t ? b2v : b2p);
        X2s[(t * 32 + row) * kFcLdo + j] = fmaxf(v, 0.f);
    }
    __syncthreads();
    if (tslot >= 0) {
        // activation cache (like the LSTM rollout forward's): this step's X1 and relu(X1 Wfc + bfc) rows of both towers go to
        // rows [tslot E, (tslot + 1) E) of the training buffers, so the update does not evaluate the forward a second time
        // (the reference builds the graph twice, agents/policies.py:94-96).  Written once, read by the update only: streaming stores.
        constexpr int XQ = H / 4;
        for (int i = tid; i < 2 * 32 * XQ; i += 512) {
            const int t = i / (32 * XQ), row = (i / XQ) % 32, q = i % XQ;
            if (e0 + row < E) {
                const float *src = X1s + (t * 32 + row) * LDX + 4 * q;
                st_stream4(X1c + ((long long)(2 * a + t) * Ntot + (long long)tslot * E + e0 + row) * H + 4 * q, make_float4(src[0], src[1], src[2], src[3]));
            }
        }
        for (int i = tid; i < 2 * 32 * (kL / 4); i += 512) {
            const int t = i / (32 * (kL / 4)), row = (i / (kL / 4)) % 32, q = i % (kL / 4);
            if (e0 + row < E) {
                const float *src = X2s + (t * 32 + row) * kFcLdo + 4 * q;
                st_stream4(Hhc + ((long long)(2 * a + t) * Ntot + (long long)tslot * E + e0 + row) * kL + 4 * q, make_float4(src[0], src[1], src[2], src[3]));
            }
        }
    }
    {   // heads + sampling: the 16 lanes of an instance each take four hidden units of both towers, the partial logits meet by
        // xor-shuffles (one shuffle row of the wavefront per instance), lane 0 of the row finishes (round 4; one thread per
        // instance walked all 64 units before: a 600-instruction chain on 32 of the 512 threads)
        const int hr = tid >> 4;
        const float *xp = X2s + hr * kFcLdo + 4 * hc, *xv = X2s + (32 + hr) * kFcLdo + 4 * hc;
        float lg[kOut], vv = 0.f;
#pragma unroll
        for (int k = 0; k < kOut; ++k) lg[k] = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float h = xp[u];
#pragma unroll
            for (int k = 0; k < kOut; ++k) lg[k] += h * hwo[u][k];
            vv += xv[u] * hwv[u];
        }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
#pragma unroll
            for (int k = 0; k < kOut; ++k) lg[k] += __shfl_xor(lg[k], o, 64);
            vv += __shfl_xor(vv, o, 64);
        }
        // softmax + np.random.choice, one action per lane of the instance's row (lane k < 8 of the 16): one exp, one division and one
        // float64 division per lane instead of eight of each on one lane; every sum keeps its sequential order (the row gathers
        // the eight terms by shuffles and adds them left to right), so pi and the sampled action are what the one-lane code gave
        const int na = n_act[a], km = hc & 7, row0 = (tid & 63) & ~15;
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < kOut; ++k) { lg[k] += hbo[k]; if (k < na && lg[k] > mx) mx = lg[k]; }
        float lgm = lg[0];
#pragma unroll
        for (int k = 1; k < kOut; ++k) lgm = km == k ? lg[k] : lgm;
        const float pk = km < na ? expf(lgm - mx) : 0.f;
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < kOut; ++k) sum += __shfl(pk, row0 + k, 64);
        const float pim = pk / sum;
        double sacc = 0.0, mine = 0.0;                   // cdf = cumsum(p) in float64, left to right (numpy)
#pragma unroll
        for (int k = 0; k < kOut; ++k) {
            const float pj = __shfl(pim, row0 + k, 64);
            if (k < na) sacc += (double)pj;
            if (k == km) mine = sacc;
        }
        const long long idx = (long long)(e0 + hr) * lay.A + a;
        const unsigned long long hh = splitmix64(splitmix64(seed ^ (step * 0xD1B54A32D192ED03ull)) + (unsigned long long)idx);
        const double uu = (double)(hh >> 11) * (1.0 / 9007199254740992.0);
        const bool below = hc < 8 && km < na && uu < mine / sacc;          // searchsorted(cdf / cdf[-1], u, 'right'): first k with u < cdf_k
        const unsigned long long bal = __ballot(below);
        const unsigned rowbits = (unsigned)((bal >> row0) & 0xFFull);
        const int ans = rowbits ? __ffs((int)rowbits) - 1 : na - 1;
        if (e0 + hr < E) {
            if (hc < 8 && km < lay.AMAX) pi_out[idx * lay.AMAX + km] = pim;
            if (hc == 0) {
                v_out[idx] = vv + hbv;
                if (action_out) action_out[idx] = ans;
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------
// Fused rollout forward (one control step, IA2C.forward agents/models.py:185-200): for one agent-tower
// and a tile of 64 env instances, in ONE launch:
//   obs -> relu(obs W1 + b1) -> [x | h] [Wx ; Wh] + b -> LSTM cell -> head (softmax / value).
// Activations never leave the CU: obs tile, X1 and h live k-major in LDS (XH[(H+64)][68]); the
// weights are streamed from L2 straight into MFMA B operands (Wx and Wh are contiguous in the
// parameter layout, so the gate GEMM is one K = H+64 loop), double-buffered in registers 8 k-steps
// ahead.  Blocks are numbered so that all tiles of a tower run on the same XCD (block b -> XCD b%8)
// and reuse its L2-resident weights.
// ------------------------------------------------------------------------------------------------
constexpr int kXLd = 68;

__global__ void __launch_bounds__(256, 2)
policy_fwd_fused_kernel(const float *__restrict__ params, Layout lay, const int *__restrict__ n_act,
                        const float *__restrict__ obs, const uint8_t *__restrict__ done, float *state, int advance,
                        int E, int n_tiles, float *__restrict__ pi_out, float *__restrict__ v_out, int *action_out,
                        unsigned long long seed, unsigned long long step, long long *dbg,
                        // activation cache for the update (slot tslot of the n_step batch; tslot < 0: off)
                        int tslot, long long Ntot, float *X1c, float *Zc, float *Hhc, float *Ccc, float *Hpc,
                        const float *__restrict__ Wg) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float *XH = (float *)smem_raw;
    const bool stamp = dbg && blockIdx.x == 8 && threadIdx.x == 0;      // a block that does real work
    int nstamp = 0;
#define FSTAMP() do { if (stamp) dbg[nstamp++] = clock64(); } while (0)
    if (dbg && threadIdx.x == 0) dbg[64 + 2 * blockIdx.x] = wall_clock64();
    FSTAMP();
    const int b = blockIdx.x, xcd = b & 7, sidx = b >> 3;
    const int g = xcd + 8 * (sidx / n_tiles), tile = sidx % n_tiles;
    if (g >= lay.G) return;
    const int a = g >> 1, tower = g & 1, e0 = tile * 64, H = lay.H, SMAX = lay.SMAX;
    float *Hs = XH + H * kXLd;                      // rows H..H+63: obs staging, then h
    const float *P = params + (long long)g * lay.stride;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, kh = lane >> 5;

    const int r0 = 32 * (wave & 1), j0 = 32 * (wave >> 1), j = j0 + li;
    // Everything this workgroup needs from memory before its first MFMA is requested up front, in the order of
    // use and UNCONDITIONALLY (indices clamped; rows past E are computed but never stored), so that each phase
    // waits with a counted vmcnt for exactly its own operands: obs tile -> first W1 column tile -> head weights
    // -> LSTM state.  (Loads under `if (e < E)` make every wait a full drain: the obs tile then also waited for
    // the 32 state loads issued before it.)
    const int q4 = SMAX >> 2, AS = lay.A * SMAX, nct = H >> 5;
    const float *W1 = P + lay.oW1, *b1 = P + lay.ob1;
    float4 ov[4];
    int om[4], ok4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int idx = tid + 256 * q;
        om[q] = -1;
        if (idx < 64 * q4) om[q] = idx / q4; else idx = 64 * q4 - 1;
        const int m = idx / q4;
        ok4[q] = (idx % q4) * 4;
        const int e = e0 + m < E ? e0 + m : E - 1;
        ov[q] = *reinterpret_cast<const float4 *>(obs + (long long)e * AS + a * SMAX + ok4[q]);
    }
    float bw0[32];
    {
        const int ct = wave < nct ? wave : nct - 1;
        const unsigned wb = (unsigned)(ct * 32 + li) * 4u;
#pragma unroll
        for (int s2 = 0; s2 < 32; ++s2) {
            const int kr = 2 * s2 + kh < SMAX ? 2 * s2 + kh : SMAX - 1;       // rows past SMAX meet a zeroed A operand
            bw0[s2] = ldg(W1, wb + (unsigned)(kr * H) * 4u);
        }
    }
    float *WoS = XH + (H + 64) * kXLd;              // [64][8] head weights + [8] bias
    if (tid < (kL * kOut + kOut) / 4)
        *reinterpret_cast<float4 *>(WoS + 4 * tid) = *reinterpret_cast<const float4 *>(P + lay.oWo + 4 * tid);
    // LSTM state of this wave's 16 (env, unit) pairs: lands under phases 0-1
    float c[16], h0v[16];
    int erow[16];
    {
        const float *stb = state + (long long)g * E * 2 * kL;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            erow[r] = r0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            const int e = e0 + erow[r] < E ? e0 + erow[r] : E - 1;
            const float keep = 1.0f - (float)done[e];
            const unsigned ob = (unsigned)(e * 2 * kL + j) * 4u;
            c[r] = ldg(stb, ob) * keep; h0v[r] = ldg(stb, ob + kL * 4u) * keep;
        }
    }
    // ---- phase 0: obs tile -> LDS, k-major; rows [SMAX, 64) of the staging area are zero
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (om[q] >= 0) {
            Hs[(ok4[q] + 0) * kXLd + om[q]] = ov[q].x; Hs[(ok4[q] + 1) * kXLd + om[q]] = ov[q].y;
            Hs[(ok4[q] + 2) * kXLd + om[q]] = ov[q].z; Hs[(ok4[q] + 3) * kXLd + om[q]] = ov[q].w;
        }
    for (int idx = tid; idx < (64 - SMAX) * 64; idx += 256) Hs[(SMAX + (idx >> 6)) * kXLd + (idx & 63)] = 0.f;
    __syncthreads();
    FSTAMP();
    // ---- phase 1: X1 = relu(obs W1 + b1) -> XH rows [0, H); column tiles wave and wave + 4
    {
        float bw1[32];
        {
            const int ct = wave + 4 < nct ? wave + 4 : nct - 1;
            const unsigned wb = (unsigned)(ct * 32 + li) * 4u;
#pragma unroll
            for (int s2 = 0; s2 < 32; ++s2) {
                const int kr = 2 * s2 + kh < SMAX ? 2 * s2 + kh : SMAX - 1;
                bw1[s2] = ldg(W1, wb + (unsigned)(kr * H) * 4u);
            }
        }
        auto tile = [&](const float *bw, int ct) {
            f32x16 acc0, acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
            const int col = ct * 32 + li;
#pragma unroll
            for (int s2 = 0; s2 < 32; ++s2) {
                const float a0 = Hs[(2 * s2 + kh) * kXLd + li], a1 = Hs[(2 * s2 + kh) * kXLd + 32 + li];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bw[s2], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bw[s2], acc1, 0, 0, 0);
            }
            const float bias = b1[col];
            int eb1 = e0;
            asm volatile("" : "+v"(eb1));                   // form the store addresses here, not at kernel entry
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;
                float v0 = acc0[r] + bias, v1 = acc1[r] + bias;
                v0 = v0 > 0.f ? v0 : 0.f; v1 = v1 > 0.f ? v1 : 0.f;
                XH[col * kXLd + row] = v0;
                XH[col * kXLd + 32 + row] = v1;
                if (tslot >= 0) {                                  // X1 of this step, in the training layout [g][n][H]
                    const long long nb = (long long)g * Ntot + (long long)tslot * E + eb1;
                    if (eb1 + row < E) st_stream(X1c + ((nb + row) * H + col), v0);
                    if (eb1 + 32 + row < E) st_stream(X1c + ((nb + 32 + row) * H + col), v1);
                }
            }
        };
        if (wave < nct) tile(bw0, wave);
        if (wave + 4 < nct) tile(bw1, wave + 4);
    }
    __syncthreads();
    FSTAMP();
    // ---- phase 1.5: h (loaded at kernel entry, done-masked) -> LDS rows [H, H+64)
    int eb15 = e0;
    asm volatile("" : "+v"(eb15));
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        Hs[j * kXLd + erow[r]] = h0v[r];
        if (tslot >= 0 && eb15 + erow[r] < E)            // masked h_{t-1}: the operand of dWh in the update
            st_stream(Hpc + (((long long)g * Ntot + (long long)tslot * E + eb15 + erow[r]) * kL + j), h0v[r]);
    }
    __syncthreads();
    FSTAMP();
    // ---- phase 2: gates = bl + [X1 | h] [Wx ; Wh]   (K = H + 64)
    f32x16 acc[4];
    {
        const float *bl = P + lay.obl;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float bv = bl[64 * q + j];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[q][r] = bv;
        }
        // B = [Wx ; Wh] in the gate-interleaved copy Wg[g][H + 64][64 units][4 gates]: one 16-byte load per k row
        const float *B = Wg + (long long)g * (H + 64) * kG4;
        constexpr int CS = 8;                          // k-steps (2 k rows each) per register chunk
        const int nchunk = (H + 64) / (2 * CS);        // even: H % 32 == 0
        const unsigned bb = (unsigned)(kh * kG4 + 4 * j) * 4u;
        float bA[CS][4], bB[CS][4];
        // unconditional loads only (the last prefetch re-reads the last chunk): the waits stay counted
        auto loadB = [&](float (*dst)[4], int chunk) {
            const float *src = B + (long long)chunk * (2 * CS * kG4);       // uniform
#pragma unroll
            for (int s2 = 0; s2 < CS; ++s2) {
                const float4 w4 = ldg((const float4 *)src, bb + s2 * (2u * kG4 * 4u));
                dst[s2][0] = w4.x; dst[s2][1] = w4.y; dst[s2][2] = w4.z; dst[s2][3] = w4.w;
            }
        };
        auto compute = [&](float (*bv)[4], int chunk) {
            const float *As = XH + (chunk * 2 * CS + kh) * kXLd + r0 + li;
#pragma unroll
            for (int s2 = 0; s2 < CS; ++s2) {
                const float av = As[(2 * s2) * kXLd];
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[s2][q], acc[q], 0, 0, 0);
            }
        };
        loadB(bA, 0);
        for (int ch = 0; ch < nchunk; ch += 2) {
            loadB(bB, ch + 1);
            compute(bA, ch);
            loadB(bA, ch + 2 < nchunk ? ch + 2 : nchunk - 1);
            compute(bB, ch + 1);
        }
    }
    FSTAMP();
    __syncthreads();                                   // everyone is done reading XH
    FSTAMP();
    // ---- phase 3: cell update, state write-back, h -> LDS rows [0, 64)
    // (the row offset goes through an opaque register so that the ~100 store addresses of this phase are formed
    //  here and not hoisted to kernel entry, where they cost 200 B/lane of scratch spills)
    int row_base = e0;
    asm volatile("" : "+v"(row_base));
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int e = row_base + erow[r];
        const float ig = sigmoidf_(acc[0][r]), fg = sigmoidf_(acc[1][r]);
        const float og = sigmoidf_(acc[2][r]), ug = tanhf_(acc[3][r]);
        const float cn = fg * c[r] + ig * ug;
        const float hn = og * tanhf_(cn);
        if (advance && e < E) {
            float *st = state + ((long long)g * E + e) * 2 * kL;
            st[j] = cn; st[kL + j] = hn;
        }
        if (tslot >= 0 && e < E) {                                 // what lstm_fwd_kernel<true> would store
            const long long n = (long long)g * Ntot + (long long)tslot * E + e;
            st_stream(Zc + (n * kG4 + j), ig); st_stream(Zc + (n * kG4 + 64 + j), fg); st_stream(Zc + (n * kG4 + 128 + j), og); st_stream(Zc + (n * kG4 + 192 + j), ug);
            st_stream(Ccc + (n * kL + j), cn); st_stream(Hhc + (n * kL + j), hn);
        }
        XH[j * kXLd + erow[r]] = hn;
    }
    __syncthreads();
    FSTAMP();
    // ---- phase 4: head.  thread -> (env = tid & 63, outputs 2*(tid>>6), 2*(tid>>6)+1)
    float *LG = XH + 64 * kXLd;                        // [64][8] logits
    {
        const float *Wo = WoS, *bo = WoS + kL * kOut;           // staged at kernel entry
        const int e = tid & 63, k0 = 2 * (tid >> 6);
        float s0 = 0.f, s1 = 0.f;
#pragma unroll 8
        for (int jj = 0; jj < kL; ++jj) {
            const float h = XH[jj * kXLd + e];
            s0 += h * Wo[jj * kOut + k0];
            s1 += h * Wo[jj * kOut + k0 + 1];
        }
        LG[e * kOut + k0] = s0 + bo[k0];
        LG[e * kOut + k0 + 1] = s1 + bo[k0 + 1];
    }
    __syncthreads();
    if (tid < 64 && e0 + tid < E) {
        const long long idx = (long long)(e0 + tid) * lay.A + a;
        const float *lg = LG + tid * kOut;
        if (tower == 0) {
            const int na = n_act[a];
            float mx = -INFINITY;
            for (int k = 0; k < na; ++k) mx = fmaxf(mx, lg[k]);
            float pk[kOut], sum = 0.f;
#pragma unroll
            for (int k = 0; k < kOut; ++k) { pk[k] = k < na ? expf(lg[k] - mx) : 0.f; sum += pk[k]; }
            float pn[kOut];
#pragma unroll
            for (int k = 0; k < kOut; ++k) pn[k] = pk[k] / sum;
            for (int k = 0; k < lay.AMAX; ++k) pi_out[idx * lay.AMAX + k] = k < kOut ? pn[k] : 0.f;
            if (action_out) action_out[idx] = sample_action(pn, na, seed, step, idx);   // utils.py:155-157
        } else {
            v_out[idx] = lg[0];
        }
    }
    FSTAMP();
    if (stamp) dbg[63] = nstamp;
    if (dbg && threadIdx.x == 0) dbg[64 + 2 * blockIdx.x + 1] = wall_clock64();
#undef FSTAMP
}

// ------------------------------------------------------------------------------------------------
// Weight-stationary rollout forward.  The kernel above streams a tower's 330 KB of weights out of L2 once per
// 64-instance tile and runs 1.56 rounds of latency-chain workgroups.  Here ONE 8-wave workgroup per (tower, fifth of
// the instances) loads the weights ONCE into registers -- wave w owns gate columns [32w, 32w + 32) of [Wx ; Wh]
// (KS2 = (H + 64) / 2 MFMA steps = 144 registers per lane) and, for w < H / 32, column tile w of W1 -- and then
// walks its 32-instance tiles as a software pipeline of TWO barrier intervals per tile (round 1: six):
//   interval 1  gate tile of this wave for tile t (KS2 dependent MFMAs, [Wx;Wh] stationary, A = [X1 | h_prev] rows
//               of XH read as 16-byte quads).  Riding in its issue shadow, placed by hand between the MFMAs: the head
//               dot products of tile t-1, tile t's X1 rows streamed to the activation cache, tile t+1's obs -> LDS,
//               the global prefetches of tile t+1's state and tile t+2's obs; every 8 tiles the softmax + sampling of
//               the buffered logits.  Ends with the gate tile -> LDS [256][33].
//   interval 2  cell update of tile t (thread = (instance, 4 units): float4 state / cache traffic, h -> LDS) and the
//               first layer of tile t+1 (X1 tile per wave, W1 in LDS, only the K rows of W1's block that feed the
//               tile).  Waves w and w+4 share a SIMD and run the two in opposite order: one is an exp / rcp chain, the
//               other a dependent MFMA chain.
// G x S workgroups (S = 256 / G) ~ one per CU; every instance tile of a tower is a loop iteration instead of a
// workgroup.  Rules the kernel is built around (each cost 5-15 % when violated, measured):
//   * vmcnt retires in order and counts stores: a prefetched register is consumed (or touched by an empty asm) BEFORE
//     the interval's streaming stores are issued; stores inside the MFMA stream are branch-free (clamped duplicates, a
//     spare tile behind the buffer when the cache is off) -- a conditional store splits the block and spills;
//   * no spills at all: a spilled VGPR is reloaded through vmcnt, i.e. behind every store in flight.  Per-thread LDS /
//     cache offsets are re-derived from an opaque copy of the thread index inside each interval instead of being
//     hoisted (144 of the 256 VGPRs hold the weights), global addresses are uniform base + 32-bit lane offset.
// ------------------------------------------------------------------------------------------------
constexpr int kWsLdx = 36;       // activations [k][32 instances + 4]
constexpr int kWsLdg = 33;       // gate pre-activations [256 columns][32 instances + 1]
constexpr int kWsBuf = 8;        // tiles whose logits are buffered before the softmax / sampling pass

template <int KS2>
__global__ void __launch_bounds__(512, 1)
policy_fwd_ws_kernel(const float *__restrict__ params, Layout lay, const int *__restrict__ n_act,
                     const float *__restrict__ obs, const uint8_t *__restrict__ done, float *state, int advance,
                     int E, int S, float *__restrict__ pi_out, float *__restrict__ v_out, int *action_out,
                     unsigned long long seed, unsigned long long step,
                     int tslot, long long Ntot, float *X1c, float *Zc, float *Hhc, float *Ccc, float *Hpc,
                     long long *dbg, const float *__restrict__ Wr, const int *__restrict__ kr, int dbg_tid, const int *__restrict__ wgmap) {
    constexpr int H = 2 * KS2 - 64, NCT = H / 32;
    static_assert((32 * (H / 4)) % 512 == 0 || (32 * (H / 4)) % 512 == 256, "X1 store: duplicates come from 256 lanes below");
    const bool stamp = dbg && blockIdx.x == 0 && (int)threadIdx.x == dbg_tid;
    int nstamp = 0;
#define WSTAMP() do { if (stamp && nstamp < 60) dbg[nstamp++] = clock64(); } while (0)
    WSTAMP();
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int LDK = H + 64 + 4;                     // [X1 | h_prev] row of one instance (+4: 16-byte reads stay conflict-free)
    float *XH = (float *)smem_raw;                      // [32 instances][LDK]: A operand of the gate GEMM, read as float4 quads
    float *Hs = XH + 32 * LDK;                          // [64 k][kWsLdx]: obs tile, k-major (A operand of the first layer)
    float *Hn = Hs + 64 * kWsLdx;                       // [64 units][kWsLdg]: new h, k-major for the head
    float *Gz = Hn + kL * kWsLdg;                       // [256][kWsLdg]
    float *WoS = Gz + kG4 * kWsLdg;                     // [64][8] + [8] (+ pad to 528)
    float *LG = WoS + 528;                              // [kWsBuf tiles][32][8] logits, then [2 halves of the units][32][8] partial sums
    float *LGp = LG + kWsBuf * 32 * kOut;
    float *W1s = LGp + 2 * 32 * kOut;                    // [SMAX][H]: W1 is small enough to sit in LDS for the whole launch
    // (numbering the S workgroups of a tower onto one XCD so that four of the five weight reads hit its L2 halves the
    //  prologue but leaves two XCDs with 35 workgroups for 32 CUs: 126 -> 207 us)
    // (tower, split) of this workgroup: by table when the host laid the S workgroups of a tower onto ONE XCD (workgroups go
    // to the 8 XCDs round-robin by id), so that four of a tower's five weight reads hit that XCD's L2
    int g = blockIdx.x % lay.G, sp = blockIdx.x / lay.G;
    if (wgmap) {
        const int gs = wgmap[blockIdx.x];
        if (gs < 0) return;
        g = gs >> 8; sp = gs & 255;
    }
    // the workgroup's instances [eb, Ee): the tower's 16-instance half tiles dealt out evenly (E = 1024 over S = 5: 13 / 13 /
    // 13 / 13 / 12 halves = at most 6.5 tiles; whole tiles would be 7 / 6 / 7 / 6 / 6).  It walks them as 32-instance tiles
    // from eb; a last tile of <= 16 instances runs the half-tile gate GEMM (gate_interval<true>)
    const int n_half = (E + 15) / 16;
    const int eb = 16 * (int)((long long)n_half * sp / S);
    const int Ee = min(E, 16 * (int)((long long)n_half * (sp + 1) / S));
    const int nt = (Ee - eb + 31) / 32;                            // tiles of this workgroup, tile i = instances eb + 32 i ...
    const bool last_half = Ee - (eb + 32 * (nt - 1)) <= 16;
    const int a = g >> 1, tower = g & 1, SMAX = lay.SMAX;
    const float *P = params + (long long)g * lay.stride;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31;
    const int col = 32 * wave + li;
    {   // W1 -> LDS: eight 16-byte loads in flight per thread and round (one round for SMAX <= 64)
        const int tot = SMAX * H / 4;
        const float4 *src = reinterpret_cast<const float4 *>(P + lay.oW1);
        for (int base = tid; base < tot; base += 8 * 512) {
            float4 wq[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) { const int i = base + 512 * q; wq[q] = src[i < tot ? i : tot - 1]; }
#pragma unroll
            for (int q = 0; q < 8; ++q) { const int i = base + 512 * q; if (i < tot) reinterpret_cast<float4 *>(W1s)[i] = wq[q]; }
        }
    }
    const float blc = P[lay.obl + col];
    const float b1c = wave < NCT ? P[lay.ob1 + col] : 0.f;
    if (tid < (kL * kOut + kOut) / 4)
        *reinterpret_cast<float4 *>(WoS + 4 * tid) = *reinterpret_cast<const float4 *>(P + lay.oWo + 4 * tid);
    // ---- per-tile roles: obs loader (one float4), cell thread = (instance ce, units cu..cu+3)
    const int q4 = SMAX >> 2, AS = lay.A * SMAX;
    const bool ob_on = tid < 32 * q4;
    const int om = ob_on ? tid / q4 : 0, ok4 = ob_on ? (tid % q4) * 4 : 0;
    const int ce = tid >> 4, cu = (tid & 15) * 4;
    // addressing: workgroup-uniform 64-bit base (SGPRs) + 32-bit lane offset, so that no per-array 64-bit lane
    // address is hoisted out of the tile loop into (spilled) VGPRs
    float *stb = state + (long long)g * E * 2 * kL;
    const float *obs_a = obs + a * SMAX;
    auto fetch_obs = [&](int e0) {
        const unsigned e = (unsigned)(e0 + om < Ee ? e0 + om : Ee - 1);
        return *reinterpret_cast<const float4 *>(obs_a + (e * (unsigned)AS + (unsigned)ok4));
    };
    auto fetch_state = [&](int e0, int off) {
        const unsigned e = (unsigned)(e0 + ce < Ee ? e0 + ce : Ee - 1);
        return *reinterpret_cast<const float4 *>(stb + (e * (unsigned)(2 * kL) + (unsigned)(off + cu)));
    };
    auto fetch_keep = [&](int e0) { return 1.0f - (float)done[(unsigned)(e0 + ce < Ee ? e0 + ce : Ee - 1)]; };
    float4 ov = make_float4(0.f, 0.f, 0.f, 0.f), c4 = ov, h4 = ov;
    float keep = 0.f;
    if (eb >= Ee) return;                                      // fewer half tiles than workgroups per tower
    if (dbg && threadIdx.x == 0) dbg[64 + 2 * blockIdx.x] = wall_clock64();      // tools/bench_fwd.py: start / end of every workgroup (100 MHz)
    ov = fetch_obs(eb); c4 = fetch_state(eb, 0); h4 = fetch_state(eb, kL); keep = fetch_keep(eb);
    // ---- stationary operand, requested LAST: the first tile's obs / first-layer phases wait (counted, in order) only
    // for what was requested before it, so the 144 weight loads land under them instead of in front of the loop
    float bwg[KS2];
    {
        // contraction index of MFMA step s2 in lane half kh: k = 8 * (s2 / 4) + 4 * kh + s2 % 4, so that a lane's A
        // operands of four consecutive steps are one 16-byte LDS read; the weights come from the register-order copy
        // (register_order_kernel): one coalesced 16-byte load per four steps
        const float4 *src4 = reinterpret_cast<const float4 *>(Wr) + ((long long)(g * 8 + wave) * (KS2 / 4)) * 64 + lane;
#pragma unroll
        for (int j4 = 0; j4 < KS2 / 4; ++j4) {
            const float4 w4 = src4[(long long)j4 * 64];
            bwg[4 * j4] = w4.x; bwg[4 * j4 + 1] = w4.y; bwg[4 * j4 + 2] = w4.z; bwg[4 * j4 + 3] = w4.w;
        }
    }
    // softmax + action of the buffered tiles (~600 instructions per instance, half of them float64): one instance
    // per thread for up to kWsBuf tiles at once instead of 32 threads after every tile
    auto emit = [&](int e0p, int nbuf) {
        if (tid >= 32 * nbuf || e0p + tid >= Ee) return;
        const long long idx = (long long)(e0p + tid) * lay.A + a;
        const float *lg = LG + tid * kOut;
        if (tower == 0) {
            const int na = n_act[a];
            float mx = -INFINITY;
            for (int k = 0; k < na; ++k) mx = fmaxf(mx, lg[k]);
            float pk[kOut], sum = 0.f;
#pragma unroll
            for (int k = 0; k < kOut; ++k) { pk[k] = k < na ? expf(lg[k] - mx) : 0.f; sum += pk[k]; }
            float pn[kOut];
#pragma unroll
            for (int k = 0; k < kOut; ++k) pn[k] = pk[k] / sum;
            for (int k = 0; k < lay.AMAX; ++k) pi_out[idx * lay.AMAX + k] = k < kOut ? pn[k] : 0.f;
            if (action_out) action_out[idx] = sample_action(pn, na, seed, step, idx);   // utils.py:155-157
        } else {
            v_out[idx] = lg[0];
        }
    };
    WSTAMP();
    // Software pipeline over the tiles: the first layer of tile tt + 1 shares a barrier interval with the cell update of
    // tile tt (one is a dependent MFMA chain, the other a chain of exp / rcp: waves w and w + 4 sit on the same SIMD and
    // run the two in opposite order), so a tile costs three barrier intervals -- gate GEMM | cell + next first layer |
    // head -- instead of six.
    // rows [SMAX, 64) of the obs staging area stay zero for the whole launch
    for (int idx = tid; idx < (64 - SMAX) * 32; idx += 512) Hs[(SMAX + (idx >> 5)) * kWsLdx + (idx & 31)] = 0.f;
    // the K range of this wave's first-layer tile: W1 is block-structured (obs rows of one kind feed one block of hidden
    // columns, agents/policies.py:41-61), rows outside the range only multiply stored zeros
    int ks_lo = 0, ks_hi = (SMAX + 1) >> 1;
    if (kr && wave < NCT) { ks_lo = kr[(a * 8 + wave) * 2]; ks_hi = kr[(a * 8 + wave) * 2 + 1]; }
    // Per-thread LDS / cache offsets are re-derived from an OPAQUE copy of the thread index inside each interval: hoisted
    // out of the tile loop they cost ~20 VGPRs the kernel does not have (144 hold the weights), and a spilled VGPR is
    // reloaded through vmcnt, i.e. behind every streaming store still in flight.
    auto opaque_tid = [&]() { int t_ = tid; asm volatile("" : "+v"(t_)); return t_; };
    auto stage_obs = [&]() {                                   // obs tile held in `ov` -> LDS, k-major
        if (ob_on) {
            Hs[(ok4 + 0) * kWsLdx + om] = ov.x; Hs[(ok4 + 1) * kWsLdx + om] = ov.y;
            Hs[(ok4 + 2) * kWsLdx + om] = ov.z; Hs[(ok4 + 3) * kWsLdx + om] = ov.w;
        }
    };
    // X1 = relu(obs W1 + b1) of the tile staged in Hs: wave w < NCT owns column tile w.  All LDS operands of eight
    // steps are requested first, then the dependent MFMA chain runs without waiting; the tile goes to XH only -- its
    // copy for the update's activation cache is streamed out of XH under the NEXT gate GEMM (store_x1)
    auto first_layer = [&]() {
        if (wave < NCT) {
            const int t_ = opaque_tid(), li = t_ & 31, kh = (t_ >> 5) & 1, col = 32 * (t_ >> 6) + li;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            for (int sb = ks_lo; sb < ks_hi; sb += 8) {
                float av[8], bv[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const bool on = sb + q < ks_hi;                // steps past the range: a zero A operand
                    const int s2 = on ? sb + q : ks_hi - 1;
                    const int row = 2 * s2 + kh < SMAX ? 2 * s2 + kh : SMAX - 1;   // obs rows past SMAX are zero; W1 rows clamped
                    const float x = Hs[(2 * s2 + kh) * kWsLdx + li];
                    av[q] = on ? x : 0.f;
                    bv[q] = W1s[row * H + col];
                }
#pragma unroll
                for (int q4 = 0; q4 < 2; ++q4) {
                    if (sb + 4 * q4 < ks_hi) {                         // wave-uniform
#pragma unroll
                        for (int q = 4 * q4; q < 4 * q4 + 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[q], acc, 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;
                float v0 = acc[r] + b1c;
                v0 = v0 > 0.f ? v0 : 0.f;
                XH[row * LDK + col] = v0;
            }
        }
    };
    // done-masked previous state of the tile held in c4 / h4 / keep: h -> columns [H, H + 64) of the instance's row;
    // returns the masked c (the cell update's input)
    float4 cm = make_float4(0.f, 0.f, 0.f, 0.f);
    auto stage_state = [&](int e0n, long long nbn, bool on) {
        const float4 hm = make_float4(h4.x * keep, h4.y * keep, h4.z * keep, h4.w * keep);
        const float4 cmn = make_float4(c4.x * keep, c4.y * keep, c4.z * keep, c4.w * keep);
        if (on) {
            const int t_ = opaque_tid(), ce = t_ >> 4, cu = (t_ & 15) * 4;
            *reinterpret_cast<float4 *>(XH + ce * LDK + H + cu) = hm;
            if (tslot >= 0 && e0n + ce < Ee) st_stream4(Hpc + nbn * kL + (unsigned)(ce * kL + cu), hm);
        }
        return cmn;
    };
    // cell update of (instance ce, units cu..cu+3) from the gate pre-activations in Gz; h -> LDS rows [0, 64)
    auto cell = [&](int e0c, long long nbc) {
        const int t_ = opaque_tid(), ce = t_ >> 4, cu = (t_ & 15) * 4;
        const unsigned st_lane = (unsigned)(ce * 2 * kL + cu), z_lane = (unsigned)(ce * kG4 + cu), c_lane = (unsigned)(ce * kL + cu);
        float gi[4], gf[4], go[4], gu[4], cn[4], hn[4];
        const float cin[4] = {cm.x, cm.y, cm.z, cm.w};
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int u = cu + jj;
            gi[jj] = sigmoidf_(Gz[u * kWsLdg + ce]); gf[jj] = sigmoidf_(Gz[(64 + u) * kWsLdg + ce]);
            go[jj] = sigmoidf_(Gz[(128 + u) * kWsLdg + ce]); gu[jj] = tanhf_(Gz[(192 + u) * kWsLdg + ce]);
            cn[jj] = gf[jj] * cin[jj] + gi[jj] * gu[jj];
            hn[jj] = go[jj] * tanhf_(cn[jj]);
            Hn[u * kWsLdg + ce] = hn[jj];
        }
        if (e0c + ce < Ee) {
            const float4 c4n = make_float4(cn[0], cn[1], cn[2], cn[3]), h4n = make_float4(hn[0], hn[1], hn[2], hn[3]);
            if (advance) {
                float *st = stb + (long long)e0c * 2 * kL;
                st_stream4(st + st_lane, c4n); st_stream4(st + (st_lane + kL), h4n);
            }
            if (tslot >= 0) {                                  // what lstm_fwd_kernel<true> would store
                float *zr = Zc + nbc * kG4;
                st_stream4(zr + z_lane, make_float4(gi[0], gi[1], gi[2], gi[3]));
                st_stream4(zr + (z_lane + 64), make_float4(gf[0], gf[1], gf[2], gf[3]));
                st_stream4(zr + (z_lane + 128), make_float4(go[0], go[1], go[2], go[3]));
                st_stream4(zr + (z_lane + 192), make_float4(gu[0], gu[1], gu[2], gu[3]));
                st_stream4(Ccc + nbc * kL + c_lane, c4n);
                st_stream4(Hhc + nbc * kL + c_lane, h4n);
            }
        }
    };
    // head, second half: logits of tile tj = (units 0..31) + (units 32..63) + bias -> slot of the buffer
    auto head_finish = [&](int ij) {                           // ij: tile index within the workgroup
        if (tid < 256) {
            const int e = tid & 31, k0 = (tid >> 5) & 7, slot = ij % kWsBuf;
            LG[(slot * 32 + e) * kOut + k0] = (LGp[tid] + LGp[256 + tid]) + WoS[kL * kOut + k0];
        }
    };
    const long long nbase = (long long)g * Ntot + (long long)(tslot < 0 ? 0 : tslot) * E;   // cache row of instance 0
    // ---- pipeline fill: tile 0 staged, first layer + state in place, tile 1 requested
    stage_obs();
    __syncthreads();
    first_layer();
    cm = stage_state(eb, nbase + eb, true);
    // (all prefetches are unconditional -- the instance index is clamped -- so that each prefetched register has ONE
    //  definition per iteration and is waited for where it is consumed, not at a control-flow join)
    ov = fetch_obs(eb + 32);
    __syncthreads();
    WSTAMP();
    // One tile = two barrier intervals.  HALF: the tile holds <= 16 instances (only ever the workgroup's last one): its gate
    // GEMM runs on v_mfma_f32_16x16x4_f32 with the SAME stationary registers, re-paired in place for that instruction by
    // v_permlane16_swap (below) -- half the matrix-core time of a 32-row tile, which is what lets the tower's instances be
    // dealt out in 16-instance units.  Everything else of the tile (riding work, cell update, head) is shared.
    auto tile = [&](auto half_tag, int it) {
        constexpr bool HALF = decltype(half_tag)::value;
        const int e0 = eb + 32 * it;
        const long long nb0 = nbase + e0;                         // first cache row of the tile
        const bool more = it + 1 < nt;
        // ---- interval 1: next tile's obs -> LDS (the staging area is free since its first layer ran), the obs of the
        // tile after it requested; gate tile of this wave = bl + [X1 | h] [Wx ; Wh][:, 32w .. 32w+32)
        // the group of kWsBuf tiles whose last logits were written two intervals ago: softmax + action (LG is rewritten
        // only after this interval's barrier)
        if (it >= 2 && (it - 2) % kWsBuf == kWsBuf - 1) emit(eb + 32 * (it - 1 - kWsBuf), kWsBuf);
        if (more) stage_obs();
        ov = fetch_obs(e0 + 64);
        c4 = fetch_state(e0 + 32, 0); h4 = fetch_state(e0 + 32, kL); keep = fetch_keep(e0 + 32);   // consumed right after the GEMM
        // head of the PREVIOUS tile, first half: thread -> (instance, output, half of the units): 8 batches of 4 LDS
        // operand pairs, placed by hand between the MFMAs of the gate GEMM (whose dependent chain leaves the issue slots
        // free); unconditional -- the result of the first iteration is never read
        float hs0 = 0.f;
        {
            const int t_ = opaque_tid(), li = t_ & 31, kh = (t_ >> 5) & 1, col = 32 * (t_ >> 6) + li;
            const float *hp = Hn + (32 * (t_ >> 8)) * kWsLdg + (t_ & 31), *wp = WoS + (32 * (t_ >> 8)) * kOut + ((t_ >> 5) & 7);
            f32x16 acc;
            f32x4 acl, ach;                                        // HALF: columns [32w, 32w + 16) / [32w + 16, 32w + 32) of 16 instances
            float bl_ = blc;
            asm volatile("" : "+v"(bl_));                          // (a hoisted 16-register splat would be spilled)
            constexpr int NJ = KS2 / 4, PER = NJ / 4, HP = NJ / 8;
            if constexpr (HALF) {
                // v_mfma_f32_16x16x4_f32 wants B[k = lane / 16][n = lane % 16]; register 4 j4 + c holds W[8 j4 + 4 (lane / 32) + c]
                // [32 w + lane % 32], i.e. its four 16-lane rows are (k_c, cols 0-15), (k_c, cols 16-31), (k_c + 4, cols 0-15),
                // (k_c + 4, cols 16-31).  v_permlane16_swap exchanges the odd rows of its first operand with the even rows of its
                // second: registers (c, c + 2) become rows (k_c, k_c + 2, k_c + 4, k_c + 6) x cols 0-15 and the same x cols 16-31 --
                // two valid B operands whose A operand is X[row = lane % 16][8 j4 + 2 (lane / 16) + c], c = 0 / 1: one 8-byte
                // LDS read per lane and j4.  In place: the half tile is the workgroup's last one, the weights are not needed again.
#pragma unroll
                for (int j4 = 0; j4 < NJ; ++j4) {
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const auto r2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(bwg[4 * j4 + c]), __float_as_uint(bwg[4 * j4 + c + 2]), false, false);
                        bwg[4 * j4 + c] = __uint_as_float(r2[0]); bwg[4 * j4 + c + 2] = __uint_as_float(r2[1]);
                    }
                }
                const float bl_lo = __shfl(bl_, (t_ & 15), 64), bl_hi = __shfl(bl_, 16 + (t_ & 15), 64);   // bias of the lane's two columns
#pragma unroll
                for (int r = 0; r < 4; ++r) { acl[r] = bl_lo; ach[r] = bl_hi; }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = bl_;
            }
            const float4 *As = reinterpret_cast<const float4 *>(XH + li * LDK + 4 * kh);
            const float2 *Ah = reinterpret_cast<const float2 *>(XH + (t_ & 15) * LDK + 2 * ((t_ >> 4) & 3));
            float hv[4], wv[4];
            float4 x1q = make_float4(0.f, 0.f, 0.f, 0.f);
            unsigned x1o = 0;
            float *x1b = tslot >= 0 ? X1c + nb0 * H : X1c + (long long)lay.G * Ntot * H;   // cache off: the spare tile behind the buffer
            float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f);
            float2 a2 = make_float2(0.f, 0.f);
            if constexpr (HALF) a2 = Ah[0]; else a4 = As[0];
#pragma unroll
            for (int j4 = 0; j4 < NJ; ++j4) {
                float4 an = a4;
                float2 a2n = a2;
                if constexpr (HALF) a2n = Ah[4 * (j4 + 1 < NJ ? j4 + 1 : j4)];      // next operands in flight under this group's MFMAs
                else an = As[2 * (j4 + 1 < NJ ? j4 + 1 : j4)];
                if (j4 % HP == 0 && j4 / HP < 8) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) { hv[q] = hp[(4 * (j4 / HP) + q) * kWsLdg]; wv[q] = wp[(4 * (j4 / HP) + q) * kOut]; }
                }
                if constexpr (HALF) {
                    acl = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.x, bwg[4 * j4], acl, 0, 0, 0);
                    ach = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.x, bwg[4 * j4 + 2], ach, 0, 0, 0);
                    acl = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.y, bwg[4 * j4 + 1], acl, 0, 0, 0);
                    ach = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.y, bwg[4 * j4 + 3], ach, 0, 0, 0);
                } else {
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, bwg[4 * j4], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, bwg[4 * j4 + 1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, bwg[4 * j4 + 2], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, bwg[4 * j4 + 3], acc, 0, 0, 0);
                }
                if (j4 % HP == HP - 1 && j4 / HP < 8) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) hs0 += hv[q] * wv[q];
                }
                // this tile's X1 (in XH, row-major) -> the update's activation cache, 16 bytes per lane and quarter.  Branch-free:
                // lanes past the tile re-store the quad of the lane 256 below, rows past the workgroup's last instance the last
                // valid row (same values; distinct addresses, so that the duplicates do not serialise on one)
                if (j4 % PER == 3 && j4 / PER < (32 * (H / 4) + 511) / 512) {
                    int f = t_ + 512 * (j4 / PER);
                    f = f < 32 * (H / 4) ? f : f - 256;
                    int row = f / (H / 4);
                    const int c4i = f - row * (H / 4);
                    row = e0 + row < Ee ? row : Ee - 1 - e0;
                    x1q = *reinterpret_cast<const float4 *>(XH + row * LDK + 4 * c4i);
                    x1o = (unsigned)(row * H + 4 * c4i);
                }
                if (j4 % PER == 5 && j4 / PER < (32 * (H / 4) + 511) / 512 ) st_stream4(x1b + x1o, x1q);
                a4 = an; a2 = a2n;
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (HALF) {                                  // D[row = 4 (lane / 16) + r][col = lane % 16]
                const int cl = 32 * (t_ >> 6) + (t_ & 15), rw = 4 * ((t_ >> 4) & 3);
#pragma unroll
                for (int r = 0; r < 4; ++r) { Gz[cl * kWsLdg + rw + r] = acl[r]; Gz[(cl + 16) * kWsLdg + rw + r] = ach[r]; }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) Gz[col * kWsLdg + (r & 3) + 8 * (r >> 2) + 4 * kh] = acc[r];
            }
            LGp[t_] = hs0;                                        // [half][output][instance]
            // what was requested before the GEMM has long arrived: wait for it HERE, in front of the stores below (vmcnt
            // retires in order and the stores are conditional, so a later wait would be a wait for the stores)
            asm volatile("" :: "v"(ov.x), "v"(ov.y), "v"(ov.z), "v"(ov.w), "v"(c4.x), "v"(c4.y), "v"(c4.z), "v"(c4.w),
                         "v"(h4.x), "v"(h4.y), "v"(h4.z), "v"(h4.w), "v"(keep));
        }
        WSTAMP();
        __syncthreads();
        WSTAMP();
        if (it > 0) head_finish(it - 1);
        // ---- interval 2: cell update of this tile | first layer of the next one (XH is free: the gate GEMM is done).
        // vmcnt retires in order and counts stores: what was requested before the GEMM is consumed (or at least waited
        // for) BEFORE this interval's ~30 streaming stores are issued, never behind them
        const float4 cmn = stage_state(e0 + 32, nb0 + 32, more);
#pragma unroll 1
        for (int ph = 0; ph < 2; ++ph) {
            if ((ph == 0) == (wave < 4)) { if (!HALF || wave < 4) cell(e0, nb0); }    // HALF: instances 16 .. 31 (waves 4 - 7) do not exist
            else if (more) first_layer();
            WSTAMP();
        }
        cm = cmn;
        __syncthreads();
        WSTAMP();
    };
    const int nt_full = last_half ? nt - 1 : nt;
    for (int it = 0; it < nt_full; ++it) tile(std::false_type{}, it);
    // The half tile MUST stay the last use of the stationary weights: it re-pairs bwg[] in place for the 16x16x4 form (72
    // v_permlane16_swap) and nothing restores the 32x32x2 order afterwards.  It also accumulates its gate pre-activations in K = 4
    // steps instead of K = 2, so an instance's logits depend, at fp32 rounding, on E and on where the instance falls in its
    // workgroup's share (tests/test_model_gpu.py keeps an E where every split ends in a half tile).
    if (last_half) tile(std::true_type{}, nt - 1);
    // ---- drain: a full group still waiting for its softmax, then the head of the last tile and its group
    {
        const int il = nt - 1;
        if (il >= 1 && (il - 1) % kWsBuf == kWsBuf - 1) emit(eb + 32 * (il - kWsBuf), kWsBuf);
        float hs0 = 0.f;
        const int e = tid & 31, k0 = (tid >> 5) & 7, hf = tid >> 8;
#pragma unroll 8
        for (int jj = 0; jj < 32; ++jj) hs0 += Hn[(32 * hf + jj) * kWsLdg + e] * WoS[(32 * hf + jj) * kOut + k0];
        LGp[tid] = hs0;
        __syncthreads();
        head_finish(il);
        __syncthreads();
        const int slot = il % kWsBuf;
        emit(eb + 32 * (il - slot), slot + 1);
    }
    if (stamp) dbg[63] = nstamp;
    if (dbg && threadIdx.x == 0) dbg[64 + 2 * blockIdx.x + 1] = wall_clock64();
#undef WSTAMP
}

// n-step returns and advantages (agents/utils.py:202-228): float64 recursion from the back with
// POST-step dones, Adv = R - v, cast to float32.   rew f64 [T][E][A], val f32, done_all u8 [T+1][E]
// The buffer holds the env's RAW rewards (so that the env can write them in place, tsc_model_rollout_slot); the
// reward / reward_norm and clip of IA2C.add_transition (agents/models.py:223-226) are applied here, same float64 operations.
__global__ void returns_kernel(const double *rew, const float *val, const uint8_t *done_all, const float *Rboot,
                               int T, int E, int A, double gamma, double rnorm, double rclip, float *Rs, float *Advs) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= E * A) return;
    const int e = idx / A;
    double R = (double)Rboot[idx];
    for (int t = T - 1; t >= 0; --t) {
        const double d = (double)done_all[(long long)(t + 1) * E + e];
        double r = rew[(long long)t * E * A + idx];
        if (rnorm != 0.0) r = r / rnorm;                         // agents/models.py:223-224
        if (rclip != 0.0) r = fmin(fmax(r, -rclip), rclip);      // :225-226
        R = r + gamma * R * (1.0 - d);
        const double adv = R - (double)val[(long long)t * E * A + idx];
        Rs[(long long)t * E * A + idx] = (float)R;
        Advs[(long long)t * E * A + idx] = (float)adv;
    }
}

__global__ void add_transition_kernel(int E, int A, int SMAX, const float *obs, const uint8_t *done_pre,
                                      const int *action, const double *reward, const float *value,
                                      const uint8_t *done_post, float *obs_t,
                                      int *act_t, double *rew_t, float *val_t, uint8_t *done_t, uint8_t *done_t1) {
    // every source may already BE the slot (tsc_model_rollout_slot): then there is nothing to copy
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long no = (long long)E * A * SMAX;
    if (i < no && obs_t != obs) obs_t[i] = obs[i];
    if (i < (long long)E * A) {
        if (rew_t != reward) rew_t[i] = reward[i];               // raw: normalised / clipped by returns_kernel
        if (act_t != action) act_t[i] = action[i];
        if (val_t != value) val_t[i] = value[i];
    }
    if (i < E) {
        if (done_t != done_pre) done_t[i] = done_pre[i];
        if (done_t1 != done_post) done_t1[i] = done_post[i];
    }
}

__global__ void sample_kernel(const float *pi, const int *n_act, int E, int A, int AMAX, unsigned long long seed,
                              unsigned long long step, int *action) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= E * A) return;
    const int a = idx % A, na = n_act[a];
    const int ans = sample_action(pi + (long long)idx * AMAX, na, seed, step, idx);
    action[idx] = ans;
}

// per-agent global norm (tf.clip_by_global_norm over both towers, agents/policies.py:54-57)
// kNormParts workgroups per agent, each a fixed slice; partial sums folded in slice order (deterministic)
constexpr int kNormParts = 16;
__global__ void grad_norm_kernel(const float *grad, long long per_agent, double gscale, double *part) {
    __shared__ double red[256];
    const int a = blockIdx.x, c = blockIdx.y;
    const long long len = (per_agent + kNormParts - 1) / kNormParts, lo = c * len, hi = lo + len < per_agent ? lo + len : per_agent;
    const float *gp = grad + (long long)a * per_agent;
    double s = 0.0;
    for (long long i = lo + threadIdx.x; i < hi; i += 256) { const double v = (double)gp[i] * gscale; s += v * v; }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[a * kNormParts + c] = red[0];
}
__global__ void grad_norm_fold_kernel(const double *part, int A, double *norm2) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= A) return;
    double s = 0.0;
    for (int c = 0; c < kNormParts; ++c) s += part[a * kNormParts + c];
    norm2[a] = s;
}

// TF1 RMSPropOptimizer (momentum 0, not centred): ms = a ms + (1-a) g^2 ; w -= lr g / sqrt(ms + eps)
__global__ void rmsprop_kernel(float *w, float *ms, const float *grad, long long per_agent, long long total,
                               const double *norm2, float gscale, float clip, float lr, float alpha, float eps) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int a = (int)(i / per_agent);
    const float nrm = (float)sqrt(norm2[a]);
    float g = grad[i] * gscale;
    if (clip > 0.f) g = g * (clip / fmaxf(nrm, clip));
    const float m = alpha * ms[i] + (1.0f - alpha) * g * g;
    ms[i] = m;
    w[i] = w[i] - lr * g / sqrtf(m + eps);
}

__global__ void transpose_wx_kernel(const float *params, Layout lay, float *WxT) {
    // WxT[g][c][h] = Wx[g][h][c]
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long per = (long long)lay.H * lay.NZ;
    if (i >= per * lay.G) return;
    const int g = (int)(i / per);
    const long long r = i % per;
    const int c = (int)(r / lay.H), h = (int)(r % lay.H);
    WxT[i] = params[(long long)g * lay.stride + lay.oWx + (long long)h * lay.NZ + c];
}

// Wg[g][k][4*u + q] = [Wx ; Wh][g][k][64*q + u]: the four gates of a unit side by side, so that the fused rollout
// forward fetches a lane's four B operands (one per gate tile) with ONE 16-byte load.  Dword loads reach only a
// fraction of the L2 rate, and the weight stream out of L2 is what bounds that kernel.
__global__ void interleave_gates_kernel(const float *params, Layout lay, float *Wg) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long per = (long long)(lay.H + kL) * kG4;
    if (i >= per * lay.G) return;
    const long long g = i / per, r = i % per;
    const int k = (int)(r / kG4), c = (int)(r % kG4), u = c >> 2, q = c & 3;
    Wg[i] = params[g * lay.stride + lay.oWx + (long long)k * kG4 + 64 * q + u];
}

// Wr[g][wave w][j4][lane][c] = [Wx ; Wh][g][k = 8 j4 + 4 (lane / 32) + c][col = 32 w + lane % 32]: the stationary operand of
// policy_fwd_ws_kernel in REGISTER order, so that its prologue is KS2 / 4 fully coalesced 16-byte loads per lane (1 KB
// per wavefront instruction) instead of KS2 dword loads that each touch two 128-byte row segments.
__global__ void register_order_kernel(const float *params, Layout lay, float *Wr) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long per = (long long)(lay.H + kL) * kG4;
    if (i >= per * lay.G) return;
    const long long g = i / per, r = i % per;
    const int c = (int)(r & 3), lane = (int)((r >> 2) & 63);
    const long long rest = r >> 8;                       // w * (KS2 / 4) + j4
    const int nj = (lay.H + kL) / 8, w = (int)(rest / nj), j4 = (int)(rest % nj);
    const int k = 8 * j4 + 4 * (lane >> 5) + c, col = 32 * w + (lane & 31);
    Wr[i] = params[g * lay.stride + lay.oWx + (long long)k * kG4 + col];
}

__global__ void fill_kernel(float *p, long long n, float v) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

}  // namespace

// ------------------------------------------------------------------------------------------------

// ------------------------------------------------------------------------------------------------
// LSTM weight gradients of one tower in ONE pass over the n-step batch (tf.gradients of
// agents/utils.py:88-116 w.r.t. wx, wh, b):   [dWx ; dWh] = [X1 | h_prev]^T dZ,   dbl = colsum(dZ).
// The output of a tower is only (H+64) x 256, so one workgroup (8 waves, one 32-column strip each, NT = (H+64)/32
// accumulator tiles per wave) keeps ALL of it in registers and streams its share of the rows exactly once:
//   * [X1 | h_prev] rows (the A operand, shared by all waves) are fetched once per workgroup with 16-byte loads,
//     staged through registers into a double-buffered LDS chunk of 16 rows;
//   * dZ (the B operand, private to a wave's column strip) goes from HBM straight into MFMA operand registers
//     (a 32x32x2 operand is two coalesced 128-byte row segments), one chunk ahead;
//   * every global load is unconditional (rows clamped; rows past the split contribute through a zeroed dZ).
// Deterministic: the row range of a tower is cut into S fixed splits (S x G workgroups ~ one per CU), partial
// sums go to the workspace and are added in split order by dwxh_reduce_kernel.  Replaces two grouped GEMM
// launches that each re-read dZ twice.
// ------------------------------------------------------------------------------------------------
template <int NT>
__global__ void __launch_bounds__(512, 1)
dwxh_kernel(const float *__restrict__ X1, const float *__restrict__ Hp, const float *__restrict__ dZ, long long N, int G,
            int S, long long rows_per_split, float *__restrict__ ws) {
    constexpr int HT = NT - 2, H = 32 * HT, KC = 16, LDA = NT * 32 + 4;     // +4: the two k rows of a step hit different banks
    constexpr int XQ = H / 4, NX = KC * XQ, NQ = NX + KC * (kL / 4), NLD = (NQ + 511) / 512;
    __shared__ __attribute__((aligned(16))) float As[2][KC][LDA];
    const int g = blockIdx.x % G, sp = blockIdx.x / G;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, kh = lane >> 5;
    const int col0 = 32 * wave;
    const long long n0 = (long long)sp * rows_per_split;
    long long n1 = n0 + rows_per_split;
    if (n1 > N) n1 = N;
    const float *x1 = X1 + (long long)g * N * H, *hp = Hp + (long long)g * N * kL;
    const float *dz = dZ + (long long)g * N * kG4 + col0 + li;
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float bsum = 0.f;
    // staging slots of this thread: (row in chunk, float4 column, source).  Kept in named scalars: as arrays
    // the compiler parks the staged float4s in scratch and waits for every global load right after issuing it.
    static_assert(NLD <= 3, "staging slots");
    struct Slot { int row, col; bool x, ok; };
    auto slot = [&](int q) {
        Slot o;
        int idx = tid + 512 * q;
        o.ok = idx < NQ;
        if (!o.ok) idx = NQ - 1;
        o.x = idx < NX;
        const int j = o.x ? idx : idx - NX, w = o.x ? XQ : kL / 4;
        o.row = j / w; o.col = 4 * (j % w);
        return o;
    };
    const Slot s0 = slot(0), s1 = slot(1), s2 = slot(2);
    float4 g0, g1, g2;
    auto fetch1 = [&](const Slot &o, long long row0) {
        long long row = row0 + o.row;
        if (row >= n1) row = n1 - 1;
        const float *src = o.x ? x1 + row * H + o.col : hp + row * kL + o.col;
        return *reinterpret_cast<const float4 *>(src);
    };
    auto commit1 = [&](const Slot &o, int buf, const float4 &v) {
        if (o.ok) *reinterpret_cast<float4 *>(&As[buf][o.row][(o.x ? 0 : H) + o.col]) = v;
    };
#define DWXH_FETCH(row0) do { g0 = fetch1(s0, row0); if (NLD > 1) g1 = fetch1(s1, row0); if (NLD > 2) g2 = fetch1(s2, row0); } while (0)
#define DWXH_COMMIT(buf) do { commit1(s0, buf, g0); if (NLD > 1) commit1(s1, buf, g1); if (NLD > 2) commit1(s2, buf, g2); } while (0)
    float bcur[KC / 2], bnxt[KC / 2];
    auto fetch_b = [&](float *b, long long row0) {
#pragma unroll
        for (int ks = 0; ks < KC / 2; ++ks) {
            const long long row = row0 + 2 * ks + kh;
            const float z = dz[(row < n1 ? row : n1 - 1) * kG4];
            b[ks] = row < n1 ? z : 0.f;
        }
    };
    if (n0 < n1) {
        DWXH_FETCH(n0); fetch_b(bcur, n0);
        DWXH_COMMIT(0);
        DWXH_FETCH(n0 + KC);
        __syncthreads();
        int buf = 0;
        for (long long row = n0; row < n1; row += KC, buf ^= 1) {
            fetch_b(bnxt, row + KC);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < KC / 2; ++ks) {
                float av[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) av[t] = As[buf][2 * ks + kh][32 * t + li];
                bsum += bcur[ks];
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bcur[ks], acc[t], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            DWXH_COMMIT(buf ^ 1);                               // chunk row+KC (fetched one iteration ago)
            DWXH_FETCH(row + 2 * KC);
#pragma unroll
            for (int ks = 0; ks < KC / 2; ++ks) bcur[ks] = bnxt[ks];
            __syncthreads();
        }
    }
    float *w = ws + ((long long)sp * G + g) * ((long long)(NT * 32 + 1) * kG4);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * kh;
            w[(long long)row * kG4 + col0 + li] = acc[t][r];
        }
    bsum += __shfl_xor(bsum, 32, 64);
    if (kh == 0) w[(long long)NT * 32 * kG4 + col0 + li] = bsum;
#undef DWXH_FETCH
#undef DWXH_COMMIT
}

// grads[g][oWx .. oWx + (H+64+1)*256) = sum over splits, in split order
__global__ void dwxh_reduce_kernel(const float *__restrict__ ws, int G, int S, long long per, float *__restrict__ grads,
                                   long long stride, long long off) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= per * G) return;
    const long long g = i / per, j = i % per;
    float acc = 0.f;
    for (int s = 0; s < S; ++s) acc += ws[((long long)s * G + g) * per + j];
    grads[g * stride + off + j] = acc;
}


// ------------------------------------------------------------------------------------------------
// First-layer gradients of one tower in ONE pass over the n-step batch, without ever writing dX1:
//   dX1 = (dZ Wx^T) * relu'(X1)   (32-row chunks, never leaves the registers)
//   dW1 = obs^T dX1,  db1 = colsum(dX1)        (agents/utils.py:66-74 through tf.gradients)
// One 8-wave workgroup per CU (S x G workgroups, each a fixed row range of one tower).  The unit of work is a 16 x 16
// tile of the chunk's dX1 (v_mfma_f32_16x16x4_f32): 2 row tiles x H/16 column units per 32-row chunk, dealt out so that the
// two waves of every SIMD (w, w + 4) own the same number of tiles (H = 224: 4 + 3 of 28), a column unit possibly split
// between two waves by row tile (round 2 gave a 32-column strip to each of 7 waves and kept the eighth as a loader: 7/8 of
// the MFMA rate at best).  All eight waves compute; the next chunk (dZ rows, row-major like HBM, and the obs rows) is
// fetched into registers at the top of a chunk and written to the other LDS buffer at its end (five 16-byte loads per
// thread).  A wave keeps the Wx^T slices of its (at most two) column units stationary (64 registers each); per tile 64
// MFMAs give dX1, masked by X1 > 0, and the accumulator registers go STRAIGHT back in as the B operand of the dW1 product
// against the obs tile (accumulator row = contraction index of the step), only for the 16-feature tiles of W1 that hold a
// structural non-zero in the column unit.
// Replaces the dX1 GEMM (second column tile 25 % empty, dZ read twice), the 11 GB dX1 round trip through HBM and the dW1
// GEMM.  Deterministic: partial dW1 | db1 per (workgroup, row-tile slot), added in that order by dx1w1_reduce2_kernel, which
// also applies the structural zeros of the block-diagonal first layer.
// ------------------------------------------------------------------------------------------------
constexpr int kD1Ld = 260;       // dZ chunk [32 rows][256 k + 4]: row-major like HBM, read as 16-byte A-operand quads
constexpr int kObLd = 68;        // obs chunk [32 rows][64 features + 4]

template <int NCU>   // H / 16
__global__ void __launch_bounds__(512, 1)
dx1w1_kernel2(const float *__restrict__ dZ, const float *__restrict__ X1, const float *__restrict__ WxT,
              const float *__restrict__ obs, long long N, int G, int S, long long rows_per_split, int A, int SMAX,
              float *__restrict__ ws, const int *__restrict__ ftm) {
    constexpr int H = 16 * NCU, NHU = 2 * NCU;                  // half units (column unit x row tile) per chunk
    constexpr int CHI = (NHU + 7) / 8, CLO = NHU / 8;           // tiles of waves 0..3 / 4..7
    static_assert(NHU % 8 == 0 || NHU % 8 == 4, "H must be a multiple of 32");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float *Az = (float *)smem_raw;                              // [2][32][kD1Ld]
    float *Ob = Az + 2 * 32 * kD1Ld;                            // [2][32][kObLd]
    const int g = blockIdx.x % G, sp = blockIdx.x / G, a = g >> 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), n = lane & 15, kq = lane >> 4;
    const long long n0 = (long long)sp * rows_per_split;
    long long n1 = n0 + rows_per_split;
    if (n1 > N) n1 = N;
    const float *dz = dZ + (long long)g * N * kG4, *x1 = X1 + (long long)g * N * H;
    const long long AS = (long long)A * SMAX;
    const int sq = SMAX >> 2;
    float *w = ws + ((long long)sp * G + g) * ((long long)2 * 65 * H);
    // this wave's half units [st, st + cnt): unit = h >> 1, row tile = h & 1
    const int st = wave < 4 ? wave * CHI : 4 * CHI + (wave - 4) * CLO, cnt = wave < 4 ? CHI : CLO;
    const int U0 = st >> 1;
    const bool a00 = 2 * U0 >= st, a01 = 2 * U0 + 1 < st + cnt, a10 = 2 * U0 + 2 < st + cnt, a11 = 2 * U0 + 3 < st + cnt;
    const int U1 = U0 + 1 < NCU ? U0 + 1 : NCU - 1;             // clamped when the second slot is unused
    const int col0 = 16 * U0 + n, col1 = 16 * U1 + n;
    // W1 is block-structured (obs rows of one kind feed one block of hidden columns, agents/policies.py:41-61): of the four
    // 16-feature tiles of dW1 a column unit only accumulates the ones with a structural non-zero (two of four on the
    // reference's nets: 72 instead of 80 MFMAs per 16 x 16 tile); the others stay zero, as dx1w1_reduce2_kernel writes them
    const int fm0 = ftm[a * NCU + U0], fm1 = ftm[a * NCU + U1];
    // stationary B operands: bwX[4 j + c] = Wx^T[k = 16 j + 4 kq + c][col]
    float bw0[64], bw1[64];
    {
        const float *src = WxT + (long long)g * kG4 * H + (long long)(4 * kq) * H;
#pragma unroll
        for (int j = 0; j < 16; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                bw0[4 * j + c] = src[(long long)(16 * j + c) * H + col0];
                bw1[4 * j + c] = src[(long long)(16 * j + c) * H + col1];
            }
    }
    f32x4 accW[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) accW[u][ft] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum0 = 0.f, bsum1 = 0.f;
    // staging: thread -> four dZ float4s (rows tid >> 6 + 8 q, 16-byte column tid & 63) and one obs float4
    const int zr = tid >> 6, zc = tid & 63, orow = tid >> 4, oc = tid & 15;
    float4 s0, s1, s2, s3, so;
    auto fetch = [&](long long row0) {
        auto rowc = [&](long long r) { return r < n1 ? r : n1 - 1; };
        s0 = *reinterpret_cast<const float4 *>(dz + rowc(row0 + zr) * kG4 + 4 * zc);
        s1 = *reinterpret_cast<const float4 *>(dz + rowc(row0 + zr + 8) * kG4 + 4 * zc);
        s2 = *reinterpret_cast<const float4 *>(dz + rowc(row0 + zr + 16) * kG4 + 4 * zc);
        s3 = *reinterpret_cast<const float4 *>(dz + rowc(row0 + zr + 24) * kG4 + 4 * zc);
        const float4 o = *reinterpret_cast<const float4 *>(obs + rowc(row0 + orow) * AS + (long long)a * SMAX + 4 * (oc < sq ? oc : 0));
        so = oc < sq ? o : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto put = [&](int buf) {
        float *zb = Az + (long long)buf * 32 * kD1Ld + 4 * zc;
        *reinterpret_cast<float4 *>(zb + (zr) * kD1Ld) = s0;
        *reinterpret_cast<float4 *>(zb + (zr + 8) * kD1Ld) = s1;
        *reinterpret_cast<float4 *>(zb + (zr + 16) * kD1Ld) = s2;
        *reinterpret_cast<float4 *>(zb + (zr + 24) * kD1Ld) = s3;
        *reinterpret_cast<float4 *>(Ob + ((long long)buf * 32 + orow) * kObLd + 4 * oc) = so;
    };
    // one column unit of the chunk in LDS buffer `buf`: both / one of its row tiles
    auto unit = [&](int buf, long long row, const float (&bw)[64], int col, bool r0, bool r1, f32x4 (&aw)[4], float &bs, int fm) {
        // relu mask rows 16 r + 4 kq + i of column col: requested first, consumed after the 64 / 128 MFMAs
        float xm0[4], xm1[4];
        int zq = 0;
        asm volatile("" : "+v"(zq));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            long long ra = row + 4 * kq + i + zq, rb = ra + 16;
            if (ra >= n1) ra = n1 - 1;
            if (rb >= n1) rb = n1 - 1;
            xm0[i] = x1[ra * H + col];
            xm1[i] = x1[rb * H + col];
        }
        f32x4 c0 = f32x4{0.f, 0.f, 0.f, 0.f}, c1 = f32x4{0.f, 0.f, 0.f, 0.f};
        const float4 *A0 = reinterpret_cast<const float4 *>(Az + ((long long)buf * 32 + n) * kD1Ld + 4 * kq);
        const float4 *A1 = reinterpret_cast<const float4 *>(Az + ((long long)buf * 32 + 16 + n) * kD1Ld + 4 * kq);
        if (r0 && r1) {
            float4 p = A0[0], q = A1[0];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float4 pn = A0[4 * (j + 1 < 16 ? j + 1 : j)], qn = A1[4 * (j + 1 < 16 ? j + 1 : j)];
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(p.x, bw[4 * j], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(q.x, bw[4 * j], c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(p.y, bw[4 * j + 1], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(q.y, bw[4 * j + 1], c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(p.z, bw[4 * j + 2], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(q.z, bw[4 * j + 2], c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(p.w, bw[4 * j + 3], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(q.w, bw[4 * j + 3], c1, 0, 0, 0);
                p = pn; q = qn;
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            const float4 *As = r0 ? A0 : A1;
            float4 p = As[0];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float4 pn = As[4 * (j + 1 < 16 ? j + 1 : j)];
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(p.x, bw[4 * j], c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(p.y, bw[4 * j + 1], c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(p.z, bw[4 * j + 2], c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(p.w, bw[4 * j + 3], c0, 0, 0, 0);
                p = pn;
                __builtin_amdgcn_sched_barrier(0);
            }
            if (!r0) { c1 = c0; }
        }
        // dW1 += obs^T dX1 : contraction step i takes rows 16 r + 4 kq + i = the accumulator's own rows
        auto tail = [&](int r, const f32x4 &c, const float (&xm)[4]) {
            const float *Os = Ob + ((long long)buf * 32 + 16 * r + 4 * kq) * kObLd + n;
            // all 16 obs operands first, unconditionally (the registers of the A-operand quads are free by now): with the
            // feature-tile test around every single MFMA each of them sat in a basic block of its own -- LDS read, wait, MFMA --
            // and paid its operand's LDS latency alone (round 5: 16 % of the kernel)
            float os[16];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int ft = 0; ft < 4; ++ft) os[4 * ft + i] = Os[i * kObLd + 16 * ft];
            float d[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                d[i] = (xm[i] > 0.f && row + 16 * r + 4 * kq + i < n1) ? c[i] : 0.f;
                bs += d[i];
            }
#pragma unroll
            for (int ft = 0; ft < 4; ++ft)
                if ((fm >> ft) & 1) {                                   // wave-uniform: one test per feature tile
#pragma unroll
                    for (int i = 0; i < 4; ++i) aw[ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(os[4 * ft + i], d[i], aw[ft], 0, 0, 0);
                }
        };
        if (r0) tail(0, c0, xm0);
        if (r1) tail(1, c1, xm1);
    };
    if (n0 < n1) {
        fetch(n0);
        put(0);
        __syncthreads();
        int buf = 0;
        for (long long row = n0; row < n1; row += 32, buf ^= 1) {
            fetch(row + 32);                                    // lands while this chunk computes
            if (a00 || a01) unit(buf, row, bw0, col0, a00, a01, accW[0], bsum0, fm0);
            if (a10 || a11) unit(buf, row, bw1, col1, a10, a11, accW[1], bsum1, fm1);
            put(buf ^ 1);
            __syncthreads();
        }
    }
    // partial results: slot 0 = row tile 0 (or both tiles of a unit owned by one wave), slot 1 = row tile 1 alone
    auto flush = [&](const f32x4 (&aw)[4], float bs, int col, bool r0, bool r1) {
        float *w0 = w + (long long)((r0 ? 0 : 1) * 65) * H;
        float *wz = w + (long long)65 * H;                      // slot 1, zeroed by the owner of both tiles
#pragma unroll
        for (int ft = 0; ft < 4; ++ft)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int f = 16 * ft + 4 * kq + i;
                w0[(long long)f * H + col] = aw[ft][i];
                if (r0 && r1) wz[(long long)f * H + col] = 0.f;
            }
        bs += __shfl_xor(bs, 16, 64);
        bs += __shfl_xor(bs, 32, 64);
        if (kq == 0) {
            w0[(long long)64 * H + col] = bs;
            if (r0 && r1) wz[(long long)64 * H + col] = 0.f;
        }
    };
    if (a00 || a01) flush(accW[0], bsum0, col0, a00, a01);
    if (a10 || a11) flush(accW[1], bsum1, col1, a10, a11);
}

// grads[g][oW1 .. +SMAX*H) and [ob1 .. +H) = sum over (split, row-tile slot) in that order; structural zeros of W1 applied
__global__ void dx1w1_reduce2_kernel(const float *__restrict__ ws, int G, int S, int H, int SMAX, const int16_t *__restrict__ rr,
                                     float *__restrict__ grads, long long stride, long long oW1, long long ob1) {
    const long long per = (long long)65 * H, out = (long long)(SMAX + 1) * H;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= out * G) return;
    const long long g = i / out, j = i % out;
    const int f = (int)(j / H), n = (int)(j % H);
    const long long src = f < SMAX ? j : (long long)64 * H + n;
    float acc = 0.f;
    for (int s = 0; s < S; ++s) {
        acc += ws[(((long long)s * G + g) * 2 + 0) * per + src];
        acc += ws[(((long long)s * G + g) * 2 + 1) * per + src];
    }
    if (f < SMAX) {
        const int16_t *q = rr + ((g >> 1) * SMAX + f) * 2;
        if (n < q[0] || n >= q[1]) acc = 0.f;
        grads[g * stride + oW1 + j] = acc;
    } else {
        grads[g * stride + ob1 + n] = acc;
    }
}

// ------------------------------------------------------------------------------------------------
// FcACPolicy (agents/policies.py:214-256; BASELINE configs[1]): both layers' weight gradients of one tower in ONE pass over
// the n-step batch.  The second layer is fc(H -> 64) + relu, so its pre-activation gradient dZ [N][64] is what head_bwd2
// leaves in dH (relu' of the 64 units folded in); then
//   dWfc = X1^T dZ,  dbfc = colsum(dZ),
//   dX1  = (dZ Wfc^T) * relu'(X1)     (16 x 16 tiles, never leaves the registers)
//   dW1  = obs^T dX1,  db1 = colsum(dX1)
// -- dx1w1_kernel2 with a 64-deep contraction (16 stationary registers per column unit instead of 64) plus the X1 chunk in
// LDS, which serves the relu mask (from HBM per lane before) and is the A operand of the dWfc tiles: H/16 x 4 tiles of
// 16 x 16, wave w owns unit column w & 3 and every second hidden unit from w >> 2 (5 tiles at H = 160, K = the chunk's 32
// rows).  Replaces three grouped GEMMs (dWfc, dX1 with its 1-GB round trip through HBM, dW1 at a quarter-empty tile).
// Deterministic like its model: fixed row splits, partial sums folded in split order by fc_bwd_reduce_kernel.
// ------------------------------------------------------------------------------------------------
constexpr int kFbLdz = kL + 4;   // dZ chunk [32 rows][64 k + 4]: row-major like HBM, read as 16-byte A-operand quads

template <int NCU>   // H / 16
__global__ void __launch_bounds__(512, 1)
fc_bwd_kernel(const float *__restrict__ dZ, const float *__restrict__ X1, const float *__restrict__ WxT,
              const float *__restrict__ obs, long long N, int G, int S, long long rows_per_split, int A, int SMAX,
              float *__restrict__ ws, const int *__restrict__ ftm) {
    constexpr int H = 16 * NCU, NHU = 2 * NCU;                  // half units (column unit x row tile) per chunk
    constexpr int CHI = (NHU + 7) / 8, CLO = NHU / 8;           // tiles of waves 0..3 / 4..7
    static_assert(NHU % 8 == 0 || NHU % 8 == 4, "H must be a multiple of 32");
    constexpr int LDX = H + 16;                                 // X1 chunk rows: (LDX % 32) == 16 -> the four k rows of an MFMA step sit on two bank halves
    constexpr int NF = NCU / 2;                                 // dWfc tiles per wave
    constexpr int XQ = H / 4, NXQ = 32 * XQ, NXL = (NXQ + 511) / 512;
    static_assert(NXL <= 3, "X1 staging slots");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float *Az = (float *)smem_raw;                              // [2][32][kFbLdz]
    float *Ob = Az + 2 * 32 * kFbLdz;                           // [2][32][kObLd]
    float *Xs = Ob + 2 * 32 * kObLd;                            // [2][32][LDX]
    const int g = blockIdx.x % G, sp = blockIdx.x / G, a = g >> 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), n = lane & 15, kq = lane >> 4;
    const long long n0 = (long long)sp * rows_per_split;
    long long n1 = n0 + rows_per_split;
    if (n1 > N) n1 = N;
    const float *dz = dZ + (long long)g * N * kL, *x1 = X1 + (long long)g * N * H;
    const long long AS = (long long)A * SMAX;
    const int sq = SMAX >> 2;
    constexpr long long kPer1 = (long long)65 * H, kPerF = (long long)(H + 1) * kL;
    float *w = ws + ((long long)sp * G + g) * (2 * kPer1 + kPerF);
    float *wF = w + 2 * kPer1;
    // ---- first layer: this wave's half units [st, st + cnt): unit = h >> 1, row tile = h & 1 (as in dx1w1_kernel2)
    const int st = wave < 4 ? wave * CHI : 4 * CHI + (wave - 4) * CLO, cnt = wave < 4 ? CHI : CLO;
    const int U0 = st >> 1;
    const bool a00 = 2 * U0 >= st, a01 = 2 * U0 + 1 < st + cnt, a10 = 2 * U0 + 2 < st + cnt, a11 = 2 * U0 + 3 < st + cnt;
    const int U1 = U0 + 1 < NCU ? U0 + 1 : NCU - 1;
    const int col0 = 16 * U0 + n, col1 = 16 * U1 + n;
    const int fm0 = ftm[a * NCU + U0], fm1 = ftm[a * NCU + U1];
    // stationary B operands: bwX[4 j + c] = Wfc^T[k = 16 j + 4 kq + c][col]
    float bw0[16], bw1[16];
    {
        const float *src = WxT + (long long)g * kL * H + (long long)(4 * kq) * H;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                bw0[4 * j + c] = src[(long long)(16 * j + c) * H + col0];
                bw1[4 * j + c] = src[(long long)(16 * j + c) * H + col1];
            }
    }
    f32x4 accW[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) accW[u][ft] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum0 = 0.f, bsum1 = 0.f;
    // ---- second layer: dWfc tiles (hidden unit hu0 + 2 j, unit column c2), dbfc from the waves with hu0 == 0
    const int c2 = wave & 3, hu0 = wave >> 2;
    f32x4 accF[NF];
#pragma unroll
    for (int j = 0; j < NF; ++j) accF[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsumF = 0.f;
    // ---- staging: thread -> one dZ float4 (row tid >> 4, 16-byte column tid & 15), one obs float4, up to three X1 float4s.
    // Rows past the split are stored as ZERO dZ rows: they then contribute nothing to either layer.
    const int zr = tid >> 4, zc = tid & 15;
    struct XSlot { int row, col; bool ok; };
    auto xslot = [&](int q) {
        XSlot o;
        int idx = tid + 512 * q;
        o.ok = idx < NXQ;
        if (!o.ok) idx = NXQ - 1;
        o.row = idx / XQ; o.col = 4 * (idx % XQ);
        return o;
    };
    const XSlot x0 = xslot(0), x1s = xslot(1), x2s = xslot(2);
    float4 sz, so, sx0, sx1, sx2;
    auto fetch = [&](long long row0) {
        auto rowc = [&](long long r) { return r < n1 ? r : n1 - 1; };
        const float4 z = *reinterpret_cast<const float4 *>(dz + rowc(row0 + zr) * kL + 4 * zc);
        sz = row0 + zr < n1 ? z : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 o = *reinterpret_cast<const float4 *>(obs + rowc(row0 + zr) * AS + (long long)a * SMAX + 4 * (zc < sq ? zc : 0));
        so = zc < sq ? o : make_float4(0.f, 0.f, 0.f, 0.f);
        sx0 = *reinterpret_cast<const float4 *>(x1 + rowc(row0 + x0.row) * H + x0.col);
        if (NXL > 1) sx1 = *reinterpret_cast<const float4 *>(x1 + rowc(row0 + x1s.row) * H + x1s.col);
        if (NXL > 2) sx2 = *reinterpret_cast<const float4 *>(x1 + rowc(row0 + x2s.row) * H + x2s.col);
    };
    auto put = [&](int buf) {
        *reinterpret_cast<float4 *>(Az + ((long long)buf * 32 + zr) * kFbLdz + 4 * zc) = sz;
        *reinterpret_cast<float4 *>(Ob + ((long long)buf * 32 + zr) * kObLd + 4 * zc) = so;
        float *xb = Xs + (long long)buf * 32 * LDX;
        if (x0.ok) *reinterpret_cast<float4 *>(xb + x0.row * LDX + x0.col) = sx0;
        if (NXL > 1 && x1s.ok) *reinterpret_cast<float4 *>(xb + x1s.row * LDX + x1s.col) = sx1;
        if (NXL > 2 && x2s.ok) *reinterpret_cast<float4 *>(xb + x2s.row * LDX + x2s.col) = sx2;
    };
    // one column unit of the chunk in LDS buffer `buf`: both / one of its row tiles
    auto unit = [&](int buf, long long row, const float (&bw)[16], int col, bool r0, bool r1, f32x4 (&aw)[4], float &bs, int fm) {
        f32x4 c0 = f32x4{0.f, 0.f, 0.f, 0.f}, c1 = f32x4{0.f, 0.f, 0.f, 0.f};
        const float4 *A0 = reinterpret_cast<const float4 *>(Az + ((long long)buf * 32 + n) * kFbLdz + 4 * kq);
        const float4 *A1 = reinterpret_cast<const float4 *>(Az + ((long long)buf * 32 + 16 + n) * kFbLdz + 4 * kq);
        // relu mask rows 16 r + 4 kq + i of column col, from the X1 chunk
        const float *Xc = Xs + ((long long)buf * 32 + 4 * kq) * LDX + col;
        float xm0[4], xm1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { xm0[i] = Xc[i * LDX]; xm1[i] = Xc[(16 + i) * LDX]; }
        if (r0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 p = A0[4 * j];
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(p.x, bw[4 * j], c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(p.y, bw[4 * j + 1], c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(p.z, bw[4 * j + 2], c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(p.w, bw[4 * j + 3], c0, 0, 0, 0);
            }
        }
        if (r1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 q = A1[4 * j];
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(q.x, bw[4 * j], c1, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(q.y, bw[4 * j + 1], c1, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(q.z, bw[4 * j + 2], c1, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(q.w, bw[4 * j + 3], c1, 0, 0, 0);
            }
        }
        // dW1 += obs^T dX1 : contraction step i takes rows 16 r + 4 kq + i = the accumulator's own rows
        auto tail = [&](int r, const f32x4 &c, const float (&xm)[4]) {
            const float *Os = Ob + ((long long)buf * 32 + 16 * r + 4 * kq) * kObLd + n;
            float os[16];                                      // all obs operands first (see dx1w1_kernel2: one test per feature tile)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int ft = 0; ft < 4; ++ft) os[4 * ft + i] = Os[i * kObLd + 16 * ft];
            float d[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                d[i] = xm[i] > 0.f ? c[i] : 0.f;               // rows past the split: c is exactly zero (zero dZ rows)
                bs += d[i];
            }
#pragma unroll
            for (int ft = 0; ft < 4; ++ft)
                if ((fm >> ft) & 1) {                                   // wave-uniform
#pragma unroll
                    for (int i = 0; i < 4; ++i) aw[ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(os[4 * ft + i], d[i], aw[ft], 0, 0, 0);
                }
        };
        if (r0) tail(0, c0, xm0);
        if (r1) tail(1, c1, xm1);
    };
    if (n0 < n1) {
        fetch(n0);
        put(0);
        __syncthreads();
        int buf = 0;
        for (long long row = n0; row < n1; row += 32, buf ^= 1) {
            fetch(row + 32);                                    // lands while this chunk computes
            {   // dWfc += X1^T dZ: A[m = hidden][k = row] from the X1 chunk, B[k = row][n = unit] from the dZ chunk
                const float *Bz = Az + ((long long)buf * 32 + kq) * kFbLdz + 16 * c2 + n;
                const float *Ax = Xs + ((long long)buf * 32 + kq) * LDX + 16 * hu0 + n;
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const float b = Bz[4 * s * kFbLdz];
                    bsumF += b;
#pragma unroll
                    for (int j = 0; j < NF; ++j)
                        accF[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ax[4 * s * LDX + 32 * j], b, accF[j], 0, 0, 0);
                }
            }
            if (a00 || a01) unit(buf, row, bw0, col0, a00, a01, accW[0], bsum0, fm0);
            if (a10 || a11) unit(buf, row, bw1, col1, a10, a11, accW[1], bsum1, fm1);
            put(buf ^ 1);
            __syncthreads();
        }
    }
    // partial results of the first layer: slot 0 = row tile 0 (or both tiles of a unit owned by one wave), slot 1 = row tile 1 alone
    auto flush = [&](const f32x4 (&aw)[4], float bs, int col, bool r0, bool r1) {
        float *w0 = w + (long long)((r0 ? 0 : 1) * 65) * H;
        float *wz = w + (long long)65 * H;                      // slot 1, zeroed by the owner of both tiles
#pragma unroll
        for (int ft = 0; ft < 4; ++ft)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int f = 16 * ft + 4 * kq + i;
                w0[(long long)f * H + col] = aw[ft][i];
                if (r0 && r1) wz[(long long)f * H + col] = 0.f;
            }
        bs += __shfl_xor(bs, 16, 64);
        bs += __shfl_xor(bs, 32, 64);
        if (kq == 0) {
            w0[(long long)64 * H + col] = bs;
            if (r0 && r1) wz[(long long)64 * H + col] = 0.f;
        }
    };
    if (a00 || a01) flush(accW[0], bsum0, col0, a00, a01);
    if (a10 || a11) flush(accW[1], bsum1, col1, a10, a11);
    // second layer: accF[j][i] = dWfc[hidden 16 (hu0 + 2 j) + 4 kq + i][unit 16 c2 + n]
#pragma unroll
    for (int j = 0; j < NF; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            wF[(long long)(16 * (hu0 + 2 * j) + 4 * kq + i) * kL + 16 * c2 + n] = accF[j][i];
    bsumF += __shfl_xor(bsumF, 16, 64);
    bsumF += __shfl_xor(bsumF, 32, 64);
    if (kq == 0 && hu0 == 0) wF[(long long)H * kL + 16 * c2 + n] = bsumF;
}

// grads[g]: W1 | b1 (sum over (split, row-tile slot), structural zeros of W1 applied) and Wfc | bfc (sum over splits), in order
__global__ void fc_bwd_reduce_kernel(const float *__restrict__ ws, int G, int S, int H, int SMAX, const int16_t *__restrict__ rr,
                                     float *__restrict__ grads, long long stride, long long oW1, long long ob1, long long oWx) {
    const long long per1 = (long long)65 * H, perF = (long long)(H + 1) * kL, per = 2 * per1 + perF;
    const long long out1 = (long long)(SMAX + 1) * H, out = out1 + perF;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= out * G) return;
    const long long g = i / out, j = i % out;
    float acc = 0.f;
    if (j < out1) {
        const int f = (int)(j / H), n = (int)(j % H);
        const long long src = f < SMAX ? j : (long long)64 * H + n;
        for (int s = 0; s < S; ++s) {
            const float *b = ws + ((long long)s * G + g) * per;
            acc += b[src];
            acc += b[per1 + src];
        }
        if (f < SMAX) {
            const int16_t *q = rr + ((g >> 1) * SMAX + f) * 2;
            if (n < q[0] || n >= q[1]) acc = 0.f;
            grads[g * stride + oW1 + j] = acc;
        } else {
            grads[g * stride + ob1 + n] = acc;
        }
    } else {
        const long long jj = j - out1;                         // Wfc [H][64] | bfc [64] are contiguous in the layout
        for (int s = 0; s < S; ++s) acc += ws[((long long)s * G + g) * per + 2 * per1 + jj];
        grads[g * stride + oWx + jj] = acc;
    }
}

struct tsc_model {
    Layout lay;
    int E, T, device;
    double gamma, rnorm, rclip, vcoef, max_norm, alpha, eps;
    hipStream_t stream;
    std::vector<void *> allocs;
    int *n_act;
    int16_t *rowrange;          // [A][SMAX][2]
    int *wgmap; int wgmap_S, wgmap_n, xcd_map_on;   // ws forward: blockIdx -> (tower << 8 | split), XCD-affine (TSC_FWD_XCD=0: off)
    int dbg_tid;                // thread of workgroup 0 that writes the clock stamps (TSC_DBG_THREAD)
    int *ftmask;                // [A][H / 16]: 16-feature tiles of dW1 with a structural non-zero in column unit U (dx1w1_kernel2)
    int *krange;                // [A][8][2]: first-layer MFMA steps (2 obs rows each) that feed hidden column tile w (ws forward)
    float *params, *grads, *ms, *WxT;
    float *Wg;                  // gate-interleaved copy of [Wx ; Wh] for the fused forward (interleave_gates_kernel)
    int wg_dirty;
    float *state_fw, *state_bw, *state_tmp;     // [G][E][128]
    // rollout (on-policy buffer)
    float *r_obs; int *r_act; double *r_rew; float *r_val; uint8_t *r_done;   // done [T+1][E]
    float *Rs, *Advs;
    // activations
    float *X1, *Z, *Hh, *Cc, *Hp, *dHh, *dL;
    double *norm2, *stats, *norm_part;
    float *ws, *wsc;            // split-K workspace
    size_t ws_floats, wsc_floats;
    size_t lds_fwd, lds_fused, lds_ws;
    int fused_fwd;
    int fused_dw;               // dWx | dWh | dbl in one pass (dwxh_kernel)
    int fc_mfma;                // FcACPolicy rollout forward on the matrix cores (TSC_FC_MFMA=0: the per-thread kernel)
    int fused_dx;               // dX1 in registers, dW1 | db1 in the same pass (dx1w1_kernel2)
    int fused_fc;               // FcACPolicy: dWfc | dbfc | dW1 | db1 in one pass, dX1 in registers (fc_bwd_kernel)
    int inplace;                // the running rollout is written straight into the buffer's slots (tsc_model_rollout_slot): slot T -> 0 carry
    int cached_next;            // next rollout slot whose activations the fused forward will cache; T = all cached
    long long *dbg;
    long long nparam;
};

namespace {

int gemm(tsc_model *m, int kid, bool tn, int epi, int groups, int M, int N, int K, const float *A, long long sA, int lda,
         int gdivA, const float *B, long long sB, int ldb, float *C, long long sC, int ldc, const float *bias,
         long long sBias, const float *aux, long long sAux, int ldaux, const int16_t *rr, long long sRR,
         float *colsum, long long sColsum) {
    GemmArgs a;
    a.A = A; a.B = B; a.C = C; a.bias = bias; a.aux = aux; a.rr = rr; a.colsum = colsum;
    a.sA = sA; a.sB = sB; a.sC = sC; a.sBias = sBias; a.sAux = sAux; a.sRR = sRR; a.sColsum = sColsum;
    a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ldaux = ldaux; a.M = M; a.N = N; a.K = K; a.gdivA = gdivA;
    tsc::plan_splitk(a, groups, tn ? m->ws : nullptr, m->wsc, m->ws_floats, m->wsc_floats);
    tsc::ProfScope ps(kid, m->stream);
    tsc::launch_gemm_dyn(tn, epi, a, groups, m->stream);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// X1 = relu(obs W1 + b1) ; then LSTM: Z = X1 Wx + bl,  FC policy: Hh = relu(X1 Wfc + bfc)
int dense_forward(tsc_model *m, const float *obs, long long rows, float *X1, float *Z) {
    const Layout &L = m->lay;
    const int AS = L.A * L.SMAX;
    if (gemm(m, tsc::KID_FC_GEMM, false, tsc::EPI_BIAS_RELU, L.G, (int)rows, L.H, L.SMAX, obs, L.SMAX, AS, 2, m->params + L.oW1,
             L.stride, L.H, X1, rows * L.H, L.H, m->params + L.ob1, L.stride, nullptr, 0, 0, nullptr, 0, nullptr, 0))
        return 1;
    if (gemm(m, tsc::KID_ZX_GEMM, false, L.fc ? tsc::EPI_BIAS_RELU : tsc::EPI_BIAS, L.G, (int)rows, L.NZ, L.H, X1, rows * L.H, L.H, 1,
             m->params + L.oWx, L.stride, L.NZ, Z, rows * L.NZ, L.NZ, m->params + L.obl, L.stride, nullptr, 0, 0, nullptr, 0,
             nullptr, 0))
        return 1;
    return 0;
}

}  // namespace

extern "C" {

#define MALLOC(ptr, T, count)                                                          \
    do {                                                                               \
        TSC_HIP(hipMalloc((void **)&(ptr), sizeof(T) * (size_t)(count)));              \
        TSC_HIP(hipMemset((ptr), 0, sizeof(T) * (size_t)(count)));                     \
        m->allocs.push_back((void *)(ptr));                                            \
    } while (0)

int tsc_model_create(const tsc_model_cfg *cfg, int32_t n_env, int32_t device, tsc_model **out) {
    if (!cfg || !out || n_env <= 0) return tsc::fail("tsc_model_create: bad arguments");
    if (cfg->n_lstm != kL) return tsc::fail("tsc_model_create: num_lstm must be %d", kL);
    if (cfg->policy_kind != 0 && cfg->policy_kind != 1) return tsc::fail("tsc_model_create: policy_kind must be 0 (lstm) or 1 (fc)");
    if (cfg->a_max > kOut) return tsc::fail("tsc_model_create: a_max %d > %d", cfg->a_max, kOut);
    if (cfg->s_max % 4) return tsc::fail("tsc_model_create: s_max must be a multiple of 4");
    TSC_HIP(hipSetDevice(device));
    tsc_model *m = new tsc_model();
    tsc::CreateGuard<tsc_model, tsc_model_destroy> guard(m);        // an error return below frees the handle and its buffers
    m->device = device; m->stream = nullptr; m->E = n_env; m->T = cfg->n_step;
    m->gamma = cfg->gamma; m->rnorm = cfg->reward_norm; m->rclip = cfg->reward_clip; m->vcoef = cfg->value_coef;
    m->max_norm = cfg->max_grad_norm; m->alpha = cfg->rmsp_alpha; m->eps = cfg->rmsp_epsilon;
    Layout &L = m->lay;
    L.A = cfg->n_agent; L.G = 2 * L.A; L.SMAX = cfg->s_max; L.AMAX = cfg->a_max;
    L.H = cfg->n_fc_wave + cfg->n_fc_fp + cfg->n_fc_wait;
    if (L.H % 4) return tsc::fail("tsc_model_create: hidden width must be a multiple of 4");
    L.fc = cfg->policy_kind; L.NZ = L.fc ? kL : kG4;
    L.oW1 = 0; L.ob1 = (long long)L.SMAX * L.H; L.oWx = L.ob1 + L.H; L.oWh = L.oWx + (long long)L.H * L.NZ;
    L.obl = L.oWh + (L.fc ? 0 : (long long)kL * kG4); L.oWo = L.obl + L.NZ; L.obo = L.oWo + kL * kOut; L.stride = L.obo + kOut;
    m->nparam = L.stride * L.G;
    // structural mask of W1: obs row j of agent a feeds hidden columns [lo, hi)
    std::vector<int16_t> rr((size_t)L.A * L.SMAX * 2, 0);
    const int cw = cfg->n_fc_wave, cf = cfg->n_fc_fp, ct = cfg->n_fc_wait;
    for (int a = 0; a < L.A; ++a) {
        const int nw = cfg->n_wave[a], nt = cfg->n_wait[a], nf = cfg->n_fp[a];
        if (nw + nt + nf > L.SMAX) return tsc::fail("tsc_model_create: agent %d obs wider than s_max", a);
        for (int j = 0; j < L.SMAX; ++j) {
            int lo = 0, hi = 0;
            if (j < nw) { lo = 0; hi = cw; }                                  // fcw
            else if (j < nw + nt) { lo = cw + cf; hi = cw + cf + ct; }        // fct (after fcf in the concat)
            else if (j < nw + nt + nf) { lo = cw; hi = cw + cf; }             // fcf
            rr[((size_t)a * L.SMAX + j) * 2] = (int16_t)lo;
            rr[((size_t)a * L.SMAX + j) * 2 + 1] = (int16_t)hi;
        }
    }
    TSC_HIP(tsc::upload<int16_t>(&m->rowrange, rr.data(), rr.size())); m->allocs.push_back(m->rowrange);
    {   // per (agent, 32-column tile of the first layer): the obs rows with a structural non-zero in the tile, in MFMA steps
        std::vector<int> kr((size_t)L.A * 8 * 2, 0);
        for (int a = 0; a < L.A; ++a)
            for (int w = 0; w < 8; ++w) {
                int j0 = L.SMAX, j1 = 0;
                for (int j = 0; j < L.SMAX; ++j) {
                    const int lo = rr[((size_t)a * L.SMAX + j) * 2], hi = rr[((size_t)a * L.SMAX + j) * 2 + 1];
                    if (lo < 32 * w + 32 && hi > 32 * w) { if (j < j0) j0 = j; j1 = j + 1; }
                }
                if (j1 <= j0) { j0 = 0; j1 = 0; }
                kr[((size_t)a * 8 + w) * 2] = j0 / 2; kr[((size_t)a * 8 + w) * 2 + 1] = (j1 + 1) / 2;
            }
        TSC_HIP(tsc::upload<int>(&m->krange, kr.data(), kr.size())); m->allocs.push_back(m->krange);
    }
    {   // per (agent, 16-column unit): which 16-row tiles of W1 hold a structural non-zero
        const int ncu = (L.H + 15) / 16;
        std::vector<int> fm((size_t)L.A * ncu, 0);
        for (int a = 0; a < L.A; ++a)
            for (int u = 0; u < ncu; ++u)
                for (int j = 0; j < L.SMAX && j < 64; ++j) {
                    const int lo = rr[((size_t)a * L.SMAX + j) * 2], hi = rr[((size_t)a * L.SMAX + j) * 2 + 1];
                    if (lo < 16 * u + 16 && hi > 16 * u) fm[(size_t)a * ncu + u] |= 1 << (j >> 4);
                }
        TSC_HIP(tsc::upload<int>(&m->ftmask, fm.data(), fm.size())); m->allocs.push_back(m->ftmask);
    }
    TSC_HIP(tsc::upload<int>(&m->n_act, cfg->n_act, L.A)); m->allocs.push_back(m->n_act);
    const long long E = n_env, T = m->T, N = E * T, G = L.G, A = L.A;
    MALLOC(m->params, float, m->nparam); MALLOC(m->grads, float, m->nparam); MALLOC(m->ms, float, m->nparam);
    MALLOC(m->WxT, float, G * L.H * kG4);
    MALLOC(m->Wg, float, G * (L.H + kL) * kG4);
    m->wg_dirty = 1;
    MALLOC(m->state_fw, float, G * E * 2 * kL); MALLOC(m->state_bw, float, G * E * 2 * kL);
    MALLOC(m->state_tmp, float, G * E * 2 * kL);
    MALLOC(m->r_obs, float, (N + E) * A * L.SMAX);          // T + 1 slots: slot t + 1 receives the obs the env returns at step t
    MALLOC(m->r_act, int, N * A); MALLOC(m->r_rew, double, N * A);
    MALLOC(m->r_val, float, N * A); MALLOC(m->r_done, uint8_t, (T + 1) * E);
    MALLOC(m->Rs, float, N * A); MALLOC(m->Advs, float, N * A);
    MALLOC(m->X1, float, G * N * L.H + 32 * L.H);        // + one tile nobody reads: the ws forward's branch-free X1 store when the cache is off
    MALLOC(m->Z, float, G * N * kG4);
    MALLOC(m->Hh, float, G * N * kL); MALLOC(m->Cc, float, G * N * kL); MALLOC(m->Hp, float, G * N * kL);
    MALLOC(m->dHh, float, G * N * kL); MALLOC(m->dL, float, G * N * kOut);
    MALLOC(m->norm2, double, A); MALLOC(m->stats, double, A * 4); MALLOC(m->norm_part, double, A * kNormParts);
    m->ws_floats = (size_t)48 << 20; m->wsc_floats = (size_t)1 << 20;      // 192 MiB + 4 MiB
    MALLOC(m->ws, float, m->ws_floats); MALLOC(m->wsc, float, m->wsc_floats);
    m->lds_fwd = sizeof(float) * (64 * kWhLd + 64 * kHsLd);
    TSC_HIP(hipFuncSetAttribute((const void *)lstm_bwd2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(float) * 32 * (kDz2Ld + kDh3Ld))));
    TSC_HIP(hipFuncSetAttribute((const void *)lstm_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)m->lds_fwd));
    TSC_HIP(hipFuncSetAttribute((const void *)lstm_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)m->lds_fwd));
    m->dbg = nullptr;
    m->cached_next = 0;
    m->inplace = 0;
    m->lds_fused = sizeof(float) * ((size_t)(L.H + 64) * kXLd + kL * kOut + kOut + 8);   // activations + head weights
    m->fused_fwd = !L.fc && (L.H % 32 == 0) && (L.SMAX <= 64) && (L.SMAX % 4 == 0) && ((L.H + 64) % 8 == 0) && m->lds_fused <= 160 * 1024;
    if (const char *ev = getenv("TSC_DBG_THREAD")) m->dbg_tid = atoi(ev);
    m->fc_mfma = 1;
    if (const char *ev = getenv("TSC_FC_MFMA")) m->fc_mfma = atoi(ev);
    m->xcd_map_on = 1;
    if (const char *ev = getenv("TSC_FWD_XCD")) m->xcd_map_on = atoi(ev);
    m->fused_dw = !L.fc && (L.H == 224 || L.H == 160 || L.H == 192 || L.H == 128);
    if (const char *ev = getenv("TSC_UNFUSED_DW")) if (atoi(ev)) m->fused_dw = 0;
    m->fused_dx = !L.fc && (L.H == 224 || L.H == 160 || L.H == 192 || L.H == 128) && L.SMAX <= 64 && L.SMAX % 4 == 0;
    if (const char *ev = getenv("TSC_UNFUSED_DX")) if (atoi(ev)) m->fused_dx = 0;
    m->fused_fc = L.fc && (L.H == 160 || L.H == 128) && L.SMAX <= 64 && L.SMAX % 4 == 0;
    if (const char *ev = getenv("TSC_UNFUSED_DX")) if (atoi(ev)) m->fused_fc = 0;
    if (m->fused_fc) {
        const int lds = (int)(sizeof(float) * (2 * 32 * kFbLdz + 2 * 32 * kObLd + 2 * 32 * (L.H + 16)));
        TSC_HIP(hipFuncSetAttribute((const void *)fc_bwd_kernel<10>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        TSC_HIP(hipFuncSetAttribute((const void *)fc_bwd_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    }
    if (m->fused_dx) {
        const int lds = (int)(sizeof(float) * (2 * 32 * kD1Ld + 2 * 32 * kObLd));
        TSC_HIP(hipFuncSetAttribute((const void *)dx1w1_kernel2<14>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        TSC_HIP(hipFuncSetAttribute((const void *)dx1w1_kernel2<10>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        TSC_HIP(hipFuncSetAttribute((const void *)dx1w1_kernel2<12>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        TSC_HIP(hipFuncSetAttribute((const void *)dx1w1_kernel2<8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    }
    m->lds_ws = sizeof(float) * ((size_t)32 * (L.H + 64 + 4) + 64 * kWsLdx + (size_t)kL * kWsLdg + (size_t)kG4 * kWsLdg + 528 + (kWsBuf + 2) * 32 * kOut + (size_t)L.SMAX * L.H);
    if (m->fused_fwd && (L.H == 224 || L.H == 160 || L.H == 192 || L.H == 128) && m->lds_ws <= 160 * 1024) {
        // weight-stationary variant (TSC_FWD_WS=0 falls back to the tile-per-workgroup kernel)
        const char *ev = getenv("TSC_FWD_WS");
        if (!(ev && ev[0] == '0')) m->fused_fwd = 2;
        TSC_HIP(hipFuncSetAttribute((const void *)policy_fwd_ws_kernel<144>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)m->lds_ws));
        TSC_HIP(hipFuncSetAttribute((const void *)policy_fwd_ws_kernel<112>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)m->lds_ws));
        TSC_HIP(hipFuncSetAttribute((const void *)policy_fwd_ws_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)m->lds_ws));
        TSC_HIP(hipFuncSetAttribute((const void *)policy_fwd_ws_kernel<96>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)m->lds_ws));
    }
    if (m->fused_fwd)
        TSC_HIP(hipFuncSetAttribute((const void *)policy_fwd_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)m->lds_fused));
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((m->nparam + 255) / 256)), dim3(256), 0, 0, m->ms, m->nparam, 1.0f);
    TSC_HIP(hipDeviceSynchronize());
    *out = guard.release();
    return 0;
}

int tsc_model_destroy(tsc_model *m) {
    if (!m) return 0;
    (void)hipSetDevice(m->device);
    for (void *p : m->allocs) (void)hipFree(p);
    if (m->wgmap) (void)hipFree(m->wgmap);
    delete m;
    return 0;
}

int tsc_model_set_stream(tsc_model *m, void *s) {
    if (!m) return tsc::fail("null handle");
    m->stream = (hipStream_t)s;
    return 0;
}

int tsc_model_layout(tsc_model *m, int64_t out[12]) {
    if (!m || !out) return tsc::fail("tsc_model_layout: bad arguments");
    const Layout &L = m->lay;
    const int64_t v[12] = {L.G, L.stride, L.H, kL, L.oW1, L.ob1, L.oWx, L.oWh, L.obl, L.oWo, L.obo, kOut};
    for (int i = 0; i < 12; ++i) out[i] = v[i];
    return 0;
}

int tsc_model_set_params(tsc_model *m, const float *h) {
    if (!m || !h) return tsc::fail("tsc_model_set_params: bad arguments");
    m->cached_next = -1;                              // activations cached under the old parameters are stale
    m->wg_dirty = 1;
    TSC_HIP(hipStreamSynchronize(m->stream));
    TSC_HIP(hipMemcpy(m->params, h, sizeof(float) * m->nparam, hipMemcpyHostToDevice));
    return 0;
}
int tsc_model_reset_opt_state(tsc_model *m) {
    if (!m) return tsc::fail("null handle");
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((m->nparam + 255) / 256)), dim3(256), 0, m->stream, m->ms, m->nparam, 1.0f);
    TSC_HIP(hipGetLastError());
    TSC_HIP(hipStreamSynchronize(m->stream));
    return 0;
}
int tsc_model_get_params(tsc_model *m, float *h) {
    if (!m || !h) return tsc::fail("tsc_model_get_params: bad arguments");
    TSC_HIP(hipStreamSynchronize(m->stream));
    TSC_HIP(hipMemcpy(h, m->params, sizeof(float) * m->nparam, hipMemcpyDeviceToHost));
    return 0;
}
int tsc_model_get_opt_state(tsc_model *m, float *h) {
    if (!m || !h) return tsc::fail("tsc_model_get_opt_state: bad arguments");
    TSC_HIP(hipStreamSynchronize(m->stream));
    TSC_HIP(hipMemcpy(h, m->ms, sizeof(float) * m->nparam, hipMemcpyDeviceToHost));
    return 0;
}
int tsc_model_set_opt_state(tsc_model *m, const float *h) {
    if (!m || !h) return tsc::fail("tsc_model_set_opt_state: bad arguments");
    TSC_HIP(hipStreamSynchronize(m->stream));
    TSC_HIP(hipMemcpy(m->ms, h, sizeof(float) * m->nparam, hipMemcpyHostToDevice));
    return 0;
}

int tsc_model_reset(tsc_model *m) {
    if (!m) return tsc::fail("null handle");
    const size_t b = sizeof(float) * (size_t)m->lay.G * m->E * 2 * kL;
    TSC_HIP(hipMemsetAsync(m->state_fw, 0, b, m->stream));
    TSC_HIP(hipMemsetAsync(m->state_bw, 0, b, m->stream));
    return 0;
}

static int model_forward(tsc_model *m, const float *obs, const uint8_t *done, float *pi, float *v, int32_t advance,
                         int32_t *action, uint64_t seed, uint64_t step, int32_t tslot) {
    if (!m || !obs || !done || !pi || !v) return tsc::fail("tsc_model_forward: bad arguments");
    const Layout &L = m->lay;
    const int E = m->E;
    // activation cache: valid only if slots 0..T-1 are filled in order by advancing forwards
    if (advance) {
        if (tslot == 0) m->cached_next = 0;         // slot 0 opens a rollout: whatever invalidated the cache before is history
        const bool fc_cache = L.fc && (L.H == 160 || L.H == 128) && L.SMAX <= 64 && L.AMAX <= kOut && m->fc_mfma;   // policy_fwd_fc_mfma_kernel
        if ((m->fused_fwd || fc_cache) && tslot >= 0 && tslot == m->cached_next && tslot < m->T) m->cached_next = tslot + 1;
        else { m->cached_next = -1; tslot = -1; }
    } else {
        tslot = -1;
    }
    if (m->fused_fwd) {
        const int n_tiles = (E + 63) / 64, per_xcd = (L.G + 7) / 8;
        if (m->wg_dirty) {                                  // parameters changed since the re-laid-out copy was made
            const long long tot = (long long)L.G * (L.H + kL) * kG4;
            if (m->fused_fwd == 1)
                hipLaunchKernelGGL(interleave_gates_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, m->stream, m->params, L, m->Wg);
            else
                hipLaunchKernelGGL(register_order_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, m->stream, m->params, L, m->Wg);
            m->wg_dirty = 0;
        }
        tsc::ProfScope ps(tsc::KID_FUSED_FWD, m->stream);
        if (m->fused_fwd == 2) {
            int S = 256 / L.G;
            if (S < 1) S = 1;
            if (S > (E + 31) / 32) S = (E + 31) / 32;
            // XCD-affine numbering: towers are placed whole on the XCD with the most free slots (cap = ceil(G S / 8) per XCD), the
            // last ones are split over what is left; built once per (S), ids without work exit at once
            const int *wgm = nullptr;
            unsigned nwg = (unsigned)(L.G * S);
            if (m->xcd_map_on && S > 1 && S < 256) {
                if (m->wgmap_S != S) {
                    const int cap = (L.G * S + 7) / 8;
                    std::vector<int> tab((size_t)8 * cap, -1), used(8, 0);
                    for (int gg = 0; gg < L.G; ++gg) {
                        int left = S, spn = 0;
                        while (left > 0) {
                            int x = 0;
                            for (int k = 1; k < 8; ++k) if (cap - used[k] > cap - used[x]) x = k;
                            const int take = left < cap - used[x] ? left : cap - used[x];
                            for (int q = 0; q < take; ++q) tab[(size_t)(used[x] + q) * 8 + x] = (gg << 8) | (spn + q);
                            used[x] += take; spn += take; left -= take;
                        }
                    }
                    if (m->wgmap) { (void)hipFree(m->wgmap); m->wgmap = nullptr; }
                    TSC_HIP(hipMalloc((void **)&m->wgmap, tab.size() * sizeof(int)));
                    TSC_HIP(hipMemcpyAsync(m->wgmap, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice, m->stream));
                    TSC_HIP(hipStreamSynchronize(m->stream));
                    m->wgmap_S = S; m->wgmap_n = (int)tab.size();
                }
                wgm = m->wgmap; nwg = (unsigned)m->wgmap_n;
            }
#define TSC_WS(KS2) hipLaunchKernelGGL(policy_fwd_ws_kernel<KS2>, dim3(nwg), dim3(512), m->lds_ws, m->stream, m->params, \
                                       L, m->n_act, obs, done, m->state_fw, (int)advance, E, S, pi, v, action, (unsigned long long)seed,     \
                                       (unsigned long long)step, (int)tslot, (long long)m->T * E, m->X1, m->Z, m->Hh, m->Cc, m->Hp, m->dbg, m->Wg, m->krange, m->dbg_tid, wgm)
            if (L.H == 224) TSC_WS(144); else if (L.H == 160) TSC_WS(112); else if (L.H == 192) TSC_WS(128); else TSC_WS(96);
#undef TSC_WS
            ps.stop();
            TSC_HIP(hipGetLastError());
            return 0;
        }
        hipLaunchKernelGGL(policy_fwd_fused_kernel, dim3(8 * per_xcd * n_tiles), dim3(256), m->lds_fused, m->stream, m->params,
                           L, m->n_act, obs, done, m->state_fw, (int)advance, E, n_tiles, pi, v, action,
                           (unsigned long long)seed, (unsigned long long)step, m->dbg, (int)tslot, (long long)m->T * E,
                           m->X1, m->Z, m->Hh, m->Cc, m->Hp, m->Wg);
        ps.stop();
        TSC_HIP(hipGetLastError());
        return 0;
    }
    if (L.fc && (L.H == 160 || L.H == 128) && L.SMAX <= 64 && L.AMAX <= kOut && m->fc_mfma) {   // FcACPolicy (IA2C: large_grid / Monaco) on the matrix cores
        const size_t lds = sizeof(float) * ((size_t)64 * kFmLd + (size_t)2 * 32 * (L.H + 1) + (size_t)8 * 32 * kFmLd + (size_t)2 * 32 * kFcLdo);
        static bool attr_set2 = false;
        if (!attr_set2) {
            TSC_HIP(hipFuncSetAttribute((const void *)policy_fwd_fc_mfma_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            TSC_HIP(hipFuncSetAttribute((const void *)policy_fwd_fc_mfma_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_set2 = true;
        }
        tsc::ProfScope ps(tsc::KID_FUSED_FWD, m->stream);
#define TSC_FCM(NCT) hipLaunchKernelGGL(policy_fwd_fc_mfma_kernel<NCT>, dim3((unsigned)((E + 31) / 32), (unsigned)L.A), dim3(512), lds, m->stream, m->params, L, \
                                        m->n_act, obs, E, pi, v, action, (unsigned long long)seed, (unsigned long long)step, (int)tslot, (long long)m->T * E, m->X1, m->Hh)
        if (L.H == 128) TSC_FCM(4); else TSC_FCM(5);
#undef TSC_FCM
        ps.stop();
        TSC_HIP(hipGetLastError());
        return 0;
    }
    if (L.fc && L.H % 16 == 0 && L.SMAX <= 64 && L.AMAX <= kOut) {     // FcACPolicy: one launch (policy_fwd_fc_kernel)
        const size_t lds = sizeof(float) * ((size_t)64 * kFcLdo + (size_t)2 * 64 * (L.H + 1) + (size_t)2 * 64 * kFcLdo);
        static bool attr_set = false;
        if (!attr_set) {
            TSC_HIP(hipFuncSetAttribute((const void *)policy_fwd_fc_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_set = true;
        }
        if (lds <= 160 * 1024) {
            tsc::ProfScope ps(tsc::KID_FUSED_FWD, m->stream);
            hipLaunchKernelGGL(policy_fwd_fc_kernel, dim3((unsigned)((E + 63) / 64), (unsigned)L.A), dim3(512), lds, m->stream, m->params, L,
                               m->n_act, obs, E, pi, v, action, (unsigned long long)seed, (unsigned long long)step);
            ps.stop();
            TSC_HIP(hipGetLastError());
            return 0;
        }
    }
    // unfused path (shapes the fused kernels do not cover): the training kernels with T = 1
    if (dense_forward(m, obs, E, m->X1, L.fc ? m->Hh : m->Z)) return tsc::fail("tsc_model_forward: gemm launch failed");
    if (L.fc) {                                   // stateless: FcACPolicy.forward (agents/policies.py:237-240)
        tsc::ProfScope psh(tsc::KID_HEAD_FWD, m->stream);
        hipLaunchKernelGGL(head_fwd_kernel, dim3((unsigned)((E + 63) / 64), L.A), dim3(64), 0, m->stream, m->params, L,
                           m->n_act, m->Hh, E, pi, v);
        psh.stop();
        TSC_HIP(hipGetLastError());
        return action ? tsc_model_sample(m, pi, action, seed, step) : 0;
    }
    tsc::ProfScope ps1(tsc::KID_LSTM_FWD, m->stream);
    hipLaunchKernelGGL(lstm_fwd_kernel<false>, dim3(L.G, (E + 63) / 64), dim3(256), m->lds_fwd, m->stream, m->params, L, m->Z,
                       m->state_fw, advance ? m->state_fw : (float *)nullptr, m->Hh, m->Cc, m->Hp, done, 1, E, 0);
    ps1.stop();
    tsc::ProfScope ps3(tsc::KID_HEAD_FWD, m->stream);
    hipLaunchKernelGGL(head_fwd_kernel, dim3((unsigned)((E + 63) / 64), L.A), dim3(64), 0, m->stream, m->params, L,
                       m->n_act, m->Hh, E, pi, v);
    ps3.stop();
    TSC_HIP(hipGetLastError());
    return action ? tsc_model_sample(m, pi, action, seed, step) : 0;
}

int tsc_model_forward(tsc_model *m, const float *obs, const uint8_t *done, float *pi, float *v, int32_t advance) {
    return model_forward(m, obs, done, pi, v, advance, nullptr, 0, 0, -1);
}

int tsc_model_forward_sample(tsc_model *m, const float *obs, const uint8_t *done, float *pi, float *v, int32_t *action,
                             uint64_t seed, uint64_t step, int32_t t_slot) {
    if (!action) return tsc::fail("tsc_model_forward_sample: bad arguments");
    if (m && t_slot >= 0 && t_slot < m->T)      // a forward that reads its observation from slot t_slot is a zero-copy rollout
        m->inplace = obs == m->r_obs + (size_t)t_slot * m->E * m->lay.A * m->lay.SMAX;
    return model_forward(m, obs, done, pi, v, 1, action, seed, step, t_slot);
}

int tsc_model_sample(tsc_model *m, const float *pi, int32_t *action, uint64_t seed, uint64_t step) {
    if (!m || !pi || !action) return tsc::fail("tsc_model_sample: bad arguments");
    const int tot = m->E * m->lay.A;
    tsc::ProfScope ps4(tsc::KID_SAMPLE, m->stream);
    hipLaunchKernelGGL(sample_kernel, dim3((tot + 255) / 256), dim3(256), 0, m->stream, pi, m->n_act, m->E, m->lay.A,
                       m->lay.AMAX, (unsigned long long)seed, (unsigned long long)step, action);
    ps4.stop();
    TSC_HIP(hipGetLastError());
    return 0;
}

int tsc_model_add_transition(tsc_model *m, int32_t t, const float *obs, const uint8_t *done_pre, const int32_t *action,
                             const double *reward, const float *value, const uint8_t *done_post) {
    if (!m || t < 0 || t >= m->T) return tsc::fail("tsc_model_add_transition: slot %d outside [0,%d)", t, m->T);
    const Layout &L = m->lay;
    const long long E = m->E, A = L.A, no = E * A * L.SMAX;
    if (obs != m->r_obs + (size_t)t * no) m->inplace = 0;     // a copied transition: slot T is not part of this rollout
    tsc::ProfScope ps5(tsc::KID_ADD_TRANS, m->stream);
    hipLaunchKernelGGL(add_transition_kernel, dim3((unsigned)((no + 255) / 256)), dim3(256), 0, m->stream, (int)E, (int)A,
                       L.SMAX, obs, done_pre, action, reward, value, done_post, m->r_obs + t * no,
                       m->r_act + t * E * A, m->r_rew + t * E * A, m->r_val + t * E * A, m->r_done + t * E,
                       m->r_done + (t + 1) * E);
    ps5.stop();
    TSC_HIP(hipGetLastError());
    return 0;
}

static int launch_head_bwd(tsc_model *m, long long N, double beta) {
    const Layout &L = m->lay;
    hipStream_t st = m->stream;
    int S = (int)((8 * 256 + L.A - 1) / L.A);        // ~8 workgroups per CU
    long long rps = (N + S - 1) / S;
    rps = (rps + 15) / 16 * 16;
    S = (int)((N + rps - 1) / rps);
    if ((size_t)((long long)S * L.A * kHbPer) > m->ws_floats) return tsc::fail("head_bwd: workspace too small");
    {
        tsc::ProfScope ps7(tsc::KID_HEAD_BWD, m->stream);
        hipLaunchKernelGGL(head_bwd2_kernel, dim3((unsigned)S, (unsigned)L.A), dim3(256), 0, st, m->params, L, m->n_act, m->Hh, m->r_act,
                           m->Rs, m->Advs, N, rps, (float)m->vcoef, (float)beta, m->dHh, m->ws, m->stats);
    }
    {
        tsc::ProfScope ps(tsc::KID_DWO_GEMM, m->stream);
        const int tot = L.A * 2 * (kL * kOut + kOut);
        hipLaunchKernelGGL(head_bwd2_reduce_kernel, dim3((tot + 255) / 256), dim3(256), 0, st, m->ws, L.A, S, m->grads, L);
    }
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

int tsc_model_rollout_slot(tsc_model *m, int32_t t, void *ptrs[6]) {
    if (!m || !ptrs || t < 0 || t > m->T) return tsc::fail("tsc_model_rollout_slot: slot %d outside [0,%d]", t, m ? m->T : 0);
    const Layout &L = m->lay;
    const long long E = m->E, A = L.A, no = E * A * L.SMAX;
    const bool in = t < m->T;
    ptrs[0] = m->r_obs + t * no;                            // obs        f32 [E,A,SMAX]   (slot T: only the obs exists)
    ptrs[1] = in ? (void *)(m->r_act + t * E * A) : nullptr;   // action     i32 [E,A]
    ptrs[2] = in ? (void *)(m->r_val + t * E * A) : nullptr;   // value      f32 [E,A]
    ptrs[3] = in ? (void *)(m->r_rew + t * E * A) : nullptr;   // reward     f64 [E,A]  (raw)
    ptrs[4] = m->r_done + t * E;                            // done before the step  u8 [E]
    ptrs[5] = in ? (void *)(m->r_done + (t + 1) * E) : nullptr;  // done after the step   u8 [E]
    return 0;
}

int tsc_model_compute_grads(tsc_model *m, const float *R_boot, double beta) {
    if (!m || !R_boot) return tsc::fail("tsc_model_compute_grads: bad arguments");
    const Layout &L = m->lay;
    const long long E = m->E, T = m->T, N = E * T, A = L.A, G = L.G;
    const int AS = L.A * L.SMAX;
    hipStream_t st = m->stream;
    TSC_HIP(hipMemsetAsync(m->stats, 0, sizeof(double) * A * 4, st));
    tsc::ProfScope ps6(tsc::KID_RETURNS, m->stream);
    hipLaunchKernelGGL(returns_kernel, dim3((unsigned)((E * A + 255) / 256)), dim3(256), 0, st, m->r_rew, m->r_val, m->r_done,
                       R_boot, (int)T, (int)E, (int)A, m->gamma, m->rnorm, m->rclip, m->Rs, m->Advs);
    ps6.stop();
    float *g = m->grads;
    if (L.fc) {
        // FcACPolicy (agents/policies.py:214-256): Hh = relu(X1 Wfc + bfc); dZ = dH * (Hh > 0) comes out of head_bwd
        // (skipped when the rollout forward cached X1 / Hh of all n_step slots under these very parameters)
        if (m->cached_next != (int)T && dense_forward(m, m->r_obs, N, m->X1, m->Hh)) return tsc::fail("gemm launch failed");
        if (launch_head_bwd(m, N, beta)) return tsc::fail("head_bwd failed");       // + dWo, dbo
        tsc::ProfScope ps9(tsc::KID_TRANSPOSE, m->stream);
        hipLaunchKernelGGL(transpose_wx_kernel, dim3((unsigned)((G * L.H * L.NZ + 255) / 256)), dim3(256), 0, st, m->params, L, m->WxT);
        ps9.stop();
        TSC_HIP(hipGetLastError());
        int S = 256 / (int)G;                       // ~ one workgroup per CU
        if (S < 1) S = 1;
        const long long perfc = (long long)2 * 65 * L.H + (long long)(L.H + 1) * kL;
        if (m->fused_fc && (size_t)((long long)S * G * perfc) <= m->ws_floats && L.ob1 == L.oW1 + (long long)L.SMAX * L.H &&
            L.obl == L.oWx + (long long)L.H * kL) {
            // both layers' weight gradients in one pass, dX1 never leaves the registers (fc_bwd_kernel)
            long long rps = (N + S - 1) / S;
            rps = (rps + 31) / 32 * 32;                  // whole 32-row chunks
            const size_t lds = sizeof(float) * (2 * 32 * kFbLdz + 2 * 32 * kObLd + 2 * 32 * (L.H + 16));
            {
                tsc::ProfScope ps(tsc::KID_DX1_GEMM, m->stream);
#define TSC_FCB(NCU) hipLaunchKernelGGL(fc_bwd_kernel<NCU>, dim3((unsigned)(S * G)), dim3(512), lds, st, m->dHh, m->X1, m->WxT, m->r_obs, N, (int)G, S, rps, (int)A, L.SMAX, m->ws, m->ftmask)
                if (L.H == 160) TSC_FCB(10); else TSC_FCB(8);
#undef TSC_FCB
            }
            {
                tsc::ProfScope ps(tsc::KID_DW1_GEMM, m->stream);
                const long long tot = ((long long)(L.SMAX + 1) * L.H + (long long)(L.H + 1) * kL) * G;
                hipLaunchKernelGGL(fc_bwd_reduce_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, m->ws, (int)G, S, L.H, L.SMAX,
                                   m->rowrange, g, L.stride, L.oW1, L.ob1, L.oWx);
            }
            TSC_HIP(hipGetLastError());
            m->cached_next = 0;
            return 0;
        }
        if (gemm(m, tsc::KID_DWX_GEMM, true, tsc::EPI_NONE, (int)G, L.H, kL, (int)N, m->X1, N * L.H, L.H, 1, m->dHh, N * kL, kL, g + L.oWx,
                 L.stride, kL, nullptr, 0, nullptr, 0, 0, nullptr, 0, g + L.obl, L.stride)) return tsc::fail("gemm failed");
        if (gemm(m, tsc::KID_DX1_GEMM, false, tsc::EPI_MASK_POS, (int)G, (int)N, L.H, kL, m->dHh, N * kL, kL, 1, m->WxT, (long long)L.H * kL, L.H,
                 m->X1, N * L.H, L.H, nullptr, 0, m->X1, N * L.H, L.H, nullptr, 0, nullptr, 0)) return tsc::fail("gemm failed");
    } else {
    // forward with stored activations, from the backward state (agents/policies.py:144-152).  The rollout
    // ran with these very parameters from this very state, so when the fused forward cached all n_step
    // slots (X1, gates, c, h, masked h_prev) the graph does not have to be evaluated a second time.
    if (m->cached_next != (int)T) {
        if (dense_forward(m, m->r_obs, N, m->X1, m->Z)) return tsc::fail("gemm launch failed");
        tsc::ProfScope ps2(tsc::KID_LSTM_FWD, m->stream);
        hipLaunchKernelGGL(lstm_fwd_kernel<true>, dim3((unsigned)G, (unsigned)((E + 63) / 64)), dim3(256), m->lds_fwd, st, m->params, L,
                           m->Z, m->state_bw, (float *)nullptr, m->Hh, m->Cc, m->Hp, m->r_done, (int)T, (int)E, 1);
        ps2.stop();
    }
    if (launch_head_bwd(m, N, beta)) return tsc::fail("head_bwd failed");           // + dWo, dbo
    tsc::ProfScope ps8(tsc::KID_LSTM_BWD, m->stream);
    hipLaunchKernelGGL(lstm_bwd2_kernel, dim3((unsigned)G, (unsigned)((E + 31) / 32)), dim3(256), sizeof(float) * 32 * (kDz2Ld + kDh3Ld), st,
                       m->params, L, m->Z, m->Cc, m->state_bw, m->dHh, m->r_done, (int)T, (int)E);
    ps8.stop();
    tsc::ProfScope ps9(tsc::KID_TRANSPOSE, m->stream);
    hipLaunchKernelGGL(transpose_wx_kernel, dim3((unsigned)((G * L.H * kG4 + 255) / 256)), dim3(256), 0, st, m->params, L, m->WxT);
    ps9.stop();
    TSC_HIP(hipGetLastError());
    // dWh = Hp^T dZ (+ dbl) ; dWx = X1^T dZ        (dWo = Hh^T dL and dbo came out of head_bwd)
    const int NT = (L.H + kL) / 32;
    int S = 256 / (int)G;                       // ~ one workgroup per CU
    if (S < 1) S = 1;
    const long long per = (long long)(NT * 32 + 1) * kG4;
    if (m->fused_dw && (size_t)((long long)S * G * per) <= m->ws_floats && L.oWh == L.oWx + (long long)L.H * kG4 &&
        L.obl == L.oWh + (long long)kL * kG4) {
        long long rps = (N + S - 1) / S;
        rps += rps & 1;                           // a k-step is two rows
        {
            tsc::ProfScope ps(tsc::KID_DWX_GEMM, m->stream);
#define TSC_DWXH(NT_) hipLaunchKernelGGL(dwxh_kernel<NT_>, dim3((unsigned)(S * G)), dim3(512), 0, st, m->X1, m->Hp, m->Z, N, (int)G, S, rps, m->ws)
            if (NT == 9) TSC_DWXH(9); else if (NT == 7) TSC_DWXH(7); else if (NT == 8) TSC_DWXH(8); else TSC_DWXH(6);
#undef TSC_DWXH
        }
        {
            tsc::ProfScope ps(tsc::KID_DWH_GEMM, m->stream);
            hipLaunchKernelGGL(dwxh_reduce_kernel, dim3((unsigned)((per * G + 255) / 256)), dim3(256), 0, st, m->ws, (int)G, S, per, g,
                               L.stride, L.oWx);
        }
        TSC_HIP(hipGetLastError());
    } else {
    if (gemm(m, tsc::KID_DWH_GEMM, true, tsc::EPI_NONE, (int)G, kL, kG4, (int)N, m->Hp, N * kL, kL, 1, m->Z, N * kG4, kG4, g + L.oWh, L.stride,
             kG4, nullptr, 0, nullptr, 0, 0, nullptr, 0, g + L.obl, L.stride)) return tsc::fail("gemm failed");
    if (gemm(m, tsc::KID_DWX_GEMM, true, tsc::EPI_NONE, (int)G, L.H, kG4, (int)N, m->X1, N * L.H, L.H, 1, m->Z, N * kG4, kG4, g + L.oWx, L.stride,
             kG4, nullptr, 0, nullptr, 0, 0, nullptr, 0, nullptr, 0)) return tsc::fail("gemm failed");
    }
    if (m->fused_dx && L.ob1 == L.oW1 + (long long)L.SMAX * L.H && (size_t)((long long)S * G * 2 * 65 * L.H) <= m->ws_floats) {
        // dX1 stays in registers: dW1 | db1 come out of the same pass (dx1w1_kernel2)
        long long rps = (N + S - 1) / S;
        rps = (rps + 31) / 32 * 32;                  // whole 32-row chunks
        const size_t lds = sizeof(float) * (2 * 32 * kD1Ld + 2 * 32 * kObLd);
        {
            tsc::ProfScope ps(tsc::KID_DX1_GEMM, m->stream);
#define TSC_DX2(NCU) hipLaunchKernelGGL(dx1w1_kernel2<NCU>, dim3((unsigned)(S * G)), dim3(512), lds, st, m->Z, m->X1, m->WxT, m->r_obs, N, (int)G, S, rps, (int)A, L.SMAX, m->ws, m->ftmask)
            if (L.H == 224) TSC_DX2(14); else if (L.H == 160) TSC_DX2(10); else if (L.H == 192) TSC_DX2(12); else TSC_DX2(8);
#undef TSC_DX2
        }
        {
            tsc::ProfScope ps(tsc::KID_DW1_GEMM, m->stream);
            const long long tot = (long long)(L.SMAX + 1) * L.H * G;
            hipLaunchKernelGGL(dx1w1_reduce2_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, m->ws, (int)G, S, L.H, L.SMAX,
                               m->rowrange, g, L.stride, L.oW1, L.ob1);
        }
        TSC_HIP(hipGetLastError());
        m->cached_next = 0;
        return 0;
    }
    // dX1 = (dZ Wx^T) * relu'(X1), in place over X1
    if (gemm(m, tsc::KID_DX1_GEMM, false, tsc::EPI_MASK_POS, (int)G, (int)N, L.H, kG4, m->Z, N * kG4, kG4, 1, m->WxT, (long long)L.H * kG4, L.H,
             m->X1, N * L.H, L.H, nullptr, 0, m->X1, N * L.H, L.H, nullptr, 0, nullptr, 0)) return tsc::fail("gemm failed");
    }
    // dW1 = obs^T dX1 masked to the block-diagonal structure (+ db1)
    if (gemm(m, tsc::KID_DW1_GEMM, true, tsc::EPI_ROWRANGE, (int)G, L.SMAX, L.H, (int)N, m->r_obs, L.SMAX, AS, 2, m->X1, N * L.H, L.H, g + L.oW1,
             L.stride, L.H, nullptr, 0, nullptr, 0, 0, m->rowrange, L.SMAX, g + L.ob1, L.stride)) return tsc::fail("gemm failed");
    m->cached_next = 0;                               // the update consumes the cache
    TSC_HIP(hipGetLastError());
    return 0;
}

int tsc_model_grad_buffer(tsc_model *m, float **grad, int64_t *count) {
    if (!m || !grad || !count) return tsc::fail("tsc_model_grad_buffer: bad arguments");
    *grad = m->grads; *count = m->nparam;
    return 0;
}

int tsc_model_apply_grads(tsc_model *m, double lr, double grad_scale, double *stats_host) {
    if (!m) return tsc::fail("null handle");
    const Layout &L = m->lay;
    const long long per_agent = 2 * L.stride;
    hipStream_t st = m->stream;
    tsc::ProfScope ps10(tsc::KID_GRADNORM, m->stream);
    hipLaunchKernelGGL(grad_norm_kernel, dim3(L.A, kNormParts), dim3(256), 0, st, m->grads, per_agent, grad_scale, m->norm_part);
    hipLaunchKernelGGL(grad_norm_fold_kernel, dim3((L.A + 63) / 64), dim3(64), 0, st, m->norm_part, L.A, m->norm2);
    ps10.stop();
    tsc::ProfScope ps11(tsc::KID_RMSPROP, m->stream);
    hipLaunchKernelGGL(rmsprop_kernel, dim3((unsigned)((m->nparam + 255) / 256)), dim3(256), 0, st, m->params, m->ms, m->grads,
                       per_agent, m->nparam, m->norm2, (float)grad_scale, (float)m->max_norm, (float)lr, (float)m->alpha,
                       (float)m->eps);
    ps11.stop();
    m->wg_dirty = 1;
    TSC_HIP(hipGetLastError());
    // states_bw <- states_fw (policies.py:153); buffer.reset(dones[-1]) (utils.py:227)
    TSC_HIP(hipMemcpyAsync(m->state_bw, m->state_fw, sizeof(float) * (size_t)L.G * m->E * 2 * kL, hipMemcpyDeviceToDevice, st));
    TSC_HIP(hipMemcpyAsync(m->r_done, m->r_done + (size_t)m->T * m->E, m->E, hipMemcpyDeviceToDevice, st));
    // zero-copy rollouts only: the observation the env wrote into slot T is the first observation of the next rollout
    // (on the add_transition path slot T is never written and slot 0 belongs to the caller's next add_transition(t = 0))
    if (m->inplace)
        TSC_HIP(hipMemcpyAsync(m->r_obs, m->r_obs + (size_t)m->T * m->E * L.A * L.SMAX, sizeof(float) * (size_t)m->E * L.A * L.SMAX,
                               hipMemcpyDeviceToDevice, st));
    if (stats_host) {
        std::vector<double> s(L.A * 4), n2(L.A);
        TSC_HIP(hipStreamSynchronize(st));
        TSC_HIP(hipMemcpy(s.data(), m->stats, sizeof(double) * L.A * 4, hipMemcpyDeviceToHost));
        TSC_HIP(hipMemcpy(n2.data(), m->norm2, sizeof(double) * L.A, hipMemcpyDeviceToHost));
        for (int a = 0; a < L.A; ++a) {
            for (int k = 0; k < 3; ++k) stats_host[a * 4 + k] = s[a * 4 + k];
            stats_host[a * 4 + 3] = sqrt(n2[a]);
        }
    }
    return 0;
}

int tsc_model_debug_clock(tsc_model *m, int32_t enable, int64_t *stamps_host, int32_t count) {
    if (!m) return tsc::fail("null handle");
    TSC_HIP(hipStreamSynchronize(m->stream));
    const size_t n = 64 + 2 * 8 * (size_t)((m->lay.G + 7) / 8) * ((m->E + 63) / 64);
    if (enable && !m->dbg) {
        TSC_HIP(hipMalloc((void **)&m->dbg, n * sizeof(long long)));
        TSC_HIP(hipMemset(m->dbg, 0, n * sizeof(long long)));
        m->allocs.push_back(m->dbg);
    }
    if (stamps_host && m->dbg)
        TSC_HIP(hipMemcpy(stamps_host, m->dbg, sizeof(long long) * ((size_t)count < n ? (size_t)count : n), hipMemcpyDeviceToHost));
    return 0;
}

int tsc_model_get_returns(tsc_model *m, float *Rs, float *Advs) {
    if (!m || !Rs || !Advs) return tsc::fail("tsc_model_get_returns: bad arguments");
    const size_t n = (size_t)m->T * m->E * m->lay.A;
    TSC_HIP(hipStreamSynchronize(m->stream));
    TSC_HIP(hipMemcpy(Rs, m->Rs, sizeof(float) * n, hipMemcpyDeviceToHost));
    TSC_HIP(hipMemcpy(Advs, m->Advs, sizeof(float) * n, hipMemcpyDeviceToHost));
    return 0;
}

int tsc_model_debug_read(tsc_model *m, int32_t what, int32_t g, int64_t row0, int64_t nrows, float *out_host) {
    if (!m || !out_host || g < 0 || g >= m->lay.G || row0 < 0 || nrows < 0) return tsc::fail("tsc_model_debug_read: bad arguments");
    const long long N = (long long)m->T * m->E;
    if (row0 + nrows > N) return tsc::fail("tsc_model_debug_read: rows [%lld, %lld) outside [0, %lld)", (long long)row0, (long long)(row0 + nrows), N);
    const float *base = nullptr;
    long long w = 0;
    switch (what) {
        case 0: base = m->X1; w = m->lay.H; break;
        case 1: base = m->Z; w = kG4; break;
        case 2: base = m->Hh; w = kL; break;
        case 3: base = m->Cc; w = kL; break;
        case 4: base = m->Hp; w = kL; break;
        case 5: base = m->dHh; w = kL; break;
        default: return tsc::fail("tsc_model_debug_read: unknown buffer %d", what);
    }
    if (m->lay.fc && (what == 1 || what == 3 || what == 4)) return tsc::fail("tsc_model_debug_read: buffer %d is LSTM-only", what);
    TSC_HIP(hipStreamSynchronize(m->stream));
    TSC_HIP(hipMemcpy(out_host, base + ((long long)g * N + row0) * w, sizeof(float) * (size_t)(nrows * w), hipMemcpyDeviceToHost));
    return 0;
}

int tsc_gemm_grouped_f32(int32_t form, int32_t epi, int32_t groups, int32_t M, int32_t N, int32_t K, const float *A,
                         int64_t sA, int32_t lda, const float *B, int64_t sB, int32_t ldb, float *C, int64_t sC,
                         int32_t ldc, const float *bias, const float *aux, const int16_t *rowrange, float *colsum,
                         float *splitk_ws, int64_t ws_floats, float *splitk_wsc, int64_t wsc_floats, void *hip_stream) {
    GemmArgs a;
    a.A = A; a.B = B; a.C = C; a.bias = bias; a.aux = aux; a.rr = rowrange; a.colsum = colsum;
    a.sA = sA; a.sB = sB; a.sC = sC; a.sBias = N; a.sAux = (long long)M * ldc; a.sRR = M; a.sColsum = N;
    a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ldaux = ldc; a.M = M; a.N = N; a.K = K; a.gdivA = 1;
    tsc::plan_splitk(a, groups, form != 0 ? splitk_ws : nullptr, splitk_wsc, (size_t)ws_floats, (size_t)wsc_floats);
    tsc::launch_gemm_dyn(form != 0, epi, a, groups, (hipStream_t)hip_stream);
    TSC_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
