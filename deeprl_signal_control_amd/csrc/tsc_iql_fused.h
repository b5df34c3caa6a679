// tsc_iql_fused.h -- the DeepQPolicy learner of csrc/tsc_iql.hip as ONE kernel per minibatch step (round 6).
//
// Replaces, for all agents of all env instances at once (agents/policies.py:307-371, agents/models.py:319-326):
//   gather of the sampled transitions, Q(s'), max_a Q(s'), Q(s), the TD target, dQ, and the gradient of every layer
//   (q_fcw | q_fct | q_fc_0 | q) -- 18 grouped GEMM launches + 4 element-wise kernels per minibatch step before, with the
//   160- and 64-wide activations of 20 480 rows x 25 agents round-tripping HBM between every pair of them (~ 400 MB per
//   first-layer launch).  Here the only HBM traffic is the two sampled observation rows of every transition (2 x 144 B)
//   and the per-workgroup partial gradients.
//
// Mapping (one workgroup = 4 wavefronts = one agent's slice of the minibatch, walked in 64-row chunks):
//   * phase A, per wavefront, 16 rows, no barrier: the nets run TRANSPOSED on v_mfma_f32_16x16x4_f32 -- out^T[feature][row] =
//     W^T[feature][k] in^T[k][row] -- so that a layer's accumulator (lane = row, registers = features 16 t + 4 (lane >> 4) + i) is
//     the next layer's B operand as it stands: the contraction order inside a 16-feature tile is permuted to (i, lane >> 4) and
//     the weight operand follows it.  Activations never leave the registers between layers.  W1 is stationary in registers
//     (the observation rows come straight from the replay ring as 16-byte loads), W2 | Wq | biases sit in LDS.
//     Q(s') -> max, Q(s) -> TD error g, dX2 = g Wq[:, a] relu'(X2) element-wise (dQ has one non-zero per row),
//     dX1 = (W2 dX2) relu'(X1) on the matrix cores.
//   * the weight gradients contract over ROWS, i.e. need lane = feature: the wavefronts write X1 | X2 | dX2 | S (then dX1)
//     transposed into LDS ([feature][64 rows]) and, behind a barrier, every wavefront accumulates the 16 x 16 tiles of
//     dW2 | dWq (phase B) and dW1 (phase D) it owns over the chunk's 64 rows; the accumulators live in registers over the whole
//     slice and are written once, as the split's partial gradient; iql_fused_reduce_kernel folds the splits in fixed order
//     (deterministic), applies W1's block-diagonal mask and sums the loss.
// Arithmetic: fp32 MFMA (exact f32 products, f32 accumulate) -- the same numbers as the grouped-GEMM path up to summation order.
#pragma once
#include <type_traits>

#ifndef TSC_IQL_SCHED
#define TSC_IQL_SCHED 1        // bit 0: pin the operand fetches of the second layer ahead of its MFMAs; 1: phase B; 2: phase D
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kFH2 = 64;          // q_fc_0 width the fused kernels are built for (num_h = 64, config/*.ini)
constexpr int kFLd = 68;          // LDS row stride of the [feature][64 rows] / [H1][64 columns] images (floats)
constexpr int kFLq = 20;          // LDS row stride of Wq [64][16 (8 used)]
constexpr int kFSF = 48;          // observation features staged (three 16-byte-per-lane groups of 16)

struct QFusedArgs {
    const float *params;
    const int *n_act, *n_wave, *n_wait, *idx;
    const float *r_obs, *r_next, *r_rew;
    const int *r_act;
    const uint8_t *r_done;
    int E, A, B, SMAX, size;
    long long cap, R;             // ring capacity; rows per agent = E * B
    float gamma;
    int S, cps;                   // row splits per agent; 64-row chunks per split
    float *ws, *wsl;              // partial gradients [S][A][stride]; partial losses [S][A]
    long long stride, oW1, ob1, oW2, ob2, oWq, obq;
    long long *dbg;               // tsc_iql_debug_clock: [64] phase stamps of workgroup 0 (16 per wavefront, chunk 2) | [2 x workgroups] start / end (100 MHz)
};

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// LDS image of one agent's second and third layer + all biases
template <int NM1>
struct QFusedLds {
    static constexpr int H1 = 16 * NM1;
    static constexpr int oW2 = 0;                         // [H1][kFLd]
    static constexpr int oWq = oW2 + H1 * kFLd;           // [64][kFLq]
    static constexpr int oB = oWq + kFH2 * kFLq;          // b1[H1] | b2[64] | bq[16 (8 used)]
    static constexpr int fwd_floats = oB + H1 + kFH2 + 16;
    static constexpr int oX1 = fwd_floats;                // [H1][kFLd]   X1, later dX1
    static constexpr int oX2 = oX1 + H1 * kFLd;           // [64][kFLd]
    static constexpr int oD2 = oX2 + kFH2 * kFLd;         // [64][kFLd]
    static constexpr int oS = oD2 + kFH2 * kFLd;          // [kFSF][kFLd]
    static constexpr int oG = oS + kFSF * kFLd;           // g[64] | action[64]
    static constexpr int grad_floats = oG + 128;
};

template <int NM1>
__device__ __forceinline__ void q_stage_weights(const float *__restrict__ P, const QFusedArgs &p, float *sm, int tid, int nthr) {
    using LD = QFusedLds<NM1>;
    constexpr int H1 = LD::H1;
    for (int q = tid; q < H1 * (kFH2 / 4); q += nthr) {
        const int r = q / (kFH2 / 4), c4 = q % (kFH2 / 4);
        *reinterpret_cast<float4 *>(sm + LD::oW2 + r * kFLd + 4 * c4) = *reinterpret_cast<const float4 *>(P + p.oW2 + (long long)r * kFH2 + 4 * c4);
    }
    for (int q = tid; q < kFH2 * 16; q += nthr) {
        const int r = q >> 4, c = q & 15;
        sm[LD::oWq + r * kFLq + c] = c < 8 ? P[p.oWq + r * 8 + c] : 0.f;
    }
    for (int q = tid; q < H1 + kFH2 + 16; q += nthr) {
        float v = 0.f;
        if (q < H1) v = P[p.ob1 + q];
        else if (q < H1 + kFH2) v = P[p.ob2 + q - H1];
        else if (q < H1 + kFH2 + 8) v = P[p.obq + q - H1 - kFH2];
        sm[LD::oB + q] = v;
    }
}

// per-agent contraction ranges of the first layer, in 16-feature groups: columns [0, 16 NMW) read the wave features
// [0, n_wave), columns [16 NMW, H1) the wait features [n_wave, n_wave + n_wait)  (W1 is block-diagonal, agents/policies.py:355-360)
struct QRanges { int qw1, qt0, qt1; };
__device__ __forceinline__ QRanges q_ranges(int nw, int nt) {
    QRanges r;
    r.qw1 = (nw + 15) >> 4;
    r.qt0 = nw >> 4;
    r.qt1 = nt > 0 ? (nw + nt + 15) >> 4 : r.qt0;
    return r;
}

// one agent's nets on 16 rows of this wavefront: lane = (row n = lane & 15, kq = lane >> 4)
//   s[q] = obs[row][16 q + 4 kq .. + 3];  w1[t][q][c] = W1[16 q + 4 kq + c][16 t + m]  (m = lane & 15 as the A operand's row)
//   X1[t][i] = relu(.)[feature 16 t + 4 kq + i][row n], X2 likewise, q[i] = Q[action 4 kq + i][row n] (kq < 2)
// stationary first-layer operands of one lane: the wave part's columns x the 16-feature groups they can read (three without a wait
// part: up to 48 wave features; two with one: the host admits n_wave <= 32 there), the wait part's columns x the (at most two)
// groups [qt0, qt0 + 2) its features [n_wave, n_wave + n_wait <= 16) fall into
template <int NM1, int NMW>
struct QW1 {
    static constexpr int QW = NM1 > NMW ? 2 : 3, NT = NM1 > NMW ? NM1 - NMW : 1;
    float w[NMW][QW][4];
    float t[NT][2][4];
};

template <int NB, int NM1, int NMW>
__device__ __forceinline__ void q_nets(const QW1<NM1, NMW> &w1, const float4 (&s)[NB][3], const float *sm, const QRanges &rg, int m, int kq,
                                       f32x4 (&X1)[NB][NM1], f32x4 (&X2)[NB][4], f32x4 (&q)[NB], long long *fst = nullptr) {
    // NB independent row sets (the gradient kernel's s' and s) go through every layer side by side: one weight operand feeds NB
    // MFMAs, and the relu / bias work of one set sits under the MFMAs of the other instead of at a serialising layer boundary
    using LD = QFusedLds<NM1>;
    using W = QW1<NM1, NMW>;
    constexpr int H1 = LD::H1;
    const float *Bs = sm + LD::oB;
#pragma unroll
    for (int t = 0; t < NM1; ++t) {
        const float4 b = *reinterpret_cast<const float4 *>(Bs + 16 * t + 4 * kq);
#pragma unroll
        for (int b_ = 0; b_ < NB; ++b_) X1[b_][t] = f32x4{b.x, b.y, b.z, b.w};
    }
    // the observation groups as plain values: with the loads visible the compiler folds the wait part's wavefront-uniform choice of a
    // group (below) into an indexed load, which sends the caller's row buffers to scratch
    float so[NB][3][4];
#pragma unroll
    for (int b_ = 0; b_ < NB; ++b_)
#pragma unroll
        for (int qp = 0; qp < 3; ++qp) {
            so[b_][qp][0] = s[b_][qp].x; so[b_][qp][1] = s[b_][qp].y; so[b_][qp][2] = s[b_][qp].z; so[b_][qp][3] = s[b_][qp].w;
#pragma unroll
            for (int c = 0; c < 4; ++c) asm("" : "+v"(so[b_][qp][c]));
        }
    // No branch on the agent's group counts: W1 holds structural zeros outside an agent's wave / wait rows and the observation is
    // zero-padded, so the groups an agent does not use contribute exact zeros -- and a wavefront-uniform branch around a block of MFMAs
    // makes the compiler shuttle every accumulator between the AGPR and VGPR files at the block's edges (measured: 49 instead of 32
    // cycles per MFMA over this layer).
#pragma unroll
    for (int qp = 0; qp < W::QW; ++qp)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int t = 0; t < NMW; ++t)
#pragma unroll
                for (int b_ = 0; b_ < NB; ++b_) X1[b_][t] = mfma16(w1.w[t][qp][c], so[b_][qp][c], X1[b_][t]);
    if (NM1 > NMW) {
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
            const int qp = rg.qt0 + qq;                            // wavefront-uniform; qt0 + 1 <= 2 (n_wave <= 32)
            float sv[NB][4];
#pragma unroll
            for (int b_ = 0; b_ < NB; ++b_)
#pragma unroll
                for (int c = 0; c < 4; ++c) sv[b_][c] = qp == 0 ? so[b_][0][c] : qp == 1 ? so[b_][1][c] : so[b_][2][c];
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int t = NMW; t < NM1; ++t)
#pragma unroll
                    for (int b_ = 0; b_ < NB; ++b_) X1[b_][t] = mfma16(w1.t[t - NMW][qq][c], sv[b_][c], X1[b_][t]);
        }
    }
#ifdef TSC_IQL_FINE
    if (fst) fst[11] = clock64();
#endif
#pragma unroll
    for (int b_ = 0; b_ < NB; ++b_)
#pragma unroll
        for (int t = 0; t < NM1; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) X1[b_][t][i] = X1[b_][t][i] > 0.f ? X1[b_][t][i] : 0.f;
#ifdef TSC_IQL_FINE
    if (fst) fst[12] = clock64();
#endif
    // second layer: A = W2[k = 16 kt + 4 kq + i][out 16 t2 + m] from LDS (consecutive lanes, consecutive banks), B = X1[kt][i]
#pragma unroll
    for (int t2 = 0; t2 < 4; ++t2) {
        const float4 b = *reinterpret_cast<const float4 *>(Bs + H1 + 16 * t2 + 4 * kq);
#pragma unroll
        for (int b_ = 0; b_ < NB; ++b_) X2[b_][t2] = f32x4{b.x, b.y, b.z, b.w};
    }
    const float *W2s = sm + LD::oW2 + 4 * kq * kFLd + m;
    // the 16 weight operands of k-tile kt + 1 are requested before the MFMAs of k-tile kt are issued (the compiler, left alone,
    // put every ds_read right in front of its two MFMAs with lgkmcnt(0))
    float wa[2][16];
    auto fetch = [&](int kt, float (&w)[16]) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int t2 = 0; t2 < 4; ++t2) w[4 * i + t2] = W2s[(16 * kt + i) * kFLd + 16 * t2];
    };
    fetch(0, wa[0]);
#pragma unroll
    for (int kt = 0; kt < NM1; ++kt) {
        if (kt + 1 < NM1) fetch(kt + 1, wa[(kt + 1) & 1]);
#if TSC_IQL_SCHED & 1
        __builtin_amdgcn_sched_barrier(0);                         // the requests stay in front of this k-tile's MFMAs
#endif
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int t2 = 0; t2 < 4; ++t2)
#pragma unroll
                for (int b_ = 0; b_ < NB; ++b_) X2[b_][t2] = mfma16(wa[kt & 1][4 * i + t2], X1[b_][kt][i], X2[b_][t2]);
#if TSC_IQL_SCHED & 1
        __builtin_amdgcn_sched_barrier(0);
#endif
    }
#ifdef TSC_IQL_FINE
    if (fst) fst[13] = clock64();
#endif
#pragma unroll
    for (int b_ = 0; b_ < NB; ++b_)
#pragma unroll
        for (int t2 = 0; t2 < 4; ++t2)
#pragma unroll
            for (int i = 0; i < 4; ++i) X2[b_][t2][i] = X2[b_][t2][i] > 0.f ? X2[b_][t2][i] : 0.f;
#ifdef TSC_IQL_FINE
    if (fst) fst[14] = clock64();
#endif
    // Q^T [16 (8 used) actions][16 rows]: two accumulators per set (the 16x16x4 form's dependent latency is 40 cycles)
    const float4 bq = *reinterpret_cast<const float4 *>(Bs + H1 + kFH2 + 4 * kq);
    f32x4 qa[NB], qb[NB];
#pragma unroll
    for (int b_ = 0; b_ < NB; ++b_) { qa[b_] = f32x4{bq.x, bq.y, bq.z, bq.w}; qb[b_] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const float *Wqs = sm + LD::oWq + 4 * kq * kFLq + m;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float wA = Wqs[(32 * h + i) * kFLq], wB = Wqs[(32 * h + 16 + i) * kFLq];
#pragma unroll
            for (int b_ = 0; b_ < NB; ++b_) {
                qa[b_] = mfma16(wA, X2[b_][2 * h][i], qa[b_]);
                qb[b_] = mfma16(wB, X2[b_][2 * h + 1][i], qb[b_]);
            }
        }
#pragma unroll
    for (int b_ = 0; b_ < NB; ++b_) q[b_] = qa[b_] + qb[b_];
}

template <int NM1, int NMW>
__device__ __forceinline__ void q_load_w1(const float *__restrict__ P, const QFusedArgs &p, const QRanges &rg, int m, int kq, QW1<NM1, NMW> &w1) {
    constexpr int H1 = 16 * NM1;
    using W = QW1<NM1, NMW>;
    auto ld = [&](int f, int col) {                      // W1[f][col], zero past the observation's width (the row index is clamped, the value selected)
        const float v = P[p.oW1 + (long long)(f < p.SMAX ? f : p.SMAX - 1) * H1 + col];
        return f < p.SMAX ? v : 0.f;
    };
#pragma unroll
    for (int t = 0; t < NMW; ++t)
#pragma unroll
        for (int qp = 0; qp < W::QW; ++qp)
#pragma unroll
            for (int c = 0; c < 4; ++c) w1.w[t][qp][c] = ld(16 * qp + 4 * kq + c, 16 * t + m);
    if (NM1 > NMW) {
#pragma unroll
        for (int t = NMW; t < NM1; ++t)
#pragma unroll
            for (int qq = 0; qq < 2; ++qq)
#pragma unroll
                for (int c = 0; c < 4; ++c) w1.t[t - NMW][qq][c] = ld(16 * (rg.qt0 + qq) + 4 * kq + c, 16 * t + m);
    }
}

// 16 bytes of an observation row (features 16 q + 4 kq ..): zero past the row's end
__device__ __forceinline__ float4 q_obs4(const float *row, int f0, int SMAX, bool ok) {
    const bool in = ok && f0 < SMAX;                     // the load itself is unconditional (a valid address either way): no branch per group
    const float4 v = *reinterpret_cast<const float4 *>(row + (f0 < SMAX ? f0 : 0));
    return in ? v : make_float4(0.f, 0.f, 0.f, 0.f);
}

// ---- the minibatch gradient ---------------------------------------------------------------------------------------------------
template <int NM1, int NMW>
__global__ void __launch_bounds__(256, 1) iql_fused_grad_kernel(QFusedArgs p) {
    using LD = QFusedLds<NM1>;
    constexpr int H1 = LD::H1;
    extern __shared__ __attribute__((aligned(16))) float q_smem[];
    float *sm = q_smem;
    const int a = blockIdx.x % p.A, sp = blockIdx.x / p.A;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kq = lane >> 4;             // n: row (B operand / accumulator column) or the A operand's row m
    const float *P = p.params + (long long)a * p.stride;
    const int na = p.n_act[a];
    const QRanges rg = q_ranges(p.n_wave[a], p.n_wait[a]);
    q_stage_weights<NM1>(P, p, sm, tid, 256);
    QW1<NM1, NMW> w1;
    q_load_w1<NM1, NMW>(P, p, rg, n, kq, w1);

    // ---- this wavefront's tiles of the weight gradients (all of them live in registers over the whole slice; equal work per wavefront)
    // dW2 [H1][64]: column tile `wave`, all NM1 row tiles;  dWq [64][8]: row tile `wave`
    // dW1 [48][H1]: wave part (columns < 16 NMW): column tiles wave, wave + 4 x the feature tiles [0, qw1);
    //               wait part (columns >= 16 NMW): tile `wave` of the list (feature tile qt0 + wave / 2, column tile NMW + wave % 2)
    static_assert(NMW == 8, "two wave-part column tiles per wavefront");
    f32x4 aW2[NM1], aWq, aW1w[2][3], aW1t;
#pragma unroll
    for (int t = 0; t < NM1; ++t) aW2[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int t = 0; t < 3; ++t) aW1w[j][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    aWq = aW1t = f32x4{0.f, 0.f, 0.f, 0.f};
    float sb2 = 0.f, sbq = 0.f, sb1[2] = {0.f, 0.f}, sb1t = 0.f, loss = 0.f;
    // the wait-part tile of this wavefront: list index = wave (at most 2 feature groups x 2 column tiles, see QW1)
    const bool wt_on = NM1 > NMW && rg.qt0 + (wave >> 1) < rg.qt1;
    const int wt_mf = wt_on ? rg.qt0 + (wave >> 1) : 0, wt_ni = NM1 > NMW ? NMW + (wave & 1) : 0;
    const long long nchunks = (p.R + 63) >> 6;
    const long long c0 = (long long)sp * p.cps;
    long long c1 = c0 + p.cps;
    if (c1 > nchunks) c1 = nchunks;
    const float fR = (float)p.R;

    struct Rows { float4 s0[3], s1[3]; float rew; int act, done; bool ok; };
    // rows are 32-bit here (the host refuses E * B >= 2^31): a 64-bit division per lane and chunk costs more than the chunk's VALU work
    const unsigned uR = (unsigned)p.R, uB = (unsigned)p.B;
    struct Slot { int slot; unsigned e; };
    auto slot_of = [&](long long c) -> Slot {            // ring slot of this lane's row in chunk c (clamped like the gather kernel's)
        unsigned row = ((unsigned)c << 6) + 16 * wave + n;
        if (row >= uR) row = uR - 1;
        Slot o;
        o.e = row / uB;
        const int s_ = p.idx[((long long)o.e * p.A + a) * p.B + (row - o.e * uB)];
        o.slot = s_ < 0 ? 0 : s_ >= p.size ? p.size - 1 : s_;
        return o;
    };
    auto load_rows = [&](long long c, const Slot &sl, Rows &r) {
        const unsigned row = ((unsigned)c << 6) + 16 * wave + n;
        r.ok = row < uR && c < c1;
        const long long tr = ((long long)sl.e * p.cap + sl.slot) * p.A + a;
        const float *o = p.r_obs + tr * p.SMAX, *o1 = p.r_next + tr * p.SMAX;
#pragma unroll
        for (int qp = 0; qp < 3; ++qp) {
            r.s0[qp] = q_obs4(o, 16 * qp + 4 * kq, p.SMAX, r.ok);
            r.s1[qp] = q_obs4(o1, 16 * qp + 4 * kq, p.SMAX, r.ok);
        }
        r.rew = p.r_rew[tr];
        r.act = p.r_act[tr];
        r.done = p.r_done[(long long)sl.e * p.cap + sl.slot];
    };

    const bool stamp_wg = p.dbg && blockIdx.x == 0 && lane == 0;
    if (p.dbg && tid == 0) p.dbg[64 + 2 * blockIdx.x] = wall_clock64();
#ifdef TSC_IQL_STAMPS
#define QSTAMP(k) do { if (stamp_wg && c == c0 + 2) p.dbg[16 * wave + (k)] = clock64(); } while (0)
#else
#define QSTAMP(k) do { } while (0)
#endif
    Rows cur, nxt;
    Slot slot_n = {0, 0};
    if (c0 < c1) {
        load_rows(c0, slot_of(c0), cur);
        slot_n = slot_of(c0 + 1 < c1 ? c0 + 1 : c0);
    }
    __syncthreads();                                      // weights are in LDS

    for (long long c = c0; c < c1; ++c) {
        // next chunk's rows (their slot arrived a chunk ago) and the slot of the chunk after it
        load_rows(c + 1 < c1 ? c + 1 : c, slot_n, nxt);
        slot_n = slot_of(c + 2 < c1 ? c + 2 : c);
        // ================= phase A: 16 rows per wavefront =================
        QSTAMP(0);
        // Q(s') (set 0: only its maximum survives) and Q(s) (set 1: its activations feed the backward pass) side by side
        f32x4 XX1[2][NM1], XX2[2][4], qq[2];
        {
            float4 ss[2][3];
#pragma unroll
            for (int qp = 0; qp < 3; ++qp) { ss[0][qp] = cur.s1[qp]; ss[1][qp] = cur.s0[qp]; }
            q_nets<2, NM1, NMW>(w1, ss, sm, rg, n, kq, XX1, XX2, qq, (stamp_wg && c == c0 + 2) ? p.dbg + 16 * wave : nullptr);
        }
        QSTAMP(1);
        f32x4 (&X1)[NM1] = XX1[1];
        f32x4 (&X2)[4] = XX2[1];
        const f32x4 q = qq[1];
        float q1 = -INFINITY;
#pragma unroll
        for (int i = 0; i < 4; ++i) if (kq < 2 && 4 * kq + i < na) q1 = fmaxf(q1, qq[0][i]);
        q1 = fmaxf(q1, __shfl_xor(q1, 16, 64));
        q1 = fmaxf(q1, __shfl_xor(q1, 32, 64));
        const int act = cur.act;
        float q0 = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) if (kq == (act >> 2) && (act & 3) == i) q0 = q[i];
        q0 += __shfl_xor(q0, 16, 64);
        q0 += __shfl_xor(q0, 32, 64);
        // tq = done ? r : r + gamma q1;  loss = mean((q0 - tq)^2);  g = dLoss / dQ[a]   (agents/policies.py:315-318)
        const float tq = cur.done ? cur.rew : cur.rew + p.gamma * q1;
        const float d = cur.ok ? q0 - tq : 0.f;
        const float g = 2.0f * d / fR;
        if (kq == 0) loss += d * d / fR;
        // transposed images for the weight gradients: [feature][row]
        const int rcol = 16 * wave + n;
        {
            float *x1s = sm + LD::oX1 + 4 * kq * kFLd + rcol, *x2s = sm + LD::oX2 + 4 * kq * kFLd + rcol;
            float *ss = sm + LD::oS + 4 * kq * kFLd + rcol;
#pragma unroll
            for (int t = 0; t < NM1; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) x1s[(16 * t + i) * kFLd] = X1[t][i];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) x2s[(16 * t + i) * kFLd] = X2[t][i];
#pragma unroll
            for (int qp = 0; qp < 3; ++qp) {
                ss[(16 * qp + 0) * kFLd] = cur.s0[qp].x; ss[(16 * qp + 1) * kFLd] = cur.s0[qp].y;
                ss[(16 * qp + 2) * kFLd] = cur.s0[qp].z; ss[(16 * qp + 3) * kFLd] = cur.s0[qp].w;
            }
            if (kq == 0) {
                sm[LD::oG + rcol] = g;
                reinterpret_cast<int *>(sm + LD::oG + 64)[rcol] = act;
            }
        }
        // dX2 = g Wq[:, a] relu'(X2)  (dQ has its one non-zero at the taken action)
        f32x4 D2[4];
        {
            const float *wq = sm + LD::oWq + 4 * kq * kFLq + act;
            float *d2s = sm + LD::oD2 + 4 * kq * kFLd + rcol;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    D2[t][i] = X2[t][i] > 0.f ? g * wq[(16 * t + i) * kFLq] : 0.f;
                    d2s[(16 * t + i) * kFLd] = D2[t][i];
                }
        }
        QSTAMP(2);
        // dX1 = (W2 dX2) relu'(X1): A = W2[16 t + m][16 t2 + 4 kq .. + 3] (16-byte LDS reads), B = dX2[t2][i]
        f32x4 D1[NM1];
        {
            const float *w2r = sm + LD::oW2 + n * kFLd + 4 * kq;
#pragma unroll
            for (int t = 0; t < NM1; ++t) D1[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t2 = 0; t2 < 4; ++t2) {
                float wv[NM1][4];
#pragma unroll
                for (int t = 0; t < NM1; ++t) {
                    const float4 w = *reinterpret_cast<const float4 *>(w2r + 16 * t * kFLd + 16 * t2);
                    wv[t][0] = w.x; wv[t][1] = w.y; wv[t][2] = w.z; wv[t][3] = w.w;
                }
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int t = 0; t < NM1; ++t) D1[t] = mfma16(wv[t][c], D2[t2][c], D1[t]);
            }
#pragma unroll
            for (int t = 0; t < NM1; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) D1[t][i] = X1[t][i] > 0.f ? D1[t][i] : 0.f;
        }
        QSTAMP(3);
        __syncthreads();
        QSTAMP(4);
        // ================= phase B: dW2 += X1^T dX2, db2, dWq += X2^T dQ, dbq over the chunk's 64 rows =================
        // contraction step (t', c): lane group kq supplies row 16 t' + 4 kq + c -- the same permutation on both operands
        {
            const float *xa = sm + LD::oX1 + n * kFLd + 4 * kq, *db = sm + LD::oD2 + (16 * wave + n) * kFLd + 4 * kq;
            const float *x2a = sm + LD::oX2 + (16 * wave + n) * kFLd + 4 * kq;
            // the operands of contraction step t' + 1 are requested before the 44 MFMAs of step t' (pinned: see q_nets)
            struct BOps { float bv[4], qv[4], xv[4], av[NM1][4]; };
            auto fetch_b = [&](int tp, BOps &o) {
                const float4 bv4 = *reinterpret_cast<const float4 *>(db + 16 * tp);
                o.bv[0] = bv4.x; o.bv[1] = bv4.y; o.bv[2] = bv4.z; o.bv[3] = bv4.w;
                const float4 gv = *reinterpret_cast<const float4 *>(sm + LD::oG + 16 * tp + 4 * kq);
                const int4 tv = *reinterpret_cast<const int4 *>(sm + LD::oG + 64 + 16 * tp + 4 * kq);
                o.qv[0] = tv.x == n ? gv.x : 0.f; o.qv[1] = tv.y == n ? gv.y : 0.f;
                o.qv[2] = tv.z == n ? gv.z : 0.f; o.qv[3] = tv.w == n ? gv.w : 0.f;
                const float4 xv4 = *reinterpret_cast<const float4 *>(x2a + 16 * tp);
                o.xv[0] = xv4.x; o.xv[1] = xv4.y; o.xv[2] = xv4.z; o.xv[3] = xv4.w;
#pragma unroll
                for (int mi = 0; mi < NM1; ++mi) {
                    const float4 v = *reinterpret_cast<const float4 *>(xa + 16 * mi * kFLd + 16 * tp);
                    o.av[mi][0] = v.x; o.av[mi][1] = v.y; o.av[mi][2] = v.z; o.av[mi][3] = v.w;
                }
            };
            BOps ob[2];
            fetch_b(0, ob[0]);
#pragma unroll
            for (int tp = 0; tp < 4; ++tp) {
                if (tp + 1 < 4) fetch_b(tp + 1, ob[(tp + 1) & 1]);
#if TSC_IQL_SCHED & 2
                __builtin_amdgcn_sched_barrier(0);
#endif
                const BOps &o = ob[tp & 1];
                sb2 += (o.bv[0] + o.bv[1]) + (o.bv[2] + o.bv[3]);
                if (wave == 0) sbq += (o.qv[0] + o.qv[1]) + (o.qv[2] + o.qv[3]);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
#pragma unroll
                    for (int mi = 0; mi < NM1; ++mi) aW2[mi] = mfma16(o.av[mi][c], o.bv[c], aW2[mi]);
                    aWq = mfma16(o.xv[c], o.qv[c], aWq);
                }
#if TSC_IQL_SCHED & 2
                __builtin_amdgcn_sched_barrier(0);
#endif
            }
        }
        QSTAMP(5);
        __syncthreads();
        QSTAMP(6);
        // ================= phase C: dX1 over X1's image =================
        {
            float *x1s = sm + LD::oX1 + 4 * kq * kFLd + rcol;
#pragma unroll
            for (int t = 0; t < NM1; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) x1s[(16 * t + i) * kFLd] = D1[t][i];
        }
        QSTAMP(7);
        __syncthreads();
        QSTAMP(8);
        // ================= phase D: dW1 += S^T dX1, db1 =================
        {
            const float *sa = sm + LD::oS + n * kFLd + 4 * kq, *db = sm + LD::oX1 + n * kFLd + 4 * kq;
            auto wave_part = [&](auto nmf_c) {
                constexpr int NMF = decltype(nmf_c)::value;
                struct DOps { float bw[2][4], bt[4], at[4], aw[NMF][4]; };
                auto fetch_d = [&](int tp, DOps &o) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const float4 v = *reinterpret_cast<const float4 *>(db + 16 * (wave + 4 * j) * kFLd + 16 * tp);
                        o.bw[j][0] = v.x; o.bw[j][1] = v.y; o.bw[j][2] = v.z; o.bw[j][3] = v.w;
                    }
                    const float4 bt4 = *reinterpret_cast<const float4 *>(db + 16 * wt_ni * kFLd + 16 * tp);
                    const float4 at4 = *reinterpret_cast<const float4 *>(sa + 16 * wt_mf * kFLd + 16 * tp);
                    o.bt[0] = bt4.x; o.bt[1] = bt4.y; o.bt[2] = bt4.z; o.bt[3] = bt4.w;
                    o.at[0] = wt_on ? at4.x : 0.f; o.at[1] = wt_on ? at4.y : 0.f; o.at[2] = wt_on ? at4.z : 0.f; o.at[3] = wt_on ? at4.w : 0.f;
#pragma unroll
                    for (int mf = 0; mf < NMF; ++mf) {
                        const float4 v = *reinterpret_cast<const float4 *>(sa + 16 * mf * kFLd + 16 * tp);
                        o.aw[mf][0] = v.x; o.aw[mf][1] = v.y; o.aw[mf][2] = v.z; o.aw[mf][3] = v.w;
                    }
                };
                DOps od[2];
                fetch_d(0, od[0]);
#pragma unroll
                for (int tp = 0; tp < 4; ++tp) {
                    if (tp + 1 < 4) fetch_d(tp + 1, od[(tp + 1) & 1]);
#if TSC_IQL_SCHED & 4
                    __builtin_amdgcn_sched_barrier(0);
#endif
                    const DOps &o = od[tp & 1];
#pragma unroll
                    for (int j = 0; j < 2; ++j) sb1[j] += (o.bw[j][0] + o.bw[j][1]) + (o.bw[j][2] + o.bw[j][3]);
                    if (NM1 > NMW && wave < 2) sb1t += (o.bt[0] + o.bt[1]) + (o.bt[2] + o.bt[3]);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int mf = 0; mf < NMF; ++mf) aW1w[j][mf] = mfma16(o.aw[mf][c], o.bw[j][c], aW1w[j][mf]);
                        if (NM1 > NMW) aW1t = mfma16(o.at[c], o.bt[c], aW1t);
                    }
#if TSC_IQL_SCHED & 4
                    __builtin_amdgcn_sched_barrier(0);
#endif
                }
            };
            wave_part(std::integral_constant<int, QW1<NM1, NMW>::QW>{});       // every feature group the instantiation admits: no branch (see q_nets)
        }
        QSTAMP(9);
        __syncthreads();
        QSTAMP(10);
        cur = nxt;
    }

    // ---- this split's partial gradient, parameter layout; the reduce kernel never reads what is not written here
    float *w = p.ws + ((long long)sp * p.A + a) * p.stride;
    auto fold_kq = [&](float v) { v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); return v; };
#pragma unroll
    for (int mi = 0; mi < NM1; ++mi)
#pragma unroll
        for (int i = 0; i < 4; ++i) w[p.oW2 + (long long)(16 * mi + 4 * kq + i) * kFH2 + 16 * wave + n] = aW2[mi][i];
    {
        const float s2 = fold_kq(sb2);
        if (kq == 0) w[p.ob2 + 16 * wave + n] = s2;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (n < 8) w[p.oWq + (long long)(16 * wave + 4 * kq + i) * 8 + n] = aWq[i];
    {
        const float sq = fold_kq(sbq);
        if (wave == 0 && kq == 0 && n < 8) w[p.obq + n] = sq;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int ni = wave + 4 * j;
#pragma unroll
        for (int mf = 0; mf < 3; ++mf)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int f = 16 * mf + 4 * kq + i;
                if (f < p.SMAX) w[p.oW1 + (long long)f * H1 + 16 * ni + n] = aW1w[j][mf][i];
            }
        const float s1 = fold_kq(sb1[j]);
        if (kq == 0) w[p.ob1 + 16 * ni + n] = s1;
    }
    if (wt_on) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = 16 * wt_mf + 4 * kq + i;
            if (f < p.SMAX) w[p.oW1 + (long long)f * H1 + 16 * wt_ni + n] = aW1t[i];
        }
    }
    if (NM1 > NMW) {
        const float st = fold_kq(sb1t);
        if (wave < 2 && kq == 0) w[p.ob1 + 16 * (NMW + wave) + n] = st;
    }
    if (p.dbg && tid == 0) p.dbg[64 + 2 * blockIdx.x + 1] = wall_clock64();
#undef QSTAMP
    // loss of the split: lanes kq == 0 hold one row each
    __syncthreads();
    {
        float l = loss;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) l += __shfl_xor(l, o, 64);
        if (lane == 0) sm[LD::oG + wave] = l;
        __syncthreads();
        if (tid == 0) p.wsl[(long long)sp * p.A + a] = (sm[LD::oG] + sm[LD::oG + 1]) + (sm[LD::oG + 2] + sm[LD::oG + 3]);
    }
}

// grads[a] = sum over splits (in split order) of the partial gradients, W1's structural zeros applied; stats[a][0] = loss
__global__ void iql_fused_reduce_kernel(const float *__restrict__ ws, const float *__restrict__ wsl, int A, int S, long long stride,
                                        long long ob1, int H1, const int16_t *__restrict__ rr, int SMAX, float *__restrict__ grads,
                                        double *__restrict__ stats) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int a = blockIdx.y;
    if (i < stride) {
        bool live = true;
        if (i < ob1) {                                   // W1 [SMAX][H1]: keep only columns [lo, hi) of the feature's row
            const int f = (int)(i / H1), c = (int)(i % H1);
            const int16_t *q = rr + ((long long)a * SMAX + f) * 2;
            live = c >= q[0] && c < q[1];
        }
        float acc = 0.f;
        if (live)
            for (int s = 0; s < S; ++s) acc += ws[((long long)s * A + a) * stride + i];
        grads[(long long)a * stride + i] = acc;
    }
    if (i == 0) {
        double l = 0.0;
        for (int s = 0; s < S; ++s) l += (double)wsl[(long long)s * A + a];
        stats[a * 2] = l;
    }
}

// ---- IQL.forward (agents/models.py:332-348) in one launch: the same nets on the E acting rows + the action choice -------
__device__ __forceinline__ unsigned long long qf_splitmix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ double qf_uniform01(unsigned long long seed, unsigned long long step, unsigned long long idx) {
    const unsigned long long h = qf_splitmix64(qf_splitmix64(seed ^ (step * 0xD1B54A32D192ED03ull)) + idx);
    return (double)(h >> 11) * (1.0 / 9007199254740992.0);
}

template <int NM1, int NMW>
__global__ void __launch_bounds__(256, 1) iql_fused_act_kernel(QFusedArgs p, const float *__restrict__ obs, int mode, double eps,
                                                               unsigned long long seed, unsigned long long step, int AMAX,
                                                               float *__restrict__ Qe, float *__restrict__ q_out, int *__restrict__ action) {
    extern __shared__ __attribute__((aligned(16))) float q_smem[];
    float *sm = q_smem;
    const int a = blockIdx.x % p.A, blk = blockIdx.x / p.A;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kq = lane >> 4;
    const float *P = p.params + (long long)a * p.stride;
    const int na = p.n_act[a];
    const QRanges rg = q_ranges(p.n_wave[a], p.n_wait[a]);
    long long e = (long long)blk * 64 + 16 * wave + n;
    const bool ok = e < p.E;
    if (!ok) e = p.E - 1;
    const float *row = obs + (e * p.A + a) * p.SMAX;
    float4 s[3];
#pragma unroll
    for (int qp = 0; qp < 3; ++qp) s[qp] = q_obs4(row, 16 * qp + 4 * kq, p.SMAX, ok);
    q_stage_weights<NM1>(P, p, sm, tid, 256);
    QW1<NM1, NMW> w1;
    q_load_w1<NM1, NMW>(P, p, rg, n, kq, w1);
    __syncthreads();
    f32x4 X1[1][NM1], X2[1][4], qo[1];
    {
        float4 ss[1][3] = {{s[0], s[1], s[2]}};
        q_nets<1, NM1, NMW>(w1, ss, sm, rg, n, kq, X1, X2, qo);
    }
    const f32x4 q = qo[0];
    // the row's eight Q values into its kq == 0 lane
    float qv[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) { qv[i] = q[i]; qv[4 + i] = __shfl_down(q[i], 16, 64); }
    if (kq != 0 || !ok) return;
    const long long idx = e * p.A + a;
    float *qe = Qe + ((long long)a * p.E + e) * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) qe[k] = qv[k];
    for (int k = 0; k < AMAX; ++k) {
        float v = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) if (j == k && k < na) v = qv[j];
        q_out[idx * AMAX + k] = v;
    }
    int best = 0;
    float qbest = qv[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) if (k < na && qv[k] > qbest) { qbest = qv[k]; best = k; }       // np.argmax: first maximum
    int act = best;
    if (mode == 1) {
        const double u0 = qf_uniform01(seed, step, 2ull * idx), u1 = qf_uniform01(seed, step, 2ull * idx + 1);
        if (u0 < eps) { act = (int)(u1 * (double)na); if (act >= na) act = na - 1; }
    } else if (mode == 2) {
        double sum = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) if (k < na) sum += (double)qv[k];
        const double u = qf_uniform01(seed, step, 2ull * idx);
        double cdf = 0.0, tot = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) if (k < na) tot += (double)qv[k] / sum;
        act = na - 1;
        bool found = false;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k < na && !found) { cdf += (double)qv[k] / sum; if (u < cdf / tot) { act = k; found = true; } }
    }
    action[idx] = act;
}

}  // namespace
