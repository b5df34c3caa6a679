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
template <int NM1, int NMW>
__device__ __forceinline__ void q_nets(const float (&w1)[NM1][3][4], const float4 (&s)[3], const float *sm, const QRanges &rg, int m, int kq,
                                       f32x4 (&X1)[NM1], f32x4 (&X2)[4], f32x4 &q) {
    using LD = QFusedLds<NM1>;
    constexpr int H1 = LD::H1;
    const float *Bs = sm + LD::oB;
#pragma unroll
    for (int t = 0; t < NM1; ++t) {
        const float4 b = *reinterpret_cast<const float4 *>(Bs + 16 * t + 4 * kq);
        X1[t] = f32x4{b.x, b.y, b.z, b.w};
    }
#pragma unroll
    for (int qp = 0; qp < 3; ++qp) {
        const float sv[4] = {s[qp].x, s[qp].y, s[qp].z, s[qp].w};
        if (qp < rg.qw1) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int t = 0; t < NMW; ++t) X1[t] = mfma16(w1[t][qp][c], sv[c], X1[t]);
        }
        if (NM1 > NMW && qp >= rg.qt0 && qp < rg.qt1) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int t = NMW; t < NM1; ++t) X1[t] = mfma16(w1[t][qp][c], sv[c], X1[t]);
        }
    }
#pragma unroll
    for (int t = 0; t < NM1; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) X1[t][i] = X1[t][i] > 0.f ? X1[t][i] : 0.f;
    // second layer: A = W2[k = 16 kt + 4 kq + i][out 16 t2 + m] from LDS (consecutive lanes, consecutive banks), B = X1[kt][i]
#pragma unroll
    for (int t2 = 0; t2 < 4; ++t2) {
        const float4 b = *reinterpret_cast<const float4 *>(Bs + H1 + 16 * t2 + 4 * kq);
        X2[t2] = f32x4{b.x, b.y, b.z, b.w};
    }
    const float *W2s = sm + LD::oW2 + 4 * kq * kFLd + m;
#pragma unroll
    for (int kt = 0; kt < NM1; ++kt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float *wr = W2s + (16 * kt + i) * kFLd;
#pragma unroll
            for (int t2 = 0; t2 < 4; ++t2) X2[t2] = mfma16(wr[16 * t2], X1[kt][i], X2[t2]);
        }
#pragma unroll
    for (int t2 = 0; t2 < 4; ++t2)
#pragma unroll
        for (int i = 0; i < 4; ++i) X2[t2][i] = X2[t2][i] > 0.f ? X2[t2][i] : 0.f;
    // Q^T [16 (8 used) actions][16 rows]: two accumulators (the 16x16x4 form's dependent latency is 40 cycles)
    const float4 bq = *reinterpret_cast<const float4 *>(Bs + H1 + kFH2 + 4 * kq);
    f32x4 qa = f32x4{bq.x, bq.y, bq.z, bq.w}, qb = f32x4{0.f, 0.f, 0.f, 0.f};
    const float *Wqs = sm + LD::oWq + 4 * kq * kFLq + m;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        qa = mfma16(Wqs[(0 + i) * kFLq], X2[0][i], qa);
        qb = mfma16(Wqs[(16 + i) * kFLq], X2[1][i], qb);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        qa = mfma16(Wqs[(32 + i) * kFLq], X2[2][i], qa);
        qb = mfma16(Wqs[(48 + i) * kFLq], X2[3][i], qb);
    }
    q = qa + qb;
}

template <int NM1>
__device__ __forceinline__ void q_load_w1(const float *__restrict__ P, const QFusedArgs &p, int m, int kq, float (&w1)[NM1][3][4]) {
    constexpr int H1 = 16 * NM1;
#pragma unroll
    for (int t = 0; t < NM1; ++t)
#pragma unroll
        for (int qp = 0; qp < 3; ++qp)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int f = 16 * qp + 4 * kq + c;
                w1[t][qp][c] = f < p.SMAX ? P[p.oW1 + (long long)f * H1 + 16 * t + m] : 0.f;
            }
}

// 16 bytes of an observation row (features 16 q + 4 kq ..): zero past the row's end
__device__ __forceinline__ float4 q_obs4(const float *row, int f0, int SMAX, bool ok) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok && f0 < SMAX) v = *reinterpret_cast<const float4 *>(row + f0);
    return v;
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
    float w1[NM1][3][4];
    q_load_w1<NM1>(P, p, n, kq, w1);

    // ---- this wavefront's tiles of the weight gradients (all of them live in registers over the whole slice)
    // dW2 [H1][64]: row tiles mi = wave + 4 j (j < 3), all four column tiles; dWq [64][8]: row tiles 2 (wave - 2) .. + 1 on wavefronts 2, 3
    // dW1 [48][H1]: column tiles ni = wave + 4 j, the three feature tiles
    f32x4 aW2[3][4], aWq[2], aW1[3][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
#pragma unroll
        for (int t = 0; t < 4; ++t) aW2[j][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 3; ++t) aW1[j][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    aWq[0] = aWq[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    float sb2[4] = {0.f, 0.f, 0.f, 0.f}, sbq = 0.f, sb1[3] = {0.f, 0.f, 0.f}, loss = 0.f;
    const long long nchunks = (p.R + 63) >> 6;
    const long long c0 = (long long)sp * p.cps;
    long long c1 = c0 + p.cps;
    if (c1 > nchunks) c1 = nchunks;
    const float fR = (float)p.R;

    struct Rows { float4 s0[3], s1[3]; float rew; int act, done; bool ok; };
    auto slot_of = [&](long long c) -> int {             // ring slot of this lane's row in chunk c (clamped like the gather kernel's)
        long long row = (c << 6) + 16 * wave + n;
        if (row >= p.R) row = p.R - 1;
        const long long e = row / p.B;
        int s = p.idx[(e * p.A + a) * p.B + row % p.B];
        return s < 0 ? 0 : s >= p.size ? p.size - 1 : s;
    };
    auto load_rows = [&](long long c, int slot, Rows &r) {
        long long row = (c << 6) + 16 * wave + n;
        r.ok = row < p.R && c < c1;
        if (row >= p.R) row = p.R - 1;
        const long long e = row / p.B;
        const long long tr = (e * p.cap + slot) * p.A + a;
        const float *o = p.r_obs + tr * p.SMAX, *o1 = p.r_next + tr * p.SMAX;
#pragma unroll
        for (int qp = 0; qp < 3; ++qp) {
            r.s0[qp] = q_obs4(o, 16 * qp + 4 * kq, p.SMAX, r.ok);
            r.s1[qp] = q_obs4(o1, 16 * qp + 4 * kq, p.SMAX, r.ok);
        }
        r.rew = p.r_rew[tr];
        r.act = p.r_act[tr];
        r.done = p.r_done[e * p.cap + slot];
    };

    Rows cur, nxt;
    int slot_n = 0;
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
        f32x4 X1[NM1], X2[4], q;
        q_nets<NM1, NMW>(w1, cur.s1, sm, rg, n, kq, X1, X2, q);       // Q(s'): only its maximum survives
        float q1 = -INFINITY;
#pragma unroll
        for (int i = 0; i < 4; ++i) if (kq < 2 && 4 * kq + i < na) q1 = fmaxf(q1, q[i]);
        q1 = fmaxf(q1, __shfl_xor(q1, 16, 64));
        q1 = fmaxf(q1, __shfl_xor(q1, 32, 64));
        q_nets<NM1, NMW>(w1, cur.s0, sm, rg, n, kq, X1, X2, q);       // Q(s) with the activations the backward pass needs
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
        __syncthreads();
        // ================= phase B: dW2 += X1^T dX2, db2, dWq += X2^T dQ, dbq over the chunk's 64 rows =================
        // contraction step (t', c): lane group kq supplies row 16 t' + 4 kq + c -- the same permutation on both operands
        {
            const float *xa = sm + LD::oX1 + n * kFLd + 4 * kq, *db = sm + LD::oD2 + n * kFLd + 4 * kq;
            const float *x2a = sm + LD::oX2 + n * kFLd + 4 * kq;
#pragma unroll
            for (int tp = 0; tp < 4; ++tp) {
                float4 bv[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) bv[t] = *reinterpret_cast<const float4 *>(db + 16 * t * kFLd + 16 * tp);
                if (wave == 0) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) sb2[t] += (bv[t].x + bv[t].y) + (bv[t].z + bv[t].w);
                }
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int mi = wave + 4 * j;
                    if (mi < NM1) {
                        const float4 av4 = *reinterpret_cast<const float4 *>(xa + 16 * mi * kFLd + 16 * tp);
                        const float av[4] = {av4.x, av4.y, av4.z, av4.w};
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            aW2[j][0] = mfma16(av[c], c == 0 ? bv[0].x : c == 1 ? bv[0].y : c == 2 ? bv[0].z : bv[0].w, aW2[j][0]);
                            aW2[j][1] = mfma16(av[c], c == 0 ? bv[1].x : c == 1 ? bv[1].y : c == 2 ? bv[1].z : bv[1].w, aW2[j][1]);
                            aW2[j][2] = mfma16(av[c], c == 0 ? bv[2].x : c == 1 ? bv[2].y : c == 2 ? bv[2].z : bv[2].w, aW2[j][2]);
                            aW2[j][3] = mfma16(av[c], c == 0 ? bv[3].x : c == 1 ? bv[3].y : c == 2 ? bv[3].z : bv[3].w, aW2[j][3]);
                        }
                    }
                }
                if (wave >= 2) {
                    const float4 gv = *reinterpret_cast<const float4 *>(sm + LD::oG + 16 * tp + 4 * kq);
                    const int4 av = *reinterpret_cast<const int4 *>(sm + LD::oG + 64 + 16 * tp + 4 * kq);
                    const float b0 = av.x == n ? gv.x : 0.f, b1 = av.y == n ? gv.y : 0.f, b2 = av.z == n ? gv.z : 0.f,
                                b3 = av.w == n ? gv.w : 0.f;
                    if (wave == 2) sbq += (b0 + b1) + (b2 + b3);
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const float4 xv = *reinterpret_cast<const float4 *>(x2a + 16 * (2 * (wave - 2) + j) * kFLd + 16 * tp);
                        aWq[j] = mfma16(xv.x, b0, aWq[j]);
                        aWq[j] = mfma16(xv.y, b1, aWq[j]);
                        aWq[j] = mfma16(xv.z, b2, aWq[j]);
                        aWq[j] = mfma16(xv.w, b3, aWq[j]);
                    }
                }
            }
        }
        __syncthreads();
        // ================= phase C: dX1 over X1's image =================
        {
            float *x1s = sm + LD::oX1 + 4 * kq * kFLd + rcol;
#pragma unroll
            for (int t = 0; t < NM1; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) x1s[(16 * t + i) * kFLd] = D1[t][i];
        }
        __syncthreads();
        // ================= phase D: dW1 += S^T dX1, db1 =================
        {
            const float *sa = sm + LD::oS + n * kFLd + 4 * kq, *db = sm + LD::oX1 + n * kFLd + 4 * kq;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int ni = wave + 4 * j;
                if (ni < NM1) {
#pragma unroll
                    for (int tp = 0; tp < 4; ++tp) {
                        const float4 bv = *reinterpret_cast<const float4 *>(db + 16 * ni * kFLd + 16 * tp);
                        sb1[j] += (bv.x + bv.y) + (bv.z + bv.w);
                        float av[3][4];
#pragma unroll
                        for (int mf = 0; mf < 3; ++mf) {
                            const float4 v = *reinterpret_cast<const float4 *>(sa + 16 * mf * kFLd + 16 * tp);
                            av[mf][0] = v.x; av[mf][1] = v.y; av[mf][2] = v.z; av[mf][3] = v.w;
                        }
                        const float bc[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
                        for (int c = 0; c < 4; ++c)
#pragma unroll
                            for (int mf = 0; mf < 3; ++mf) aW1[j][mf] = mfma16(av[mf][c], bc[c], aW1[j][mf]);
                    }
                }
            }
        }
        __syncthreads();
        cur = nxt;
    }

    // ---- this split's partial gradient, parameter layout; the reduce kernel never reads what is not written here
    float *w = p.ws + ((long long)sp * p.A + a) * p.stride;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int mi = wave + 4 * j;
        if (mi < NM1) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) w[p.oW2 + (long long)(16 * mi + 4 * kq + i) * kFH2 + 16 * t + n] = aW2[j][t][i];
#pragma unroll
            for (int mf = 0; mf < 3; ++mf)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int f = 16 * mf + 4 * kq + i;
                    if (f < p.SMAX) w[p.oW1 + (long long)f * H1 + 16 * mi + n] = aW1[j][mf][i];
                }
            float s = sb1[j];
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            if (kq == 0) w[p.ob1 + 16 * mi + n] = s;
        }
    }
    if (wave == 0) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float s = sb2[t];
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            if (kq == 0) w[p.ob2 + 16 * t + n] = s;
        }
    }
    if (wave >= 2) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (n < 8) w[p.oWq + (long long)(16 * (2 * (wave - 2) + j) + 4 * kq + i) * 8 + n] = aWq[j][i];
    }
    if (wave == 2) {
        float s = sbq;
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        if (kq == 0 && n < 8) w[p.obq + n] = s;
    }
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
    float w1[NM1][3][4];
    q_load_w1<NM1>(P, p, n, kq, w1);
    __syncthreads();
    f32x4 X1[NM1], X2[4], q;
    q_nets<NM1, NMW>(w1, s, sm, rg, n, kq, X1, X2, q);
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
