// tsc_iql.hip -- independent Q-learning agents (IQL-LR / IQL-DNN) on gfx950.
//
// Replaces, for all agents of all env instances at once:
//   IQL.forward ............ agents/models.py:332-348  (epsilon-greedy over per-agent Q nets)
//   IQL.add_transition ..... agents/models.py:354-361  (reward norm / clip) + ReplayBuffer (agents/utils.py:231-263)
//   IQL.backward ........... agents/models.py:319-330  -> QPolicy.prepare_loss (agents/policies.py:305-328):
//                            loss = mean((Q(s)[a] - stop_grad(done ? r : r + gamma max Q(s')))^2) with the SAME network for
//                            Q(s') (no target network), tf.clip_by_global_norm per agent, tf.train.AdamOptimizer
//   LRQPolicy / DeepQPolicy  agents/policies.py:343-389: q = fc(S -> n_a)  /
//                            [relu(fc(wave -> n_fc0)), relu(fc(wait -> n_fc0/4))] -> relu(fc(. -> n_h)) -> fc(-> n_a)
//
// Batched over E env instances the way the A2C learner is: every instance keeps its own ring of `buffer_size`
// transitions per agent; one minibatch step draws `batch_size` distinct transitions from EVERY instance's ring
// (counter-based Floyd sampling, one draw per (instance, agent)) and the loss is the mean over the E * batch_size rows of
// an agent -- E = 1 is the reference.  Parameters, gradients and both Adam moments share one flat layout per agent
//   DQN: W1[SMAX][H1] | b1[H1] | W2[H1][H2] | b2[H2] | Wq[H2][8] | bq[8]     (W1 block-diagonal: wave rows -> columns
//        [0, n_fc0), wait rows -> [n_fc0, H1); structural zeros kept zero by the row-range mask on its gradient)
//   LR : Wq[SMAX][8] | bq[8]
// so the gradient buffer is one contiguous all-reduce.  Every contraction runs on the grouped fp32 MFMA GEMM of
// tsc_gemm.h (groups = agents); the element-wise pieces (sampling, gather, TD target, Adam) are small HBM-bound kernels.
#include "tsc_common.h"
#include "tsc_gemm.h"
#include "../../include/tsc.h"

#include <cmath>
#include <cstdlib>
#include <vector>

namespace {

using tsc::GemmArgs;
constexpr int kQ = 8;              // padded action width (n_a <= 8)

struct QLayout {
    int A, SMAX, AMAX, H1, H2, dqn;
    long long stride, oW1, ob1, oW2, ob2, oWq, obq;
};

__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
// the uniform tsc_model_sample documents: U(seed, step, idx)
__device__ __forceinline__ double uniform01(unsigned long long seed, unsigned long long step, unsigned long long idx) {
    const unsigned long long h = splitmix64(splitmix64(seed ^ (step * 0xD1B54A32D192ED03ull)) + idx);
    return (double)(h >> 11) * (1.0 / 9007199254740992.0);
}

// IQL.forward (agents/models.py:332-348): mode 0 = argmax, 1 = explore (np.random.random() < eps -> randint), 2 = stochastic
// (qs / sum(qs) -> np.random.choice).  One thread per (instance, agent); q rows come from the Q GEMM.
__global__ void iql_act_kernel(const float *Q, const int *n_act, int E, int A, int AMAX, int mode, double eps,
                               unsigned long long seed, unsigned long long step, float *q_out, int *action) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= E * A) return;
    const int e = idx / A, a = idx % A, na = n_act[a];
    const float *q = Q + ((long long)a * E + e) * kQ;
    for (int k = 0; k < AMAX; ++k) q_out[(long long)idx * AMAX + k] = k < na ? q[k] : 0.f;
    int best = 0;
    for (int k = 1; k < na; ++k) if (q[k] > q[best]) best = k;          // np.argmax: first maximum
    int act = best;
    if (mode == 1) {
        const double u0 = uniform01(seed, step, 2ull * idx), u1 = uniform01(seed, step, 2ull * idx + 1);
        if (u0 < eps) { act = (int)(u1 * (double)na); if (act >= na) act = na - 1; }
    } else if (mode == 2) {
        double s = 0.0;
        for (int k = 0; k < na; ++k) s += (double)q[k];
        const double u = uniform01(seed, step, 2ull * idx);
        double c = 0.0, tot = 0.0;
        for (int k = 0; k < na; ++k) tot += (double)q[k] / s;
        act = na - 1;
        for (int k = 0; k < na; ++k) { c += (double)q[k] / s; if (u < c / tot) { act = k; break; } }
    }
    action[idx] = act;
}

// ReplayBuffer.add_transition for slot `slot` of every instance's ring; rewards normalised / clipped in float64
// (agents/models.py:355-358) and stored as the float32 the TF placeholder holds.
__global__ void iql_add_kernel(int E, int A, int SMAX, long long cap, long long slot, const float *obs, const int *action,
                               const double *reward, const float *next_obs, const uint8_t *done, double rnorm, double rclip,
                               float *r_obs, float *r_next, int *r_act, float *r_rew, uint8_t *r_done) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long per = (long long)A * SMAX;
    if (i < (long long)E * per) {
        const long long e = i / per, j = i % per;
        r_obs[(e * cap + slot) * per + j] = obs[i];
        r_next[(e * cap + slot) * per + j] = next_obs[i];
    }
    if (i < (long long)E * A) {
        const long long e = i / A, a = i % A;
        double r = reward[i];
        if (rnorm != 0.0) r = r / rnorm;
        if (rclip != 0.0) r = fmin(fmax(r, -rclip), rclip);
        r_rew[(e * cap + slot) * A + a] = (float)r;
        r_act[(e * cap + slot) * A + a] = action[i];
    }
    if (i < E) r_done[i * cap + slot] = done[i];
}

// random.sample(buffer, batch_size) per (instance, agent) (agents/utils.py:251-253), as Floyd's algorithm on the
// documented counter-based uniform: for i in [0, B): j = size - B + i; t = floor(U * (j + 1)); pick t, or j if t was
// picked before.  idx [E][A][B].
__global__ void iql_sample_kernel(int E, int A, int B, long long size, unsigned long long seed, unsigned long long upd, int *idx) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= E * A) return;
    int *out = idx + (long long)p * B;
    for (int i = 0; i < B; ++i) {
        const long long j = size - B + i;
        const double u = uniform01(seed, upd, (unsigned long long)p * B + i);
        long long t = (long long)(u * (double)(j + 1));
        if (t > j) t = j;
        bool seen = false;
        for (int q = 0; q < i; ++q) seen |= out[q] == (int)t;
        out[i] = seen ? (int)j : (int)t;
    }
}
// the same draw for a compile-time batch size (the reference's 20, config/config_iql*.ini): the picks stay in registers instead of
// being re-read from the index buffer for every membership test (16.5 -> ~ 5 us at E = 1024)
template <int BB>
__global__ void iql_sample_fixed_kernel(int E, int A, long long size, unsigned long long seed, unsigned long long upd, int *idx) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= E * A) return;
    int pk[BB];
#pragma unroll
    for (int i = 0; i < BB; ++i) {
        const long long j = size - BB + i;
        const double u = uniform01(seed, upd, (unsigned long long)p * BB + i);
        long long t = (long long)(u * (double)(j + 1));
        if (t > j) t = j;
        bool seen = false;
#pragma unroll
        for (int q = 0; q < i; ++q) seen |= pk[q] == (int)t;
        pk[i] = seen ? (int)j : (int)t;
    }
    int *out = idx + (long long)p * BB;
#pragma unroll
    for (int i = 0; i < BB; ++i) out[i] = pk[i];
}

// minibatch rows of agent a: row = e * B + i  <-  transition idx[e][a][i] of instance e
// (a caller-supplied index outside the filled part [0, size) of the ring is clamped into it: no out-of-bounds read)
__global__ void iql_gather_kernel(int E, int A, int SMAX, int B, long long cap, int size, const int *idx, const float *r_obs,
                                  const float *r_next, const int *r_act, const float *r_rew, const uint8_t *r_done, float *S,
                                  float *S1, int *act, float *rew, uint8_t *done) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int q4 = SMAX >> 2;
    const long long R = (long long)E * B, tot = (long long)A * R * q4;
    if (i < tot) {
        const int c = (int)(i % q4);
        const long long row = (i / q4) % R, a = i / ((long long)q4 * R);
        const long long e = row / B;
        int s = idx[(e * A + a) * B + row % B];
        s = s < 0 ? 0 : s >= size ? size - 1 : s;
        const long long src = ((e * cap + s) * A + a) * SMAX + 4 * c;
        reinterpret_cast<float4 *>(S)[i] = *reinterpret_cast<const float4 *>(r_obs + src);
        reinterpret_cast<float4 *>(S1)[i] = *reinterpret_cast<const float4 *>(r_next + src);
    }
    if (i < (long long)A * R) {
        const long long row = i % R, a = i / R, e = row / B;
        int s = idx[(e * A + a) * B + row % B];
        s = s < 0 ? 0 : s >= size ? size - 1 : s;
        act[i] = r_act[(e * cap + s) * A + a];
        rew[i] = r_rew[(e * cap + s) * A + a];
        done[i] = r_done[e * cap + s];
    }
}

// q1 = max_k Q(s')[k] over the agent's actions (agents/policies.py:315-316)
__global__ void iql_qmax_kernel(const float *Q, const int *n_act, long long R, int A, float *q1) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)A * R) return;
    const int na = n_act[i / R];
    const float *q = Q + i * kQ;
    float m = q[0];
    for (int k = 1; k < na; ++k) m = fmaxf(m, q[k]);
    q1[i] = m;
}

// tq = done ? r : r + gamma q1 ; loss = mean((q0 - tq)^2) ; dQ[k] = 2 (q0 - tq) / R at k = a   (agents/policies.py:317-318)
__global__ void iql_td_kernel(const float *Q, const float *q1, const int *act, const float *rew, const uint8_t *done,
                              long long R, int A, float gamma, float *dQ, double *stats) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    float l = 0.f;
    int a = 0;
    if (i < (long long)A * R) {
        a = (int)(i / R);
        const float r = rew[i];
        const float tq = done[i] ? r : r + gamma * q1[i];
        const int k0 = act[i];
        const float d = Q[i * kQ + k0] - tq;
        const float g = 2.0f * d / (float)R;
        float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
        float *v = k0 < 4 ? &lo.x : &hi.x;
        v[k0 & 3] = g;
        reinterpret_cast<float4 *>(dQ + i * kQ)[0] = lo;
        reinterpret_cast<float4 *>(dQ + i * kQ)[1] = hi;
        l = d * d / (float)R;
    }
    // logging only (policies.py:330-337): rows of one agent are contiguous, a wave may straddle two agents -> per-lane atomics
    // are avoided by reducing only when the whole wave belongs to one agent
    const int a0 = __shfl(a, 0, 64);
    const bool uni = __all(a == a0 || i >= (long long)A * R);
    if (uni) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) l += __shfl_down(l, o, 64);
        if ((threadIdx.x & 63) == 0 && l != 0.f) atomicAdd(&stats[a0 * 2], (double)l);
    } else if (l != 0.f) {
        atomicAdd(&stats[a * 2], (double)l);
    }
}

__global__ void iql_transpose_kernel(const float *params, QLayout L, float *W2T, float *WqT) {
    // W2T[a][n][k] = W2[a][k][n]  (H2 x H1) ; WqT[a][n][k] = Wq[a][k][n]  (8 x H2)
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long p2 = (long long)L.H1 * L.H2, pq = (long long)L.H2 * kQ;
    if (i < p2 * L.A) {
        const long long a = i / p2, r = i % p2;
        const int n = (int)(r / L.H1), k = (int)(r % L.H1);
        W2T[i] = params[a * L.stride + L.oW2 + (long long)k * L.H2 + n];
    }
    if (i < pq * L.A) {
        const long long a = i / pq, r = i % pq;
        const int n = (int)(r / L.H2), k = (int)(r % L.H2);
        WqT[i] = params[a * L.stride + L.oWq + (long long)k * kQ + n];
    }
}

constexpr int kNormSlices = 8;     // an agent's squared norm is summed in 8 slices (one workgroup each), folded in slice order by the readers
__global__ void iql_norm_kernel(const float *grad, long long per_agent, double gscale, double *norm2) {
    __shared__ double red[256];
    const int a = blockIdx.x / kNormSlices, sl = blockIdx.x % kNormSlices;
    const float *gp = grad + (long long)a * per_agent;
    const long long lo = per_agent * sl / kNormSlices, hi = per_agent * (sl + 1) / kNormSlices;
    double s = 0.0;
    for (long long i = lo + threadIdx.x; i < hi; i += 256) { const double v = (double)gp[i] * gscale; s += v * v; }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) norm2[blockIdx.x] = red[0];
}
__device__ __forceinline__ double iql_norm2_of(const double *norm2, long long a) {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < kNormSlices; ++k) s += norm2[a * kNormSlices + k];
    return s;
}

// tf.train.AdamOptimizer (TF 1.12 defaults beta1 .9, beta2 .999, epsilon 1e-8):
//   lr_t = lr sqrt(1 - b2^t) / (1 - b1^t);  m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;  w -= lr_t m / (sqrt(v) + eps)
__global__ void iql_adam_kernel(float *w, float *m1, float *m2, const float *grad, long long per_agent, long long total,
                                const double *norm2, float gscale, float clip, float lr_t) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const float nrm = (float)sqrt(iql_norm2_of(norm2, i / per_agent));
    float g = grad[i] * gscale;
    if (clip > 0.f) g = g * (clip / fmaxf(nrm, clip));
    const float m = 0.9f * m1[i] + (1.0f - 0.9f) * g;
    const float v = 0.999f * m2[i] + (1.0f - 0.999f) * g * g;
    m1[i] = m; m2[i] = v;
    w[i] = w[i] - lr_t * m / (sqrtf(v) + 1e-8f);
}

}  // namespace

#include "tsc_iql_fused.h"

struct tsc_iql {
    QLayout lay;
    int E, B, device;
    long long cap, cum;           // ring capacity / transitions added so far (per instance)
    double gamma, rnorm, rclip, max_norm;
    long long adam_t;
    hipStream_t stream;
    std::vector<void *> allocs;
    int *n_act;
    int16_t *rowrange;            // [A][SMAX][2]
    float *params, *grads, *m1, *m2, *W2T, *WqT;
    float *r_obs, *r_next, *r_rew; int *r_act; uint8_t *r_done;
    int *idx;
    float *S, *S1, *rew, *q1; int *act; uint8_t *done;
    float *X1, *X2, *Q, *dQ, *dX2;
    float *Qe;                    // [A][E][8] q rows of the acting forward
    float *X1e, *X2e;
    double *norm2, *stats;
    float *ws, *wsc; size_t ws_floats, wsc_floats;
    long long nparam;
    // fused DeepQPolicy learner (tsc_iql_fused.h): 0 = grouped-GEMM path, 8 / 10 = first-layer column tiles of the instantiation
    int fused, fS, fcps;
    int *n_wave, *n_wait;
    float *fws, *fwsl;
    long long *dbg;
};

namespace {

int qgemm(tsc_iql *h, bool tn, int epi, int M, int N, int K, const float *A, long long sA, int lda, const float *B, long long sB,
          int ldb, float *C, long long sC, int ldc, const float *bias, long long sBias, const float *aux, long long sAux,
          int ldaux, const int16_t *rr, long long sRR, float *colsum, long long sColsum) {
    GemmArgs a;
    a.A = A; a.B = B; a.C = C; a.bias = bias; a.aux = aux; a.rr = rr; a.colsum = colsum;
    a.sA = sA; a.sB = sB; a.sC = sC; a.sBias = sBias; a.sAux = sAux; a.sRR = sRR; a.sColsum = sColsum;
    a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ldaux = ldaux; a.M = M; a.N = N; a.K = K; a.gdivA = 1;
    tsc::plan_splitk(a, h->lay.A, tn ? h->ws : nullptr, h->wsc, h->ws_floats, h->wsc_floats);
    tsc::launch_gemm_dyn(tn, epi, a, h->lay.A, h->stream);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// Q(S) for `rows` rows per agent: S [A][rows][SMAX] -> X1, X2 (DQN) -> Q [A][rows][8]
int q_forward(tsc_iql *h, const float *S, long long sS, int ldS, long long rows, float *X1, float *X2, float *Q) {
    const QLayout &L = h->lay;
    const float *P = h->params;
    if (!L.dqn)
        return qgemm(h, false, tsc::EPI_BIAS, (int)rows, kQ, L.SMAX, S, sS, ldS, P + L.oWq, L.stride, kQ, Q, rows * kQ, kQ,
                     P + L.obq, L.stride, nullptr, 0, 0, nullptr, 0, nullptr, 0);
    if (qgemm(h, false, tsc::EPI_BIAS_RELU, (int)rows, L.H1, L.SMAX, S, sS, ldS, P + L.oW1, L.stride, L.H1, X1, rows * L.H1, L.H1,
              P + L.ob1, L.stride, nullptr, 0, 0, nullptr, 0, nullptr, 0)) return 1;
    if (qgemm(h, false, tsc::EPI_BIAS_RELU, (int)rows, L.H2, L.H1, X1, rows * L.H1, L.H1, P + L.oW2, L.stride, L.H2, X2, rows * L.H2,
              L.H2, P + L.ob2, L.stride, nullptr, 0, 0, nullptr, 0, nullptr, 0)) return 1;
    return qgemm(h, false, tsc::EPI_BIAS, (int)rows, kQ, L.H2, X2, rows * L.H2, L.H2, P + L.oWq, L.stride, kQ, Q, rows * kQ, kQ,
                 P + L.obq, L.stride, nullptr, 0, 0, nullptr, 0, nullptr, 0);
}

QFusedArgs fused_args(const tsc_iql *h, long long size) {
    const QLayout &L = h->lay;
    QFusedArgs fa;
    fa.params = h->params; fa.n_act = h->n_act; fa.n_wave = h->n_wave; fa.n_wait = h->n_wait; fa.idx = h->idx;
    fa.r_obs = h->r_obs; fa.r_next = h->r_next; fa.r_rew = h->r_rew; fa.r_act = h->r_act; fa.r_done = h->r_done;
    fa.E = h->E; fa.A = L.A; fa.B = h->B; fa.SMAX = L.SMAX; fa.size = (int)size; fa.cap = h->cap; fa.R = (long long)h->E * h->B;
    fa.gamma = (float)h->gamma; fa.S = h->fS; fa.cps = h->fcps; fa.ws = h->fws; fa.wsl = h->fwsl;
    fa.dbg = h->dbg;
    fa.stride = L.stride; fa.oW1 = L.oW1; fa.ob1 = L.ob1; fa.oW2 = L.oW2; fa.ob2 = L.ob2; fa.oWq = L.oWq; fa.obq = L.obq;
    return fa;
}

}  // namespace

extern "C" {

#define QMALLOC(ptr, T, count)                                                         \
    do {                                                                               \
        TSC_HIP(hipMalloc((void **)&(ptr), sizeof(T) * (size_t)(count)));              \
        TSC_HIP(hipMemset((ptr), 0, sizeof(T) * (size_t)(count)));                     \
        h->allocs.push_back((void *)(ptr));                                            \
    } while (0)

int tsc_iql_create(const tsc_iql_cfg *cfg, int32_t n_env, int32_t device, tsc_iql **out) {
    if (!cfg || !out || n_env <= 0) return tsc::fail("tsc_iql_create: bad arguments");
    if (cfg->a_max > kQ) return tsc::fail("tsc_iql_create: a_max %d > %d", cfg->a_max, kQ);
    if (cfg->s_max % 4) return tsc::fail("tsc_iql_create: s_max must be a multiple of 4");
    if (cfg->kind != 0 && cfg->kind != 1) return tsc::fail("tsc_iql_create: kind must be 0 (lr) or 1 (dqn)");
    if (cfg->batch_size <= 0 || cfg->batch_size > 64 || cfg->buffer_size < cfg->batch_size)
        return tsc::fail("tsc_iql_create: need 0 < batch_size <= 64 <= buffer_size");
    TSC_HIP(hipSetDevice(device));
    tsc_iql *h = new tsc_iql();
    tsc::CreateGuard<tsc_iql, tsc_iql_destroy> guard(h);        // an error return below frees the handle and its buffers
    h->device = device; h->stream = nullptr; h->E = n_env; h->B = cfg->batch_size; h->cap = cfg->buffer_size; h->cum = 0;
    h->gamma = cfg->gamma; h->rnorm = cfg->reward_norm; h->rclip = cfg->reward_clip; h->max_norm = cfg->max_grad_norm;
    h->adam_t = 0;
    QLayout &L = h->lay;
    L.A = cfg->n_agent; L.SMAX = cfg->s_max; L.AMAX = cfg->a_max; L.dqn = cfg->kind;
    bool any_wait = false;
    for (int a = 0; a < L.A; ++a) any_wait |= cfg->n_wait[a] > 0;
    const int ft = any_wait ? cfg->n_fc0 / 4 : 0;            // q_fct width: n_fc0 / 4 (agents/policies.py:359)
    L.H1 = L.dqn ? cfg->n_fc0 + ft : 0; L.H2 = L.dqn ? cfg->n_h : 0;
    if (L.dqn && (L.H1 % 4 || L.H2 % 4)) return tsc::fail("tsc_iql_create: hidden widths must be multiples of 4");
    if (L.dqn) {
        L.oW1 = 0; L.ob1 = (long long)L.SMAX * L.H1; L.oW2 = L.ob1 + L.H1; L.ob2 = L.oW2 + (long long)L.H1 * L.H2;
        L.oWq = L.ob2 + L.H2; L.obq = L.oWq + (long long)L.H2 * kQ;
    } else {
        L.oW1 = L.ob1 = L.oW2 = L.ob2 = 0; L.oWq = 0; L.obq = (long long)L.SMAX * kQ;
    }
    L.stride = L.obq + kQ;
    h->nparam = L.stride * L.A;
    std::vector<int16_t> rr((size_t)L.A * L.SMAX * 2, 0);
    for (int a = 0; a < L.A; ++a) {
        const int nw = cfg->n_wave[a], nt = cfg->n_wait[a];
        if (nw + nt > L.SMAX) return tsc::fail("tsc_iql_create: agent %d obs wider than s_max", a);
        for (int j = 0; j < L.SMAX; ++j) {
            int lo = 0, hi = 0;
            if (j < nw) { lo = 0; hi = cfg->n_fc0; }
            else if (j < nw + nt) { lo = cfg->n_fc0; hi = cfg->n_fc0 + ft; }
            rr[((size_t)a * L.SMAX + j) * 2] = (int16_t)lo; rr[((size_t)a * L.SMAX + j) * 2 + 1] = (int16_t)hi;
        }
    }
    TSC_HIP(tsc::upload<int16_t>(&h->rowrange, rr.data(), rr.size())); h->allocs.push_back(h->rowrange);
    TSC_HIP(tsc::upload<int>(&h->n_act, cfg->n_act, L.A)); h->allocs.push_back(h->n_act);
    const long long E = n_env, A = L.A, R = E * h->B, per = A * L.SMAX;
    QMALLOC(h->params, float, h->nparam); QMALLOC(h->grads, float, h->nparam);
    QMALLOC(h->m1, float, h->nparam); QMALLOC(h->m2, float, h->nparam);
    QMALLOC(h->r_obs, float, E * h->cap * per); QMALLOC(h->r_next, float, E * h->cap * per);
    QMALLOC(h->r_rew, float, E * h->cap * A); QMALLOC(h->r_act, int, E * h->cap * A); QMALLOC(h->r_done, uint8_t, E * h->cap);
    QMALLOC(h->idx, int, E * A * h->B);
    QMALLOC(h->Qe, float, A * E * kQ);
    QMALLOC(h->norm2, double, A * kNormSlices); QMALLOC(h->stats, double, A * 2);
    // The fused DeepQPolicy learner (tsc_iql_fused.h) is built for the reference's widths (config/config_iqld_*.ini: num_fc 128,
    // num_h 64 -> H1 = 160 with wait inputs, 128 without) and observations of at most 48 features; anything else, IQL-LR, and
    // TSC_IQL_FUSED=0 (the A/B switch of tests/test_iql_gpu.py) take the grouped-GEMM path.
    h->fused = 0; h->fS = h->fcps = 0; h->fws = h->fwsl = nullptr; h->dbg = nullptr;
    TSC_HIP(tsc::upload<int>(&h->n_wave, cfg->n_wave, L.A)); h->allocs.push_back(h->n_wave);
    TSC_HIP(tsc::upload<int>(&h->n_wait, cfg->n_wait, L.A)); h->allocs.push_back(h->n_wait);
    {
        const char *sw = getenv("TSC_IQL_FUSED");
        const bool want = !(sw && sw[0] == '0');
        int max_wave = 0, max_wait = 0;
        for (int a = 0; a < L.A; ++a) {
            max_wave = cfg->n_wave[a] > max_wave ? cfg->n_wave[a] : max_wave;
            max_wait = cfg->n_wait[a] > max_wait ? cfg->n_wait[a] : max_wait;
        }
        // with a wait part the kernel keeps two 16-feature groups of W1 per column tile in registers (tsc_iql_fused.h QW1)
        const bool fits = L.H1 == 128 || (L.H1 == 160 && max_wave <= 32 && max_wait <= 16);
        if (want && L.dqn && cfg->n_fc0 == 128 && L.H2 == kFH2 && L.SMAX <= kFSF && fits) h->fused = L.H1 / 16;
    }
    if (h->fused && R >= ((long long)1 << 31) / 64) h->fused = 0;       // the fused kernel indexes rows and chunks in 32 bits
    if (h->fused) {
        // row splits per agent: one workgroup per CU (the kernel holds its gradient tiles in registers over its whole slice)
        const long long nchunks = (R + 63) / 64;
        long long S = 256 / A;
        if (S < 1) S = 1;
        if (S > nchunks) S = nchunks;
        h->fcps = (int)((nchunks + S - 1) / S);
        h->fS = (int)((nchunks + h->fcps - 1) / h->fcps);
        QMALLOC(h->fws, float, (long long)h->fS * A * L.stride); QMALLOC(h->fwsl, float, (long long)h->fS * A);
        const int lds_g = (h->fused == 10 ? QFusedLds<10>::grad_floats : QFusedLds<8>::grad_floats) * 4;
        const int lds_f = (h->fused == 10 ? QFusedLds<10>::fwd_floats : QFusedLds<8>::fwd_floats) * 4;
        if (h->fused == 10) {
            TSC_HIP(hipFuncSetAttribute((const void *)iql_fused_grad_kernel<10, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_g));
            TSC_HIP(hipFuncSetAttribute((const void *)iql_fused_act_kernel<10, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_f));
        } else {
            TSC_HIP(hipFuncSetAttribute((const void *)iql_fused_grad_kernel<8, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_g));
            TSC_HIP(hipFuncSetAttribute((const void *)iql_fused_act_kernel<8, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_f));
        }
        h->S = h->S1 = h->rew = h->q1 = h->Q = h->dQ = nullptr; h->act = nullptr; h->done = nullptr;
        h->X1 = h->X2 = h->dX2 = h->X1e = h->X2e = h->W2T = h->WqT = nullptr;
        h->ws = h->wsc = nullptr; h->ws_floats = h->wsc_floats = 0;
    } else {
        QMALLOC(h->S, float, A * R * L.SMAX); QMALLOC(h->S1, float, A * R * L.SMAX);
        QMALLOC(h->rew, float, A * R); QMALLOC(h->q1, float, A * R); QMALLOC(h->act, int, A * R); QMALLOC(h->done, uint8_t, A * R);
        QMALLOC(h->Q, float, A * R * kQ); QMALLOC(h->dQ, float, A * R * kQ);
        if (L.dqn) {
            QMALLOC(h->X1, float, A * R * L.H1); QMALLOC(h->X2, float, A * R * L.H2); QMALLOC(h->dX2, float, A * R * L.H2);
            QMALLOC(h->X1e, float, A * E * L.H1); QMALLOC(h->X2e, float, A * E * L.H2);
            QMALLOC(h->W2T, float, A * L.H1 * L.H2); QMALLOC(h->WqT, float, A * L.H2 * kQ);
        } else {
            h->X1 = h->X2 = h->dX2 = h->X1e = h->X2e = h->W2T = h->WqT = nullptr;
        }
        h->ws_floats = (size_t)16 << 20; h->wsc_floats = (size_t)1 << 18;
        QMALLOC(h->ws, float, h->ws_floats); QMALLOC(h->wsc, float, h->wsc_floats);
    }
    *out = guard.release();
    return 0;
}

int tsc_iql_destroy(tsc_iql *h) {
    if (!h) return 0;
    (void)hipSetDevice(h->device);
    for (void *p : h->allocs) (void)hipFree(p);
    delete h;
    return 0;
}

int tsc_iql_set_stream(tsc_iql *h, void *s) {
    if (!h) return tsc::fail("null handle");
    h->stream = (hipStream_t)s;
    return 0;
}

int tsc_iql_layout(tsc_iql *h, int64_t out[12]) {
    if (!h || !out) return tsc::fail("tsc_iql_layout: bad arguments");
    const QLayout &L = h->lay;
    const int64_t v[12] = {L.A, L.stride, L.H1, L.H2, L.oW1, L.ob1, L.oW2, L.ob2, L.oWq, L.obq, kQ, L.dqn};
    for (int i = 0; i < 12; ++i) out[i] = v[i];
    return 0;
}

int tsc_iql_set_params(tsc_iql *h, const float *p) {
    if (!h || !p) return tsc::fail("tsc_iql_set_params: bad arguments");
    TSC_HIP(hipStreamSynchronize(h->stream));
    TSC_HIP(hipMemcpy(h->params, p, sizeof(float) * h->nparam, hipMemcpyHostToDevice));
    return 0;
}
int tsc_iql_get_params(tsc_iql *h, float *p) {
    if (!h || !p) return tsc::fail("tsc_iql_get_params: bad arguments");
    TSC_HIP(hipStreamSynchronize(h->stream));
    TSC_HIP(hipMemcpy(p, h->params, sizeof(float) * h->nparam, hipMemcpyDeviceToHost));
    return 0;
}
int tsc_iql_get_opt_state(tsc_iql *h, float *m, float *v, int64_t *t) {
    if (!h || !m || !v || !t) return tsc::fail("tsc_iql_get_opt_state: bad arguments");
    TSC_HIP(hipStreamSynchronize(h->stream));
    TSC_HIP(hipMemcpy(m, h->m1, sizeof(float) * h->nparam, hipMemcpyDeviceToHost));
    TSC_HIP(hipMemcpy(v, h->m2, sizeof(float) * h->nparam, hipMemcpyDeviceToHost));
    *t = h->adam_t;
    return 0;
}
int tsc_iql_set_opt_state(tsc_iql *h, const float *m, const float *v, int64_t t) {
    if (!h || !m || !v || t < 0) return tsc::fail("tsc_iql_set_opt_state: bad arguments");
    TSC_HIP(hipStreamSynchronize(h->stream));
    TSC_HIP(hipMemcpy(h->m1, m, sizeof(float) * h->nparam, hipMemcpyHostToDevice));
    TSC_HIP(hipMemcpy(h->m2, v, sizeof(float) * h->nparam, hipMemcpyHostToDevice));
    h->adam_t = t;
    return 0;
}

int tsc_iql_forward(tsc_iql *h, const float *obs, float *q_out, int32_t *action, int32_t mode, double eps, uint64_t seed,
                    uint64_t step) {
    if (!h || !obs || !q_out || !action || mode < 0 || mode > 2) return tsc::fail("tsc_iql_forward: bad arguments");
    const QLayout &L = h->lay;
    if (h->fused) {
        QFusedArgs fa = fused_args(h, 0);
        const unsigned grid = (unsigned)(L.A * ((h->E + 63) / 64));
        tsc::ProfScope ps(tsc::KID_IQL_ACT, h->stream);
        if (h->fused == 10)
            hipLaunchKernelGGL((iql_fused_act_kernel<10, 8>), dim3(grid), dim3(256), QFusedLds<10>::fwd_floats * 4, h->stream, fa, obs, (int)mode,
                               eps, (unsigned long long)seed, (unsigned long long)step, L.AMAX, h->Qe, q_out, action);
        else
            hipLaunchKernelGGL((iql_fused_act_kernel<8, 8>), dim3(grid), dim3(256), QFusedLds<8>::fwd_floats * 4, h->stream, fa, obs, (int)mode,
                               eps, (unsigned long long)seed, (unsigned long long)step, L.AMAX, h->Qe, q_out, action);
        TSC_HIP(hipGetLastError());
        return 0;
    }
    // obs [E][A][SMAX]: agent a's rows start at a * SMAX with row stride A * SMAX
    if (q_forward(h, obs, L.SMAX, L.A * L.SMAX, h->E, h->X1e, h->X2e, h->Qe)) return tsc::fail("tsc_iql_forward: gemm launch failed");
    const int tot = h->E * L.A;
    hipLaunchKernelGGL(iql_act_kernel, dim3((tot + 255) / 256), dim3(256), 0, h->stream, h->Qe, h->n_act, h->E, L.A, L.AMAX,
                       (int)mode, eps, (unsigned long long)seed, (unsigned long long)step, q_out, action);
    TSC_HIP(hipGetLastError());
    return 0;
}

int tsc_iql_add_transition(tsc_iql *h, const float *obs, const int32_t *action, const double *reward, const float *next_obs,
                           const uint8_t *done) {
    if (!h || !obs || !action || !reward || !next_obs || !done) return tsc::fail("tsc_iql_add_transition: bad arguments");
    const QLayout &L = h->lay;
    const long long n = (long long)h->E * L.A * L.SMAX;
    tsc::ProfScope ps(tsc::KID_IQL_ADD, h->stream);
    hipLaunchKernelGGL(iql_add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->E, L.A, L.SMAX, h->cap,
                       h->cum % h->cap, obs, action, reward, next_obs, done, h->rnorm, h->rclip, h->r_obs, h->r_next, h->r_act,
                       h->r_rew, h->r_done);
    TSC_HIP(hipGetLastError());
    h->cum += 1;
    return 0;
}

int tsc_iql_replay_size(tsc_iql *h, int64_t *size, int64_t *cum) {
    if (!h || !size || !cum) return tsc::fail("tsc_iql_replay_size: bad arguments");
    *size = h->cum < h->cap ? h->cum : h->cap; *cum = h->cum;
    return 0;
}

static int iql_compute_grads(tsc_iql *h, uint64_t seed, uint64_t update_index, const int32_t *idx_dev);

int tsc_iql_compute_grads(tsc_iql *h, uint64_t seed, uint64_t update_index) {
    return iql_compute_grads(h, seed, update_index, nullptr);
}

int tsc_iql_compute_grads_at(tsc_iql *h, const int32_t *idx_dev) {
    if (!idx_dev) return tsc::fail("tsc_iql_compute_grads_at: null index buffer");
    return iql_compute_grads(h, 0, 0, idx_dev);
}

static int iql_compute_grads(tsc_iql *h, uint64_t seed, uint64_t update_index, const int32_t *idx_dev) {
    if (!h) return tsc::fail("null handle");
    const QLayout &L = h->lay;
    const long long size = h->cum < h->cap ? h->cum : h->cap;
    if (size < h->B) return tsc::fail("tsc_iql_compute_grads: replay holds %lld < batch_size %d transitions", size, h->B);
    hipStream_t st = h->stream;
    const long long E = h->E, A = L.A, R = E * h->B;
    TSC_HIP(hipMemsetAsync(h->stats, 0, sizeof(double) * A * 2, st));
    if (idx_dev) {      // the caller's draw (e.g. the reference's random.sample); the gather clamps every index into [0, size)
        TSC_HIP(hipMemcpyAsync(h->idx, idx_dev, sizeof(int) * E * A * h->B, hipMemcpyDeviceToDevice, st));
    } else {
        tsc::ProfScope ps(tsc::KID_IQL_SAMPLE, st);
        if (h->B == 20)
            hipLaunchKernelGGL(iql_sample_fixed_kernel<20>, dim3((unsigned)((E * A + 127) / 128)), dim3(128), 0, st, (int)E, (int)A, size,
                               (unsigned long long)seed, (unsigned long long)update_index, h->idx);
        else
            hipLaunchKernelGGL(iql_sample_kernel, dim3((unsigned)((E * A + 127) / 128)), dim3(128), 0, st, (int)E, (int)A, h->B, size,
                               (unsigned long long)seed, (unsigned long long)update_index, h->idx);
    }
    if (h->fused) {
        QFusedArgs fa = fused_args(h, size);
        const unsigned grid = (unsigned)(A * h->fS);
        {
            tsc::ProfScope ps(tsc::KID_IQL_GRAD, st);
            if (h->fused == 10)
                hipLaunchKernelGGL((iql_fused_grad_kernel<10, 8>), dim3(grid), dim3(256), QFusedLds<10>::grad_floats * 4, st, fa);
            else
                hipLaunchKernelGGL((iql_fused_grad_kernel<8, 8>), dim3(grid), dim3(256), QFusedLds<8>::grad_floats * 4, st, fa);
        }
        TSC_HIP(hipGetLastError());
        tsc::ProfScope ps(tsc::KID_IQL_REDUCE, st);
        hipLaunchKernelGGL(iql_fused_reduce_kernel, dim3((unsigned)((L.stride + 255) / 256), (unsigned)A), dim3(256), 0, st, h->fws, h->fwsl,
                           (int)A, h->fS, L.stride, L.ob1, L.H1, h->rowrange, L.SMAX, h->grads, h->stats);
        TSC_HIP(hipGetLastError());
        return 0;
    }
    const long long tot = A * R * (L.SMAX / 4);
    hipLaunchKernelGGL(iql_gather_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, (int)E, (int)A, L.SMAX, h->B, h->cap,
                       (int)size, h->idx, h->r_obs, h->r_next, h->r_act, h->r_rew, h->r_done, h->S, h->S1, h->act, h->rew, h->done);
    TSC_HIP(hipGetLastError());
    // Q(s') first (its activations are not needed afterwards), then Q(s) with the activations the backward pass reads
    if (q_forward(h, h->S1, R * L.SMAX, L.SMAX, R, h->X1, h->X2, h->Q)) return tsc::fail("gemm launch failed");
    hipLaunchKernelGGL(iql_qmax_kernel, dim3((unsigned)((A * R + 255) / 256)), dim3(256), 0, st, h->Q, h->n_act, R, (int)A, h->q1);
    if (q_forward(h, h->S, R * L.SMAX, L.SMAX, R, h->X1, h->X2, h->Q)) return tsc::fail("gemm launch failed");
    hipLaunchKernelGGL(iql_td_kernel, dim3((unsigned)((A * R + 255) / 256)), dim3(256), 0, st, h->Q, h->q1, h->act, h->rew, h->done, R,
                       (int)A, (float)h->gamma, h->dQ, h->stats);
    TSC_HIP(hipGetLastError());
    float *g = h->grads;
    if (!L.dqn) {
        // dWq = S^T dQ, dbq = colsum(dQ); padded obs columns are zero, so their rows of dWq are exactly zero
        if (qgemm(h, true, tsc::EPI_NONE, L.SMAX, kQ, (int)R, h->S, R * L.SMAX, L.SMAX, h->dQ, R * kQ, kQ, g + L.oWq, L.stride, kQ, nullptr, 0,
                  nullptr, 0, 0, nullptr, 0, g + L.obq, L.stride)) return tsc::fail("gemm failed");
        return 0;
    }
    const long long pt = (long long)L.H1 * L.H2 > (long long)L.H2 * kQ ? (long long)L.H1 * L.H2 : (long long)L.H2 * kQ;
    hipLaunchKernelGGL(iql_transpose_kernel, dim3((unsigned)((pt * A + 255) / 256)), dim3(256), 0, st, h->params, L, h->W2T, h->WqT);
    // dWq = X2^T dQ (+ dbq)
    if (qgemm(h, true, tsc::EPI_NONE, L.H2, kQ, (int)R, h->X2, R * L.H2, L.H2, h->dQ, R * kQ, kQ, g + L.oWq, L.stride, kQ, nullptr, 0, nullptr, 0,
              0, nullptr, 0, g + L.obq, L.stride)) return tsc::fail("gemm failed");
    // dX2 = (dQ Wq^T) * (X2 > 0)
    if (qgemm(h, false, tsc::EPI_MASK_POS, (int)R, L.H2, kQ, h->dQ, R * kQ, kQ, h->WqT, (long long)L.H2 * kQ, L.H2, h->dX2, R * L.H2, L.H2,
              nullptr, 0, h->X2, R * L.H2, L.H2, nullptr, 0, nullptr, 0)) return tsc::fail("gemm failed");
    // dW2 = X1^T dX2 (+ db2)
    if (qgemm(h, true, tsc::EPI_NONE, L.H1, L.H2, (int)R, h->X1, R * L.H1, L.H1, h->dX2, R * L.H2, L.H2, g + L.oW2, L.stride, L.H2, nullptr, 0,
              nullptr, 0, 0, nullptr, 0, g + L.ob2, L.stride)) return tsc::fail("gemm failed");
    // dX1 = (dX2 W2^T) * (X1 > 0), in place over X1
    if (qgemm(h, false, tsc::EPI_MASK_POS, (int)R, L.H1, L.H2, h->dX2, R * L.H2, L.H2, h->W2T, (long long)L.H1 * L.H2, L.H1, h->X1, R * L.H1,
              L.H1, nullptr, 0, h->X1, R * L.H1, L.H1, nullptr, 0, nullptr, 0)) return tsc::fail("gemm failed");
    // dW1 = S^T dX1 masked to the block-diagonal structure (+ db1)
    if (qgemm(h, true, tsc::EPI_ROWRANGE, L.SMAX, L.H1, (int)R, h->S, R * L.SMAX, L.SMAX, h->X1, R * L.H1, L.H1, g + L.oW1, L.stride, L.H1,
              nullptr, 0, nullptr, 0, 0, h->rowrange, L.SMAX, g + L.ob1, L.stride)) return tsc::fail("gemm failed");
    return 0;
}

int tsc_iql_grad_buffer(tsc_iql *h, float **grad, int64_t *count) {
    if (!h || !grad || !count) return tsc::fail("tsc_iql_grad_buffer: bad arguments");
    *grad = h->grads; *count = h->nparam;
    return 0;
}

int tsc_iql_apply_grads(tsc_iql *h, double lr, double grad_scale, double *stats_host) {
    if (!h) return tsc::fail("null handle");
    const QLayout &L = h->lay;
    hipStream_t st = h->stream;
    tsc::ProfScope ps(tsc::KID_IQL_ADAM, st);          // norm + Adam
    hipLaunchKernelGGL(iql_norm_kernel, dim3(L.A * kNormSlices), dim3(256), 0, st, h->grads, L.stride, grad_scale, h->norm2);
    h->adam_t += 1;
    const double lr_t = lr * sqrt(1.0 - pow(0.999, (double)h->adam_t)) / (1.0 - pow(0.9, (double)h->adam_t));
    hipLaunchKernelGGL(iql_adam_kernel, dim3((unsigned)((h->nparam + 255) / 256)), dim3(256), 0, st, h->params, h->m1, h->m2, h->grads,
                       L.stride, h->nparam, h->norm2, (float)grad_scale, (float)h->max_norm, (float)lr_t);
    TSC_HIP(hipGetLastError());
    ps.stop();
    if (stats_host) {
        std::vector<double> s(L.A * 2), n2p((size_t)L.A * kNormSlices), n2(L.A, 0.0);
        TSC_HIP(hipStreamSynchronize(st));
        TSC_HIP(hipMemcpy(s.data(), h->stats, sizeof(double) * L.A * 2, hipMemcpyDeviceToHost));
        TSC_HIP(hipMemcpy(n2p.data(), h->norm2, sizeof(double) * L.A * kNormSlices, hipMemcpyDeviceToHost));
        for (int a = 0; a < L.A; ++a)
            for (int k = 0; k < kNormSlices; ++k) n2[a] += n2p[(size_t)a * kNormSlices + k];
        for (int a = 0; a < L.A; ++a) { stats_host[a * 2] = s[a * 2]; stats_host[a * 2 + 1] = sqrt(n2[a]); }
    }
    return 0;
}

int tsc_iql_debug_clock(tsc_iql *h, int32_t enable, int64_t *stamps_host, int32_t count) {
    if (!h) return tsc::fail("null handle");
    if (!h->fused) return tsc::fail("tsc_iql_debug_clock: only the fused learner carries clock stamps");
    TSC_HIP(hipStreamSynchronize(h->stream));
    const size_t n = 64 + 2 * (size_t)h->lay.A * h->fS;
    if (enable && !h->dbg) {
        TSC_HIP(hipMalloc((void **)&h->dbg, n * sizeof(long long)));
        TSC_HIP(hipMemset(h->dbg, 0, n * sizeof(long long)));
        h->allocs.push_back(h->dbg);
    }
    if (stamps_host && h->dbg)
        TSC_HIP(hipMemcpy(stamps_host, h->dbg, sizeof(long long) * ((size_t)count < n ? (size_t)count : n), hipMemcpyDeviceToHost));
    return 0;
}

int tsc_iql_path(tsc_iql *h, int32_t *fused) {
    if (!h || !fused) return tsc::fail("tsc_iql_path: bad arguments");
    *fused = h->fused ? 1 : 0;
    return 0;
}

int tsc_iql_debug_batch(tsc_iql *h, int32_t *idx_host) {
    if (!h || !idx_host) return tsc::fail("tsc_iql_debug_batch: bad arguments");
    TSC_HIP(hipStreamSynchronize(h->stream));
    TSC_HIP(hipMemcpy(idx_host, h->idx, sizeof(int) * (size_t)h->E * h->lay.A * h->B, hipMemcpyDeviceToHost));
    return 0;
}

}  // extern "C"
