// tsc_gemm.h -- grouped fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32).
//
// The reference's nets are fp32 TF graphs (agents/utils.py:66-116), one tiny graph per agent
// evaluated in a Python loop (agents/models.py:177-196).  Here all agent-towers of all env
// instances go through one launch: blockIdx.z = group (agent, tower), every group has its own
// weights.  fp32-in/fp32-accumulate MFMA is exact f32 (k-ordered fma chain), so results differ
// from a NumPy restatement only by summation order.
//
// Two forms cover every contraction of the forward and backward pass:
//   NN:  C[g][M,N] = epi(A[g][M,K] * B[g][K,N])        A row-major (lda), B row-major (ldb)
//   TN:  C[g][M,N] = epi(sum_k A[g][k,M]^T B[g][k,N])  reduction over rows (samples) of A and B
// Block tile 128x128, 4 waves (2x2), each wave 64x64 = 2x2 MFMA tiles of 32x32, BK = 16.
// Operands are staged k-major in LDS (As[k][m], Bs[k][n]) so a wave's MFMA operand fetch is one
// conflict-free ds_read_b32 per lane; the next tile is prefetched into registers while the
// current one is multiplied.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tsc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum GemmEpi : int {
    EPI_NONE = 0,
    EPI_BIAS = 1,        // + bias[n]
    EPI_BIAS_RELU = 2,   // relu(. + bias[n])
    EPI_MASK_POS = 3,    // . * (aux[m,n] > 0)          (relu backward, aux = forward activation)
    EPI_ROWRANGE = 4,    // keep only n in [lo[m], hi[m])  (structural zeros of the block-diagonal FC)
};

struct GemmArgs {
    const float *A, *B;
    float *C;
    const float *bias;      // [G][N]            (EPI_BIAS*)
    const float *aux;       // [G][M][ldaux]     (EPI_MASK_POS)
    const int16_t *rr;      // [G][M][2]         (EPI_ROWRANGE)
    float *colsum;          // TN only: [G][N] column sums of B over k (bias gradients), or null
    long long sA, sB, sC, sBias, sAux, sRR, sColsum;   // group strides (elements)
    int lda, ldb, ldc, ldaux;
    int M, N, K;
    int gdivA;              // A (and rr) belong to group g / gdivA (both towers of an agent read the same obs)
    // split-K (TN only): the reduction range is cut into `splitk` chunks of kchunk rows; chunk s writes its raw
    // partial tile to ws[s][g][M][N] (and partial column sums to wsc[s][g][N]); splitk_reduce_kernel folds them
    // in fixed order (deterministic) and applies the epilogue.
    int splitk, kchunk;
    float *ws, *wsc;
    int groups;
};

constexpr int BM = 128, BN = 128, BK = 16, LDS_PAD = 4;

template <bool TN, int EPI>
__global__ void __launch_bounds__(256, 2) gemm_grouped_kernel(GemmArgs p) {
    __shared__ float As[2][BK][BM + LDS_PAD];
    __shared__ float Bs[2][BK][BN + LDS_PAD];
    const int g = blockIdx.z;
    const int split = TN ? (int)blockIdx.x % p.splitk : 0;
    const int m0 = (TN ? (int)blockIdx.x / p.splitk : (int)blockIdx.x) * BM, n0 = blockIdx.y * BN;
    const float *A = p.A + (long long)(g / p.gdivA) * p.sA;
    const float *B = p.B + (long long)g * p.sB;
    float *C = p.C + (long long)g * p.sC;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave & 1) * 64, wn = (wave >> 1) * 64;
    const int M = p.M, N = p.N;
    const int kbeg = TN ? split * p.kchunk : 0;
    const int K = TN ? min(p.K, kbeg + p.kchunk) : p.K;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // which 32-wide sub-tiles are inside the problem (skip dead MFMAs for ragged N / M)
    const bool live_m0 = m0 + wm < M, live_m1 = m0 + wm + 32 < M;
    const bool live_n0 = n0 + wn < N, live_n1 = n0 + wn + 32 < N;

    float4 ra[2], rb[2];
    float csum = 0.0f;

    auto load_tiles = [&](int k0) {
        // ---- A tile
        if (TN) {   // A[k][m]: 16 x 128 floats = 512 float4, thread -> (k = q / 32, m4 = q % 32)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int q = tid + 256 * j, k = q >> 5, m = (q & 31) * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k0 + k < K) {
                    const float *src = A + (long long)(k0 + k) * p.lda + m0 + m;
                    if (m0 + m + 3 < M) v = *reinterpret_cast<const float4 *>(src);
                    else {
                        if (m0 + m < M) v.x = src[0];
                        if (m0 + m + 1 < M) v.y = src[1];
                        if (m0 + m + 2 < M) v.z = src[2];
                    }
                }
                ra[j] = v;
            }
        } else {    // A[m][k]: 128 x 16 floats, thread -> (m = q / 4, k4 = q % 4)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int q = tid + 256 * j, m = q >> 2, k = (q & 3) * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (m0 + m < M) {
                    const float *src = A + (long long)(m0 + m) * p.lda + k0 + k;
                    if (k0 + k + 3 < K) v = *reinterpret_cast<const float4 *>(src);
                    else {
                        if (k0 + k < K) v.x = src[0];
                        if (k0 + k + 1 < K) v.y = src[1];
                        if (k0 + k + 2 < K) v.z = src[2];
                    }
                }
                ra[j] = v;
            }
        }
        // ---- B tile  B[k][n]
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = tid + 256 * j, k = q >> 5, n = (q & 31) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k0 + k < K) {
                const float *src = B + (long long)(k0 + k) * p.ldb + n0 + n;
                if (n0 + n + 3 < N) v = *reinterpret_cast<const float4 *>(src);
                else {
                    if (n0 + n < N) v.x = src[0];
                    if (n0 + n + 1 < N) v.y = src[1];
                    if (n0 + n + 2 < N) v.z = src[2];
                }
            }
            rb[j] = v;
        }
    };
    auto store_tiles = [&](int buf) {
        if (TN) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int q = tid + 256 * j, k = q >> 5, m = (q & 31) * 4;
                *reinterpret_cast<float4 *>(&As[buf][k][m]) = ra[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int q = tid + 256 * j, m = q >> 2, k = (q & 3) * 4;
                As[buf][k + 0][m] = ra[j].x; As[buf][k + 1][m] = ra[j].y;
                As[buf][k + 2][m] = ra[j].z; As[buf][k + 3][m] = ra[j].w;
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = tid + 256 * j, k = q >> 5, n = (q & 31) * 4;
            *reinterpret_cast<float4 *>(&Bs[buf][k][n]) = rb[j];
        }
    };

    const int nk = (K - kbeg + BK - 1) / BK;
    load_tiles(kbeg);
    store_tiles(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tiles(kbeg + (kt + 1) * BK);
        const int kh = lane >> 5, li = lane & 31;
        // One liveness test per K tile, not per MFMA (round 5): with the test around every MFMA each sat in a basic block of its own and
        // the next step's LDS operands could not be requested under it.  The whole-wave cases -- all four 32 x 32 sub-tiles live, or one
        // column of them (N <= 32 beyond the wave's first column: the skinny heads) -- run straight-line; ragged tiles keep the tests.
        if (live_m1 && live_n1) {
#pragma unroll
            for (int kk = 0; kk < BK; kk += 2) {
                const float a0 = As[buf][kk + kh][wm + li], a1 = As[buf][kk + kh][wm + 32 + li];
                const float b0 = Bs[buf][kk + kh][wn + li], b1 = Bs[buf][kk + kh][wn + 32 + li];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
        } else if (live_m1 && live_n0) {                    // (live_n1 false: one live column of sub-tiles)
#pragma unroll
            for (int kk = 0; kk < BK; kk += 2) {
                const float a0 = As[buf][kk + kh][wm + li], a1 = As[buf][kk + kh][wm + 32 + li];
                const float b0 = Bs[buf][kk + kh][wn + li];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < BK; kk += 2) {
                const float a0 = As[buf][kk + kh][wm + li], a1 = As[buf][kk + kh][wm + 32 + li];
                const float b0 = Bs[buf][kk + kh][wn + li], b1 = Bs[buf][kk + kh][wn + 32 + li];
                if (live_m0 && live_n0) acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                if (live_m0 && live_n1) acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                if (live_m1 && live_n0) acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                if (live_m1 && live_n1) acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
        }
        if (TN && p.colsum && m0 == 0 && tid < BN) {
#pragma unroll
            for (int kk = 0; kk < BK; ++kk) csum += Bs[buf][kk][tid];
        }
        if (kt + 1 < nk) {
            store_tiles(buf ^ 1);
        }
        __syncthreads();
    }

    const bool partial = TN && p.splitk > 1;
    if (TN && p.colsum && m0 == 0 && tid < BN && n0 + tid < N) {
        if (partial) p.wsc[((long long)split * p.groups + g) * N + n0 + tid] = csum;
        else p.colsum[(long long)g * p.sColsum + n0 + tid] = csum;
    }
    if (partial) {      // raw partial tile -> workspace, epilogue happens in the reduce kernel
        float *W = p.ws + ((long long)split * p.groups + g) * M * N;
        const int li2 = lane & 31, kh2 = lane >> 5;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = n0 + wn + 32 * j + li2;
                if (n >= N) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * kh2;
                    if (m < M) W[(long long)m * N + n] = acc[i][j][r];
                }
            }
        return;
    }

    // ---- epilogue: C/D layout of 32x32x2: row = (r&3) + 8*(r>>2) + 4*(lane>>5), col = lane&31
    const int li = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn + 32 * j + li;
            if (n >= N) continue;
            float bias = 0.0f;
            if (EPI == EPI_BIAS || EPI == EPI_BIAS_RELU) bias = p.bias[(long long)g * p.sBias + n];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (m >= M) continue;
                float v = acc[i][j][r];
                if (EPI == EPI_BIAS) v = v + bias;
                if (EPI == EPI_BIAS_RELU) { v = v + bias; v = v > 0.0f ? v : 0.0f; }
                if (EPI == EPI_MASK_POS) {
                    const float a = p.aux[(long long)g * p.sAux + (long long)m * p.ldaux + n];
                    v = a > 0.0f ? v : 0.0f;
                }
                if (EPI == EPI_ROWRANGE) {
                    const int16_t *rr = p.rr + ((long long)(g / p.gdivA) * p.sRR + m) * 2;
                    if (n < rr[0] || n >= rr[1]) v = 0.0f;
                }
                C[(long long)m * p.ldc + n] = v;
            }
        }
}

template <int EPI>
__global__ void splitk_reduce_kernel(GemmArgs p) {
    const long long per = (long long)p.M * p.N;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int g = blockIdx.y;
    if (idx < per) {
        const int m = (int)(idx / p.N), n = (int)(idx % p.N);
        float s = 0.0f;
        for (int k = 0; k < p.splitk; ++k) s += p.ws[((long long)k * p.groups + g) * per + idx];
        if (EPI == EPI_ROWRANGE) {
            const int16_t *rr = p.rr + ((long long)(g / p.gdivA) * p.sRR + m) * 2;
            if (n < rr[0] || n >= rr[1]) s = 0.0f;
        }
        p.C[(long long)g * p.sC + (long long)m * p.ldc + n] = s;
    }
    if (p.colsum && idx < p.N) {
        float s = 0.0f;
        for (int k = 0; k < p.splitk; ++k) s += p.wsc[((long long)k * p.groups + g) * p.N + idx];
        p.colsum[(long long)g * p.sColsum + idx] = s;
    }
}

template <bool TN, int EPI>
inline void launch_gemm(const GemmArgs &a, int groups, hipStream_t st) {
    const int sk = TN ? a.splitk : 1;
    dim3 grid(((a.M + BM - 1) / BM) * sk, (a.N + BN - 1) / BN, groups);
    hipLaunchKernelGGL((gemm_grouped_kernel<TN, EPI>), grid, dim3(256), 0, st, a);
    if (TN && sk > 1) {
        const long long per = (long long)a.M * a.N;
        dim3 rg((unsigned)((per + 255) / 256), groups);
        hipLaunchKernelGGL((splitk_reduce_kernel<EPI>), rg, dim3(256), 0, st, a);
    }
}

// choose a split so that the launch has ~2k workgroups and every chunk keeps >= 1024 rows
inline void plan_splitk(GemmArgs &a, int groups, float *ws, float *wsc, size_t ws_floats, size_t wsc_floats) {
    a.groups = groups; a.ws = ws; a.wsc = wsc; a.splitk = 1; a.kchunk = a.K;
    if (!ws) return;
    const long long tiles = (long long)((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN) * groups;
    long long s = 2048 / (tiles > 0 ? tiles : 1);
    if (s > a.K / 1024) s = a.K / 1024;
    while (s > 1 && ((size_t)s * groups * a.M * a.N > ws_floats || (size_t)s * groups * a.N > wsc_floats)) --s;
    if (s <= 1) return;
    int kc = (int)((a.K + s - 1) / s);
    kc = (kc + BK - 1) / BK * BK;
    a.kchunk = kc;
    a.splitk = (a.K + kc - 1) / kc;
}

inline void launch_gemm_dyn(bool tn, int epi, const GemmArgs &a, int groups, hipStream_t st) {
    if (!tn) {
        switch (epi) {
            case EPI_NONE: launch_gemm<false, EPI_NONE>(a, groups, st); break;
            case EPI_BIAS: launch_gemm<false, EPI_BIAS>(a, groups, st); break;
            case EPI_BIAS_RELU: launch_gemm<false, EPI_BIAS_RELU>(a, groups, st); break;
            case EPI_MASK_POS: launch_gemm<false, EPI_MASK_POS>(a, groups, st); break;
            default: launch_gemm<false, EPI_ROWRANGE>(a, groups, st); break;
        }
    } else {
        switch (epi) {
            case EPI_NONE: launch_gemm<true, EPI_NONE>(a, groups, st); break;
            case EPI_ROWRANGE: launch_gemm<true, EPI_ROWRANGE>(a, groups, st); break;
            default: launch_gemm<true, EPI_NONE>(a, groups, st); break;
        }
    }
}

}  // namespace tsc
