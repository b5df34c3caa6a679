// tsc_env.hip -- batched traffic microsimulator + reference env semantics on gfx950.
//
// Replaces, for E parallel env instances, the reference's TrafficSimulator.step/reset
// (envs/env.py:544-631) *including* the SUMO process behind its TraCI socket:
//   K1 signal FSM ............ envs/env.py:128-152,455-459 (tables from scenario.py)
//   K2 vehicle update ........ MICROSIM_SPEC.md (IDM + safe-speed clamp, lane queues)
//   K3 hand-off / insertion .. MICROSIM_SPEC.md (feeder gather, vehsPerHour flows)
//   K4 detectors ............. envs/env.py:325-407 (wave / halting / head-vehicle wait)
//   K5 observation gather .... envs/env.py:163-205,439-442 (float64 arithmetic, cast to f32)
//   K6 reward + shaping ...... envs/env.py:356-367,580,590-631 (float64, numpy sum order)
//
// Mapping: one workgroup per env instance: one thread per lane that a route can ever put a vehicle on (a
// prefix after load sorting, padded to a multiple of 64 = NLA) plus helper threads up to 256.  Vehicles live
// in HBM as one 16-byte record per slot {x, v, desired-speed factor, bits of (wait | route << 16)}, lane-major: S[E][NLP][CAP]
// (vslot) -- a lane's queue is contiguous, front first, so the flat phase (consecutive threads own consecutive queued vehicles
// of a lane; it is most of the kernel) reads and writes runs of up to 27 x 16 bytes with one load / one store per vehicle.
// Until round 5 the state was four slot-major arrays X / V / SF / M [CAP][NLP], coalesced for the lane threads' head walk
// instead: every flat-phase access instruction touched 64 cache lines (92.7 -> 86.3 us per control step at E = 1024 over
// whole episodes with lane-major arrays, -> 81.1 us with the records; PMC traffic per launch 159 -> 57 MB).
// Everything lanes need from *other* lanes
// (tail/head summaries, signal states, the per-step hand-off outbox, the route tables) is staged in LDS; one
// control step (2 yellow + 3 green simulated seconds, detectors, obs, reward) is a single launch.  Per
// simulated second: phase H (lane threads: the platoon that crosses and the first vehicle that stays), phase F (all
// threads: every other queued vehicle in a flat, load-balanced order -- car-following from the old state, the queue
// constraint as a segmented prefix-min scan on the wavefront's DPP path; its first half runs on the helper wavefronts
// WHILE the lane wavefronts are in phase H), phase B (gather + demand) (see step_kernel).  The reference's large_grid
// and Monaco run instantiations whose table dimensions are compile-time constants (kSpec), with 256, 512 or 1024
// threads per instance depending on how many instances share a CU (tsc_env_create).
//
// Arithmetic is fp32 with one rounding per operation (-ffp-contract=off, IEEE div/sqrt) so the
// vehicle state is bit-identical to the CPU oracle; obs/reward are computed in float64 exactly
// like the reference's NumPy code.
#include "tsc_common.h"
#include "../../include/tsc.h"

#include <cmath>
#include <cstdlib>
#include <vector>

namespace {

constexpr float kLen = 5.0f, kS0 = 2.0f /* standstill gap (SUMO minGap 2.5 less the storage of its junction interiors, MICROSIM_SPEC.md) */,
                kAcc = 5.0f, kDec = 10.0f, kTHead = 1.0f;   // headway = SUMO's default tau
constexpr float kICab = 0.070710678f /* 1 / (2 sqrt(acc dec)): a multiply instead of an IEEE division */, kHalt = 0.1f, kYieldT = 3.0f, kYieldD = 10.0f;
constexpr int kCap = TSC_LANE_CAP, kMaxCross = TSC_MAX_CROSS, kMaxUp = TSC_MAX_UP;
// Element index of slot i of lane q in an instance's vehicle arrays X / V / SF / M (and R0 / R1).
#ifndef TSC_LANE_MAJOR
#define TSC_LANE_MAJOR 1
#endif
__host__ __device__ __forceinline__ int vslot(int i, int q, int NLP) { return TSC_LANE_MAJOR ? q * kCap + i : i * NLP + q; }

constexpr int kMaxEntry = 8;           // routes that may share one entry lane (small_grid: 6 paths leave np1_nt1)
// movement word of (lane, route):  [11:0] next lane (0xFFF = route ends here, 0xFFE = n/a)
//   [17:12] signal link (63 = none)   [29:18] lane this movement yields to (0xFFF = none)   [30] priority movement
__host__ __device__ __forceinline__ int mv_tl(int w) { const int t = w & 0xFFF; return t == 0xFFF ? -1 : t == 0xFFE ? -2 : t; }
__host__ __device__ __forceinline__ int mv_k(int w) { const int k = (w >> 12) & 0x3F; return k == 63 ? -1 : k; }
__host__ __device__ __forceinline__ int mv_yield(int w) { const int y = (w >> 18) & 0xFFF; return y == 0xFFF ? -1 : y; }
__host__ __device__ __forceinline__ bool mv_prio(int w) { return (w >> 30) & 1; }

struct EnvDev {
    int NL, NLP, NR, A, NF, KMAX, PMAX, LMAX, SMAX, NBR, E;
    int NS, KC;                    // insertion streams (= NR when every route is its own stream), route choices per stream
    int NU, NLA;                   // lanes that can ever hold a vehicle (prefix after load sorting), rounded up to wavefronts
    int help;                      // phase A1 enabled (step_kernel)
    const float *lane_len, *lane_vmax, *lane_det;
    const int *lane_node, *lane_up;
    const int *lane_sib;           // [NL] sibling lane of a two-lane street or -1; null = no lane changing (rule 10)
    const int *mv;                 // [NL*NR] packed movement word, see mv_* helpers
    const uint8_t *zip;            // [NL*NR] zipper-merge slot: rank | count << 4 (0 = none)
    const int *route_entry;        // [NS] entry lane of every stream
    const int *flow_ptr;           // [NS+1] CSR over flows sorted by stream
    const int *sroute;             // [NS] route of a stream's vehicles (-1: drawn); null = the stream index itself
    const int *smode;              // [NS] 0 fixed, 1 per-vehicle draw from schoice, 2 per-instance route (iroute)
    const int *schoice;            // [NS][KC][2] (route, cumulative weight of 65536)
    const float *sorigin;          // [NS][2] the stretch of the entry lane vehicles are inserted on, or null (whole lane)
    int NI, ilen;                  // choice intervals per stream and their length in seconds
    int *iroute;                   // [E][NS] routes of the mode-2 streams of the running episode
    const int *flows;              // [NF*4] begin,end,vph,route (sorted by route, stable)
    const uint32_t *lane_routes;   // [NL][2] the (<= kMaxEntry) routes whose entry lane this is, one byte each, 0xFF = none
    const uint8_t *emit_tab;       // [NS][emit_len] vehicles each stream's flows emit at second t
    int emit_len;
    const int *agent_lanes, *agent_nlane, *agent_nlink, *agent_nphase;
    const uint8_t *green_tab, *yellow_tab;
    const int *nbr, *obs_kind, *obs_src;
    const int *obs_ks;             // [A*SMAX] obs_kind << 16 | obs_src (one load per observation entry)
    int ctrl, yellow, episode, teleport, queue_cap, objective, agent_kind, realnet_scale;
    double coop_gamma, norm_wave, norm_wait, clip_wave, clip_wait, coef_wait;
    float4 *S;                     // [E][NLP][CAP] vehicle records {x, v, desired-speed factor, bits of (w | route << 16)}
    int *N;                        // [E][NLP]
    int *pending, *serial;         // [E][NS]
    int *tsec;
    uint32_t *seed;
    int *prev_action;              // [E][A]
    float *fp;                     // [E][A][PMAX]
    unsigned long long *arrived;   // [E] vehicles that reached the end of their route this episode
    unsigned long long *teleported; // [E] heads the teleport surrogate took out of the network (truncated trips, NOT arrivals)
    double *reward_acc;            // [E] running sum of the global reward (training-curve logging, utils.py:161,296-305)
    const float *fp_bound;         // zero-copy fingerprint source (tsc_env_bind_fingerprint) or null
    long long *dbg;                // optional: shader-clock stamps of workgroup 0 / thread 0 (tsc_env_debug_clock)
    const int *order;              // [E] instance of workgroup b (tsc_env_set_block_order / the load balancer); null = b
    // ---- evaluation recording (envs/env.py:409-437,498-515; tsc_env_record): off on the training path
    int rec;                       // per-second network statistics + trip log are being kept
    int trip_cap;                  // trip records per instance
    const float *lane_origin;      // [NL] start of the SUMO lane inside the compiled lane (lane.* getters count from here)
    uint32_t *R0, *R1;             // per vehicle, indexed like X: depart_sec | serial << 16 ; waiting seconds | waiting count << 16
    long long *rec_int;            // [E][8 sec][4]: vehicles, departed, arrived, sum of waiting times
    double *rec_speed;             // [E][8 sec]: sum of speeds (per-lane partial sums added in lane order)
    int *rec_queue;                // [E][8 sec][A * LMAX]: halting vehicles on every incoming lane (whole SUMO lane)
    int *trips;                    // [E][trip_cap][6]: route, serial, depart, arrival, waiting seconds, waiting count
    int *n_trips;                  // [E]
    unsigned long long *live_acc;  // [E] sum over control steps of the vehicles in the network (window-mean V)
    // ---- greedy controllers (tsc_env_set_greedy / tsc_env_greedy_actions): candidate flows per agent
    int GC, GT;                    // candidates per agent (padded), terms per candidate (padded)
    const int *g_ncand;            // [A]
    const int8_t *g_term;          // [A][GC][GT] observation index of a term, -1 = end
    const int *g_action;           // [A][GC] action a winning candidate stands for
};

__device__ __forceinline__ uint32_t hash32(uint32_t seed, uint32_t route, uint32_t serial, uint32_t stream) {
    uint32_t h = seed * 0x9E3779B1u + route * 0x85EBCA77u + serial * 0xC2B2AE3Du + stream * 0x27D4EB2Fu;
    h ^= h >> 16; h *= 0x85EBCA6Bu;
    h ^= h >> 13; h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}
__device__ __forceinline__ float u01(uint32_t h) { return (float)(h >> 8) * (1.0f / 16777216.0f); }

// Wavefront scans on the DPP data path (gfx9: row_shr within the four 16-lane rows, then row_bcast:15 into rows 1 / 3 and
// row_bcast:31 into rows 2 / 3): a cross-lane step is one VALU move instead of a ds_bpermute_b32 with its address arithmetic
// (~5 instructions and an LDS round trip each; the flat phase ran 14 of them per super-round).  Lanes without a source lane
// keep `ident`, the identity of the operator, so no lane test is needed.
#define TSC_DPP_I(ident, v, ctrl, rows) __builtin_amdgcn_update_dpp((ident), (v), (ctrl), (rows), 0xF, false)
#define TSC_DPP_F(ident, v, ctrl, rows) __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(ident), __float_as_int(v), (ctrl), (rows), 0xF, false))
constexpr int kDppShr1 = 0x111, kDppShr2 = 0x112, kDppShr4 = 0x114, kDppShr8 = 0x118, kDppBcast15 = 0x142, kDppBcast31 = 0x143,
              kDppWaveShr1 = 0x138;
// inclusive prefix sum over the wavefront
__device__ __forceinline__ int wave_scan_add(int v) {
    v += TSC_DPP_I(0, v, kDppShr1, 0xF); v += TSC_DPP_I(0, v, kDppShr2, 0xF);
    v += TSC_DPP_I(0, v, kDppShr4, 0xF); v += TSC_DPP_I(0, v, kDppShr8, 0xF);
    v += TSC_DPP_I(0, v, kDppBcast15, 0xA); v += TSC_DPP_I(0, v, kDppBcast31, 0xC);
    return v;
}
// inclusive prefix maximum over the wavefront (values >= 0)
__device__ __forceinline__ int wave_scan_max(int v) {
    v = max(v, TSC_DPP_I(0, v, kDppShr1, 0xF)); v = max(v, TSC_DPP_I(0, v, kDppShr2, 0xF));
    v = max(v, TSC_DPP_I(0, v, kDppShr4, 0xF)); v = max(v, TSC_DPP_I(0, v, kDppShr8, 0xF));
    v = max(v, TSC_DPP_I(0, v, kDppBcast15, 0xA)); v = max(v, TSC_DPP_I(0, v, kDppBcast31, 0xC));
    return v;
}
// segmented inclusive min-scan: (v, f) <- (f ? v : min(v, v_prev), f | f_prev); f = 1 marks the first lane of a segment
__device__ __forceinline__ void wave_seg_scan_min(float &v, int &f) {
#define TSC_SEG_STEP(ctrl, rows) do {                                     \
        const float ov = TSC_DPP_F(INFINITY, v, ctrl, rows);              \
        const int of = TSC_DPP_I(0, f, ctrl, rows);                       \
        v = f ? v : fminf(v, ov); f |= of; } while (0)
    TSC_SEG_STEP(kDppShr1, 0xF); TSC_SEG_STEP(kDppShr2, 0xF); TSC_SEG_STEP(kDppShr4, 0xF); TSC_SEG_STEP(kDppShr8, 0xF);
    TSC_SEG_STEP(kDppBcast15, 0xA); TSC_SEG_STEP(kDppBcast31, 0xC);
#undef TSC_SEG_STEP
}

// one car-following evaluation against one leader (MICROSIM_SPEC.md rule 3)
__device__ __forceinline__ float follow(float v, float v0, bool has_lead, float g, float vl, float s0gap) {
    float ratio = v / v0;
    float r2 = ratio * ratio;
    float acc = kAcc * (1.0f - r2 * r2);
    float vsafe = INFINITY;
    if (has_lead) {
        float sstar = (kS0 + v * kTHead) + (v * (v - vl)) * kICab;
        if (sstar < kS0) sstar = kS0;
        float s = g < 0.5f ? 0.5f : g;
        float q = sstar / s;
        acc = acc - kAcc * (q * q);
        float gs = g - s0gap;
        if (gs < 0.0f) gs = 0.0f;
        vsafe = sqrtf((kDec * kDec + vl * vl) + (2.0f * kDec) * gs) - kDec;
    }
    if (acc < -kDec) acc = -kDec;
    float vn = v + acc;
    if (vn > vsafe) vn = vsafe;
    if (vn > v0 && v <= v0) vn = v0;
    if (vn < 0.0f) vn = 0.0f;
    return vn;
}

__device__ __forceinline__ bool sig_open(int tl, int k, int a, int w, float x, float v, float L,
                                         const uint8_t *link, int KMAX, int teleport) {
    if (tl == -1) return true;                      // route ends at the lane end
    if (tl < -1) return false;                      // route n/a on this lane: hold
    if (a < 0) return true;                         // uncontrolled junction: always open
    if (k < 0) return false;
    if (w >= teleport) return true;
    uint8_t st = link[a * KMAX + k];
    if (st == 'G' || st == 'g') return true;
    if (st == 'y') {
        float need = (v * v) / (2.0f * kDec);
        return (L - x) < need;
    }
    return false;
}

// Table dimensions of the reference's scenarios as compile-time constants (SPEC of step_kernel; matched against the
// scenario at create time): 1 = large_grid (5x5), 2 = real_net (Monaco, lane chains contracted)
struct SpecDims { int NLP, NLA, NU, NR, A, KMAX, PMAX, LMAX, NBR, ctrl, yellow, teleport; };
constexpr SpecDims kSpec[3] = {{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
                               {192, 128, 88, 12, 25, 12, 5, 6, 4, 5, 2, 600},     // NU: 81 lanes carry vehicles (83 with lane changing); the rest stay empty
                               {192, 128, 113, 16, 28, 22, 6, 11, 5, 5, 2, 300}};

struct Smem {
    int *mv;                                   // [NU*NR]
    int *n; float *tx, *tv, *hx, *hv; uint32_t *hm;   // lane summaries [NLA]
    float *ox, *ov, *osf; uint32_t *om; int *oto;     // outbox [kMaxCross*NLA]
    int *nout;                                  // [NLA]
    int *wave, *halt, *hwait;                   // detectors [NLP]
    double *r;                                  // local rewards [A] (+1 for global)
    uint8_t *link_y, *link_g;                   // [A*KMAX]
    float *len, *vmax; int *node;               // lane length / speed limit / downstream agent [NLA]
    int *sib;                                   // sibling lane (rule 10) or -1 [NLA]
    int *pend, *ser; uint8_t *emit;             // per-stream insertion state [NS], emissions [NS*8]
    uint8_t *zip;                               // [NU*NR]
    uint32_t *up4;                              // [NLA] the lane's feeders, a byte each (0xFF = none): merge arbitration
    int *pre, *wtot;                            // wave-local inclusive scan of queued vehicles [NLA] (from phase H on: the lane's
                                                // first flat index), wave totals [16]
    uint8_t *mark; uint16_t *bound;             // flat phase: mark[k] = (lane & 63) + 1 where a lane's queued vehicles start at flat index
                                                // k (else 0) [NLA * (kCap - 1)]; bound[j] = lane + 1 of the owner of flat index 64 j
    int *nc; float *seed;                       // per lane: vehicles ahead of the first one that stays; its chain key [NLA]
    float *wtail, *hz;                          // wave tails of the chain scan [2][16]; old (x, v) across super-rounds [2]
    uint32_t *or0, *or1;                        // outbox of the trip records [kMaxCross*NLA]      (recording only)
    int *rq; double *rsp; long long *rint;      // per-lane halting [NLA], speed partial sums [NLA], counters [4]  (recording only)
};

// One layout for the step and the reset kernel.
template <class Take>
__host__ __device__ __forceinline__ void smem_layout(Smem &s, const EnvDev &P, Take take) {
    s.r = (double *)take(sizeof(double) * (P.A + 1));
    s.mv = (int *)take(sizeof(int) * P.NU * P.NR);
    s.n = (int *)take(4 * P.NLA); s.tx = (float *)take(4 * P.NLA); s.tv = (float *)take(4 * P.NLA);
    s.hx = (float *)take(4 * P.NLA); s.hv = (float *)take(4 * P.NLA); s.hm = (uint32_t *)take(4 * P.NLA);
    s.ox = (float *)take(4 * kMaxCross * P.NLA); s.ov = (float *)take(4 * kMaxCross * P.NLA);
    s.osf = (float *)take(4 * kMaxCross * P.NLA); s.om = (uint32_t *)take(4 * kMaxCross * P.NLA);
    s.oto = (int *)take(4 * kMaxCross * P.NLA);
    s.nout = (int *)take(4 * P.NLA);
    s.wave = (int *)take(4 * P.NLP); s.halt = (int *)take(4 * P.NLP); s.hwait = (int *)take(4 * P.NLP);
    s.link_y = (uint8_t *)take(P.A * P.KMAX); s.link_g = (uint8_t *)take(P.A * P.KMAX);
    s.len = (float *)take(4 * P.NLA); s.vmax = (float *)take(4 * P.NLA); s.node = (int *)take(4 * P.NLA);
    s.sib = (int *)take(4 * P.NLA);
    s.pend = (int *)take(4 * P.NS); s.ser = (int *)take(4 * P.NS); s.emit = (uint8_t *)take(8 * P.NS);
    s.zip = (uint8_t *)take((P.NU * P.NR + 3) / 4 * 4);
    s.up4 = (uint32_t *)take(4 * P.NLA);
    s.pre = (int *)take(4 * P.NLA); s.wtot = (int *)take(4 * 16);
    s.mark = (uint8_t *)take((size_t)P.NLA * (kCap - 1) + 8); s.bound = (uint16_t *)take(2 * ((size_t)P.NLA * (kCap - 1) / 64 + 8));
    s.nc = (int *)take(4 * P.NLA); s.seed = (float *)take(4 * P.NLA); s.wtail = (float *)take(4 * 32); s.hz = (float *)take(4 * 4);
    if (P.rec) {
        s.or0 = (uint32_t *)take(4 * kMaxCross * P.NLA); s.or1 = (uint32_t *)take(4 * kMaxCross * P.NLA);
        s.rq = (int *)take(4 * P.NLA); s.rsp = (double *)take(8 * P.NLA); s.rint = (long long *)take(8 * 4);
    } else {
        s.or0 = s.or1 = nullptr; s.rq = nullptr; s.rsp = nullptr; s.rint = nullptr;
    }
}

__device__ __forceinline__ Smem carve(char *base, const EnvDev &P) {
    Smem s;
    char *p = base;
    smem_layout(s, P, [&](size_t bytes) { char *q = p; p += (bytes + 15) & ~size_t(15); return q; });
    return s;
}

size_t smem_bytes(const EnvDev &P) {
    Smem s;
    size_t tot = 0;
    smem_layout(s, P, [&](size_t bytes) { tot += (bytes + 15) & ~size_t(15); return (char *)nullptr; });
    return tot;
}

// numpy's float64 pairwise sum for n <= 128 (np.sum at envs/env.py:580)
__device__ double np_sum(const double *a, int n) {
    if (n < 8) {
        double r = 0.0;
        for (int i = 0; i < n; ++i) r += a[i];
        return r;
    }
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int i = 8;
    for (; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
}

__device__ __forceinline__ double norm_clip(double x, double norm, double clip) {
    x = x / norm;                                   // envs/env.py:439-442
    if (clip < 0) return x;
    return fmin(fmax(x, 0.0), clip);
}

// K5: float32(state) for every agent of env e (envs/env.py:163-205)
__device__ void emit_obs(const EnvDev &P, const Smem &s, int e, float *obs) {
    const int tot = P.A * P.SMAX;
#pragma unroll 4
    for (int idx = threadIdx.x; idx < tot; idx += blockDim.x) {
        int kind = P.obs_kind[idx], src = P.obs_src[idx];
        float o = 0.0f;
        if (kind == 1) o = (float)norm_clip((double)s.wave[src], P.norm_wave, P.clip_wave);
        else if (kind == 2) o = (float)(norm_clip((double)s.wave[src], P.norm_wave, P.clip_wave) * P.coop_gamma);
        else if (kind == 3) o = (float)norm_clip((double)s.hwait[src], P.norm_wait, P.clip_wait);
        else if (kind == 4) o = (P.fp_bound ? P.fp_bound : P.fp)[(size_t)e * P.A * P.PMAX + src];
        obs[(size_t)e * tot + idx] = o;
    }
}

__global__ void reset_kernel(EnvDev P, const uint32_t *seeds, float *obs) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    Smem s = carve(smem_raw, P);
    const int e = blockIdx.x, l = threadIdx.x;
    if (l < P.NLP) { P.N[(size_t)e * P.NLP + l] = 0; s.wave[l] = 0; s.hwait[l] = 0; }
    for (int r = l; r < P.NS; r += blockDim.x) {
        P.pending[(size_t)e * P.NS + r] = 0;
        P.serial[(size_t)e * P.NS + r] = 0;
    }
    for (int a = l; a < P.A; a += blockDim.x) {
        P.prev_action[(size_t)e * P.A + a] = 0;                    // envs/env.py:448
        int na = P.agent_nphase[a];
        float p = (float)(1.0 / (double)na);                       // envs/env.py:263-269
        for (int k = 0; k < P.PMAX; ++k) P.fp[((size_t)e * P.A + a) * P.PMAX + k] = k < na - 1 ? p : 0.0f;
    }
    if (l == 0) { P.tsec[e] = 0; P.seed[e] = seeds[e]; P.arrived[e] = 0ull; P.teleported[e] = 0ull; P.n_trips[e] = 0; }
    __syncthreads();
    emit_obs(P, s, e, obs);
}

__global__ void fingerprint_kernel(EnvDev P, const float *pi) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t tot = (size_t)P.E * P.A * P.PMAX;
    if (i < tot) P.fp[i] = pi[i];                                  // pi[:-1] is applied at gather time
}

// The reference's greedy controllers (LargeGridController envs/large_grid_env.py:45-60, RealNetController
// envs/real_net_env.py:78-111, SmallGridController envs/small_grid_env.py:40-55) are one rule over different tables: every
// candidate sums some of the agent's OWN wave entries in a fixed order, float64, starting from 0 -- `ob[0] + ob[3]`, `wave
// += ob[j]` per green link in link order, `ob[k]` alone -- and np.argmax keeps the FIRST maximum; the winner stands for an
// action (the phase itself, or small_grid's STATE_PHASE_MAP entry).  One thread per (instance, agent).
// The controllers see the env's float64 state, this kernel the float32 observation: a wave entry is a vehicle count over
// norm_wave, clipped (envs/env.py:439-442), so the count is recovered (rint(ob * norm_wave); exact for counts < 2^20) and
// the float64 value recomputed with the reference's operations -- ties between candidates then fall as they do in the
// reference (0.2 + 0.4 > 0.6 in float64, a tie on the float32 values).
__global__ void __launch_bounds__(256) greedy_kernel(EnvDev P, const float *__restrict__ obs, int *__restrict__ action) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.E * P.A) return;
    const int a = i % P.A;
    const float *ob = obs + (size_t)i * P.SMAX;
    const int8_t *term = P.g_term + (size_t)a * P.GC * P.GT;
    const float clipf = (float)P.clip_wave;
    double best = 0.0;
    int arg = 0;
    const int nc = P.g_ncand[a];
    for (int c = 0; c < nc; ++c) {
        double flow = 0.0;
        for (int k = 0; k < P.GT; ++k) {
            const int j = term[c * P.GT + k];
            if (j < 0) break;
            const float o = ob[j];
            double x = rint((double)o * P.norm_wave) / P.norm_wave;
            if (P.clip_wave >= 0.0 && (x > P.clip_wave || o == clipf)) x = P.clip_wave;
            flow += x;
        }
        if (c == 0 || flow > best) { best = flow; arg = c; }
    }
    action[i] = P.g_action[a * P.GC + arg];
}

// Work decomposition of one workgroup (= one env instance):
//   * lane threads (l < NLA) own one lane each: the head walk (phase H), the gather (phase B);
//   * with HELP, phase F evaluates the car-following law of every queued vehicle behind the first one that stays with
//     ALL threads of the workgroup, KF vehicles per thread in a flat, load-balanced order (prefix sum over the lane
//     counts; the lane of a flat index from byte marks + a wavefront prefix maximum).  A queued vehicle that is not the head of a platoon crossing in this very
//     second cannot cross, so its new speed depends only on OLD state (itself, the vehicle ahead, the signal):
//     min(follow(leader), follow(stop line) if the link is closed).  The walk then only applies the clamps and
//     the bookkeeping (~50 instructions per vehicle instead of ~500); heads and vehicles right behind a vehicle
//     that crossed this second are still evaluated inline.  Lanes are very unevenly loaded (a wavefront walks as
//     long as its fullest lane while most of its 64 threads idle), so this halves the VALU work of the kernel.
// Register budget: the 256-thread instantiation is capped at 128 VGPRs (4 waves / SIMD).  With 155 VGPRs the
// grid fits the chip with no slack, and whenever another kernel ran in between (i.e. always, in the training
// loop) XCD 0 admitted ~30 workgroups one full round late -> 178 us became 316 us per step
// (tools/bench_env.py, tsc_env_debug_clock).  The cap costs ~120 B of scratch per lane and 8 % in isolation.
// SPEC 1: the table dimensions of the reference's large_grid are compile-time constants (LDS offsets and index products
// fold into immediates: the kernel holds > 100 scalar values otherwise and reloads the spilled ones with v_readlane)
template <int MAXT, bool HELP, bool REC = false, int KF = 4, int SPEC = 0>      // REC: evaluation recording (plain walk only); KF: vehicles per thread and super-round of the flat phase
__global__ void __launch_bounds__(MAXT, MAXT <= 512 ? 4 : 1)      // (HIP: the second figure is wavefronts per SIMD) 128 VGPRs whatever the workgroup size
step_kernel(EnvDev P, const int *__restrict__ action, float *__restrict__ obs, double *__restrict__ reward,
            double *__restrict__ greward, uint8_t *__restrict__ done, int train_mode) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    if constexpr (SPEC > 0) {
        constexpr SpecDims D = kSpec[SPEC];
        P.NLP = D.NLP; P.NLA = D.NLA; P.NU = D.NU; P.NR = D.NR; P.A = D.A; P.KMAX = D.KMAX;
        P.PMAX = D.PMAX; P.LMAX = D.LMAX; P.NBR = D.NBR; P.ctrl = D.ctrl; P.yellow = D.yellow; P.teleport = D.teleport;
        P.NS = D.NR;                            // the reference's configurations: every route is its own stream
    }
    Smem s = carve(smem_raw, P);
    const int e = P.order ? P.order[blockIdx.x] : (int)blockIdx.x, l = threadIdx.x, NLP = P.NLP, NLA = P.NLA, NR = P.NR, NS = P.NS;
    const bool lane = l < P.NU;             // lanes >= NU are never entered by any route: always empty
    const bool lthr = l < NLA;              // threads >= NLA only help in phase A1 and in the strided loops
    const bool stamp = P.dbg && blockIdx.x == 0 && threadIdx.x == 0;
    int nstamp = 0;
    // A measurement build (-DTSC_ENV_PHASE_SUMS, tools/build_variant.sh) also lets every workgroup's thread 0 sum its shader-clock
    // cycles per kind of phase (tsc_env_debug_clock enable == 3): bucket 0 prologue + epilogue, 1 head walk, 2 flat phase, 3 gather +
    // demand, 4 the barriers between them.  Not in the product build: the sums live in registers over the whole kernel (25 spilled
    // dwords in the benchmarked instantiation).
#ifdef TSC_ENV_PHASE_SUMS
    const bool wsum = P.dbg && threadIdx.x == 0;
    long long pacc_[5] = {0, 0, 0, 0, 0}, wprev = 0;
    int wk = 0;
#define TSC_STAMP() do { if (stamp && nstamp < 62) P.dbg[nstamp++] = clock64();                                        \
        if (wsum) { const long long now_ = clock64();                                                                    \
            if (wk > 0) { const int d_ = wk - 1, r_ = (d_ - 1) % 6;                                                       \
                const int b_ = (d_ == 0 || d_ > 30) ? 0 : (r_ == 0 ? 1 : r_ == 2 ? 2 : r_ == 4 ? 3 : 4); pacc_[b_] += now_ - wprev; } \
            wprev = now_; ++wk; } } while (0)
#else
#define TSC_STAMP() do { if (stamp && nstamp < 62) P.dbg[nstamp++] = clock64(); } while (0)
#endif
    TSC_STAMP();
    if (P.dbg && threadIdx.x == 0) P.dbg[64 + 2 * blockIdx.x] = wall_clock64();
    float4 *S = P.S + (size_t)e * kCap * NLP;
    static_assert(!(HELP && REC), "recording uses the plain walk");
    uint32_t *R0 = REC ? P.R0 + (size_t)e * kCap * NLP : nullptr, *R1 = REC ? P.R1 + (size_t)e * kCap * NLP : nullptr;
    const float origin = REC && lane ? P.lane_origin[l] : 0.0f;
    if (REC && l < 4) s.rint[l] = 0;

    // ---- prologue.  Everything is requested in two levels -- first all loads that depend on nothing, then (under
    // the table copies) the ones that need a first-level value -- and unconditionally (lane index clamped), so
    // that the prologue costs a few memory round trips instead of one per table.
    const int lc = lane ? l : P.NU - 1;                                  // clamped lane for the loads
    const bool ag = l < P.A;
    const int ac = ag ? l : P.A - 1;
    // level 1
    const int act = action[(size_t)e * P.A + ac];
    const int prev = P.prev_action[(size_t)e * P.A + ac];
    int n = P.N[(size_t)e * NLP + lc];
    float L_ = P.lane_len[lc], vmax_ = P.lane_vmax[lc], det = P.lane_det[lc];
    int my_node = P.lane_node[lc];
    int up0 = P.lane_up[lc * kMaxUp], up1 = P.lane_up[lc * kMaxUp + 1], up2 = P.lane_up[lc * kMaxUp + 2], up3 = P.lane_up[lc * kMaxUp + 3];
    uint32_t myr_lo = P.lane_routes[lc * 2], myr_hi = P.lane_routes[lc * 2 + 1];     // entry routes, a byte each
    int t = P.tsec[e];
    const uint32_t seed = P.seed[e];
    const int rc = l < NS ? l : NS - 1;
    const int pend0 = P.pending[(size_t)e * NS + rc], ser0 = P.serial[(size_t)e * NS + rc];
    if (!lane) {
        n = 0; L_ = 1.0f; vmax_ = 1.0f; det = 0.0f; my_node = -1; up0 = up1 = up2 = up3 = -1;
        myr_lo = myr_hi = 0xFFFFFFFFu;
    }
    // table copies: four (clamped, unconditional) loads in flight per thread and round instead of one
    auto copy_words = [&](uint32_t *dst, const uint32_t *src, int count) {
        for (int base = l; base < count; base += 4 * (int)blockDim.x) {
            uint32_t w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int i = base + u * (int)blockDim.x; w[u] = src[i < count ? i : count - 1]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int i = base + u * (int)blockDim.x; if (i < count) dst[i] = w[u]; }
        }
    };
    // Round 6: the first 8 / 4 words per thread of the two tables are REQUESTED here, next to the level-1 loads, and stored behind the
    // level-2 requests (one memory round trip for both tables instead of one per 4-word round: three on large_grid -- the workgroup is a
    // latency chain, 64 us alone on a CU against 80 us with three neighbours, so a round trip is ~ 1 % of the step); what does not fit
    // (no reference scenario) takes the loop.
    constexpr int kTW = 8, kTZ = 4;
    const int n_mv = P.NU * NR, n_zip = (P.NU * NR + 3) / 4;                            // zip padded to 4 B
    uint32_t tw[kTW], tz[kTZ];
#pragma unroll
    for (int u = 0; u < kTW; ++u) { const int i = l + u * (int)blockDim.x; tw[u] = ((const uint32_t *)P.mv)[i < n_mv ? i : n_mv - 1]; }
#pragma unroll
    for (int u = 0; u < kTZ; ++u) { const int i = l + u * (int)blockDim.x; tz[u] = ((const uint32_t *)P.zip)[i < n_zip ? i : n_zip - 1]; }
    // level 2 (needs n / act / t), issued before the remaining LDS fills
    float hx = 0, hv = 0, hsf = 0, tx = 0, tv = 0; uint32_t hm = 0;      // head record (slot 0) / tail of my lane, kept in registers
    {
        const int nl = n > 0 ? n - 1 : 0;
        const int s0 = vslot(0, lc, NLP), sl = vslot(nl, lc, NLP);
        const float4 a0 = S[s0];
        const float2 al = *(const float2 *)(S + sl);
        if (n > 0) { hx = a0.x; hv = a0.y; hsf = a0.z; hm = __float_as_uint(a0.w); tx = al.x; tv = al.y; }
    }
#pragma unroll
    for (int u = 0; u < kTW; ++u) { const int i = l + u * (int)blockDim.x; if (i < n_mv) ((uint32_t *)s.mv)[i] = tw[u]; }
#pragma unroll
    for (int u = 0; u < kTZ; ++u) { const int i = l + u * (int)blockDim.x; if (i < n_zip) ((uint32_t *)s.zip)[i] = tz[u]; }
    if (n_mv > kTW * (int)blockDim.x) copy_words((uint32_t *)s.mv + kTW * blockDim.x, (const uint32_t *)P.mv + kTW * blockDim.x, n_mv - kTW * (int)blockDim.x);
    if (n_zip > kTZ * (int)blockDim.x) copy_words((uint32_t *)s.zip + kTZ * blockDim.x, (const uint32_t *)P.zip + kTZ * blockDim.x, n_zip - kTZ * (int)blockDim.x);
    // K1: signal FSM (envs/env.py:128-152) -> link chars for the yellow and the green interval
    if (ag) {
        P.prev_action[(size_t)e * P.A + l] = act;
        const uint8_t *g = P.green_tab + ((size_t)l * P.PMAX + act) * P.KMAX;
        const uint8_t *y = (prev < 0 || prev == act) ? g : P.yellow_tab + (((size_t)l * P.PMAX + prev) * P.PMAX + act) * P.KMAX;
        for (int k = 0; k < P.KMAX; ++k) { s.link_y[l * P.KMAX + k] = y[k]; s.link_g[l * P.KMAX + k] = g[k]; }
    }
    // per-route insertion state and this step's emissions live in LDS for the duration of the launch
    if (l < NS) {
        s.pend[l] = pend0; s.ser[l] = ser0;
        for (int q = 0; q < 8; ++q) s.emit[l * 8 + q] = q < P.ctrl ? P.emit_tab[(size_t)l * P.emit_len + t + q] : 0;
    }
    for (int r = blockDim.x + l; r < NS; r += blockDim.x) {              // more streams than threads (not on the reference scenarios)
        s.pend[r] = P.pending[(size_t)e * NS + r]; s.ser[r] = P.serial[(size_t)e * NS + r];
        for (int q = 0; q < 8; ++q) s.emit[r * 8 + q] = q < P.ctrl ? P.emit_tab[(size_t)r * P.emit_len + t + q] : 0;
    }
    for (int a2 = blockDim.x + l; a2 < P.A; a2 += blockDim.x) {          // more agents than threads (ditto)
        const int act2 = action[(size_t)e * P.A + a2], prev2 = P.prev_action[(size_t)e * P.A + a2];
        P.prev_action[(size_t)e * P.A + a2] = act2;
        const uint8_t *g = P.green_tab + ((size_t)a2 * P.PMAX + act2) * P.KMAX;
        const uint8_t *y = (prev2 < 0 || prev2 == act2) ? g : P.yellow_tab + (((size_t)a2 * P.PMAX + prev2) * P.PMAX + act2) * P.KMAX;
        for (int k = 0; k < P.KMAX; ++k) { s.link_y[a2 * P.KMAX + k] = y[k]; s.link_g[a2 * P.KMAX + k] = g[k]; }
    }
    for (int q = l; q < NLA; q += blockDim.x) {
        const bool in = q < P.NU;
        s.len[q] = in ? P.lane_len[q] : 1.0f; s.node[q] = in ? P.lane_node[q] : -1; s.vmax[q] = in ? P.lane_vmax[q] : 1.0f;
        int sbq = (in && P.lane_sib) ? P.lane_sib[q] : -1;
        s.sib[q] = sbq < P.NU ? sbq : -1;               // (a sibling no route ever reaches has no rows here and is never needed)
    }
    for (int q = l; q < NLP; q += blockDim.x) { s.wave[q] = 0; s.halt[q] = 0; s.hwait[q] = 0; }
    if constexpr (HELP) {                           // flat-phase marks: all clear (a second sets and clears its own)
        uint32_t *mk = (uint32_t *)s.mark;
        for (int q = l; q < (NLA * (kCap - 1) + 8) / 4; q += blockDim.x) mk[q] = 0u;
    }
    // publishes a lane's summary for the next second; with HELP also the wave-local inclusive scan of the number
    // of queued vehicles (slots >= 1) that phase A1 distributes over the workgroup
    auto publish = [&]() {
        if (!lthr) return;
        s.n[l] = n; s.hx[l] = hx; s.hv[l] = hv; s.hm[l] = hm; s.tx[l] = tx; s.tv[l] = tv;
        if constexpr (HELP) {
            const int inc = wave_scan_add(n > 1 ? n - 1 : 0);
            s.pre[l] = inc;
            if ((l & 63) == 63) s.wtot[l >> 6] = inc;
        }
    };
    publish();
    if (lthr) {
        auto b8 = [&](int u) { return (uint32_t)((u >= 0 && u < 255 && u < P.NU) ? u : 0xFF); };
        s.up4[l] = b8(up0) | b8(up1) << 8 | b8(up2) << 16 | b8(up3) << 24;
        s.nout[l] = 0;
    }
    unsigned arrived = 0, tele = 0;
    __syncthreads();
    TSC_STAMP();

    int d_wave = 0, d_halt = 0;                     // detector counts, taken during the last simulated second
#pragma unroll 1
    for (int sub = 0; sub < P.ctrl; ++sub, ++t) {
        const uint8_t *link = sub < P.yellow ? s.link_y : s.link_g;
        const bool last = sub == P.ctrl - 1;
        // ================= phase H (K2): the platoon that crosses in this second and the first vehicle that stays,
        // from the OLD state (without HELP: the whole lane, sequentially) =================
        int kept = 0, nsent = 0;
        bool has_first = false;                        // HELP: the first stayer is written in phase B (its old slot is
        float fxn = 0.0f, fvn = 0.0f, fsf = 0.0f;      // still read by the flat phase)
        uint32_t fmeta = 0u;
        int i0 = 0;
        // recording: this lane's share of the per-second network statistics (envs/env.py:409-437)
        int rq_halt = 0, rq_wait = 0, rq_arr = 0, rq_dep = 0;
        double rq_speed = 0.0;
        auto tally = [&](float xn, float vn, uint32_t nmeta) {
            if constexpr (REC) {
                rq_wait += (int)(nmeta & 0xFFFFu);
                rq_speed += (double)vn;
                if (vn < kHalt && xn >= origin) ++rq_halt;
            }
        };
        if constexpr (HELP) {
            // flat order of the queued vehicles (slots >= 1) of the instance, lane after lane: this lane's first flat index P0
            // (wave-local inclusive scan from publish() + the totals of the lane wavefronts before mine), a mark at P0 and the
            // owner of the one multiple of 64 its <= 27 vehicles can cover -- what the flat phase needs to turn a flat index
            // into (lane, slot) with one LDS read, a wavefront max-scan and one more read (a 6-step binary search per vehicle
            // before round 4)
            if (lthr) {
                const int c = n > 1 ? n - 1 : 0;
                int P0 = s.pre[l] - c;
                for (int w = 0; w < (l >> 6); ++w) P0 += s.wtot[w];
                s.pre[l] = P0;
                if (c > 0) {
                    s.mark[P0] = (uint8_t)((l & 63) + 1);
                    const int m64 = (P0 + c - 1) & ~63;
                    if (m64 >= P0) s.bound[m64 >> 6] = (uint16_t)(l + 1);
                }
            }
            // From here on the lane wavefronts walk their heads (phase H) while every other wavefront already starts the flat
            // phase's first half (locate, load, car-following: old state only) -- the two only meet at the barrier in front of
            // the chain keys (round 4; before, all helper wavefronts idled through phase H).
            __syncthreads();
        }
        if (lane) {
            // lane constants from LDS per phase (HELP): two registers less across the flat phase than holding them for the launch
            const float L = HELP ? s.len[l] : L_, vmax = HELP ? s.vmax[l] : vmax_;
            int ncross = 0;
            bool all_crossed = true;
            float pnx = INFINITY, pox = 0.0f, pov = 0.0f;
            float K = INFINITY;                            // plain walk: running min of the chain keys (MICROSIM_SPEC.md rule 4)
            // a vehicle that stays on the lane: compact it to slot `kept`, refresh the summary, count detectors
            auto keep = [&](float xn, float vn, float sf, uint32_t nmeta, uint32_t r0 = 0u, uint32_t r1 = 0u) {
                const unsigned ob = (unsigned)vslot(kept, l, NLP) * 4u;
                stg(S, 4u * ob, make_float4(xn, vn, sf, __uint_as_float(nmeta)));
                if constexpr (REC) { stg(R0, ob, r0); stg(R1, ob, r1); tally(xn, vn, nmeta); }
                if (kept == 0) { hx = xn; hv = vn; hsf = sf; hm = nmeta; }
                tx = xn; tv = vn;
                ++kept;
                if (last && xn >= det) { ++d_wave; if (vn < kHalt) ++d_halt; }
            };
            // ---- head walk: full evaluation.  Without HELP it covers the whole lane; with HELP only the platoon
            // that is crossing in this second plus the first vehicle that stays (everything behind it is phase A1).
            struct Raw { float x, v, sf; uint32_t m, r0, r1; };
            // Loads are UNCONDITIONAL (slot index clamped; every slot is allocated): the compiler can only keep a
            // load in flight across the evaluation of the previous vehicle (s_waitcnt vmcnt(N > 0)) when the
            // number of younger memory operations is known, which a load under `if (i < n)` destroys.
            auto load_raw = [&](int i) {
                const unsigned ob = (unsigned)vslot(i < kCap ? i : kCap - 1, l, NLP) * 4u;
                const float4 a4 = ldg(S, 4u * ob);
                Raw r; r.x = a4.x; r.v = a4.y; r.sf = a4.z; r.m = __float_as_uint(a4.w);
                r.r0 = 0u; r.r1 = 0u;
                if constexpr (REC) { r.r0 = ldg(R0, ob); r.r1 = ldg(R1, ob); }
                return r;
            };
            int i = 0;
            // the head's record is what this thread wrote last (hx / hv / hsf / hm track slot 0): with HELP the walk starts from the
            // registers instead of waiting for a load of it -- one memory round trip less on the second's critical path
            Raw cur;
            if constexpr (HELP && !REC) { cur.x = hx; cur.v = hv; cur.sf = hsf; cur.m = hm; cur.r0 = 0u; cur.r1 = 0u; }
            else cur = load_raw(0);
            for (; i < n; ++i) {
                if (HELP && !all_crossed) break;
                const Raw nxt = load_raw(i + 1);
                const float x = cur.x, v = cur.v, sf = cur.sf;
                const uint32_t meta = cur.m;
                int w = (int)(meta & 0xFFFFu);
                const int r = (int)(meta >> 16);
                const int mvp = s.mv[l * NR + r];
                const int tl = mv_tl(mvp), k = mv_k(mvp), y = mv_yield(mvp), z = s.zip[l * NR + r];
                const float v0 = vmax * sf;
                const bool sink = tl == -1;
                const bool open = sig_open(tl, k, HELP ? s.node[l] : my_node, w, x, v, L, link, P.KMAX, P.teleport);
                bool can_cross = false, lc = false;
                int sb = -1;                                   // rule 10: the sibling lane this vehicle has to move over to
                if (all_crossed) {
                    can_cross = open;
                    if (can_cross && y >= 0 && w < P.teleport && s.n[y] > 0) {
                        // right of way: wait while the yield lane's head has an open priority movement and is near
                        const uint32_t om = s.hm[y];
                        const int mo = s.mv[y * NR + (int)(om >> 16)];
                        if (mv_prio(mo)) {
                            const float xo = s.hx[y], vo = s.hv[y], Lo = s.len[y];
                            if (sig_open(mv_tl(mo), mv_k(mo), s.node[y], (int)(om & 0xFFFFu), xo, vo, Lo, link, P.KMAX, P.teleport)) {
                                const float d = Lo - xo;
                                if (d < vo * kYieldT + kYieldD) can_cross = false;
                            }
                        }
                    }
                    // zipper merge by readiness: of the feeders of tl whose HEAD wants tl, sees an open signal and can reach
                    // its stop line within this second, the one with the smallest rotating rank (t + rank) % count sends
                    // (large_grid, SPEC 1, has no unsignalised merge: its zip table is all zero, checked at create time)
                    if (SPEC != 1 && can_cross && (z >> 4) > 1) {
                        const uint32_t ups = s.up4[tl];
                        int best = -1, bestp = 1 << 30;
#pragma unroll
                        for (int u = 0; u < kMaxUp; ++u) {
                            const int f = (int)((ups >> (8 * u)) & 0xFFu);
                            if (f == 0xFF || s.n[f] == 0) continue;
                            const uint32_t om = s.hm[f];
                            const int ro = (int)(om >> 16), mo = s.mv[f * NR + ro];
                            if (mv_tl(mo) != tl) continue;
                            const int zo = s.zip[f * NR + ro], cnt = zo >> 4;
                            if (cnt <= 1) continue;
                            const float xo = s.hx[f], vo = s.hv[f], Lo = s.len[f];
                            if (!sig_open(tl, mv_k(mo), s.node[f], (int)(om & 0xFFFFu), xo, vo, Lo, link, P.KMAX, P.teleport)) continue;
                            if (!((Lo - xo) < vo + kAcc)) continue;
                            const int pr = (t + (zo & 0xF)) % cnt;
                            if (pr < bestp) { bestp = pr; best = f; }
                        }
                        if (best != l) can_cross = false;
                    }
                    if (can_cross && tl >= 0 && s.n[tl] + kMaxCross > kCap) can_cross = false;
                    if (can_cross && tl >= 0 && s.n[tl] > 0 && s.tx[tl] < kLen) can_cross = false;   // no room behind the tail
                    if (can_cross && ncross >= kMaxCross) can_cross = false;
                    if (tl < -1) can_cross = false;
                    // rule 10: a head-platoon vehicle on the wrong lane of a two-lane street moves over to the sibling lane as a
                    // hand-off that keeps its position: behind the sibling's OLD tail by the standstill gap + 1 s of the closing
                    // speed, room for two full platoons there (the sibling may receive from its junction in the same second)
                    if (tl < -1) {
                        const int sq = s.sib[l];
                        if (sq >= 0 && mv_tl(s.mv[sq * NR + r]) >= -1) {
                            sb = sq;
                            const int nsb = s.n[sb];
                            lc = (nsb + 2 * kMaxCross <= kCap) && (ncross < kMaxCross);
                            if (lc && nsb > 0) {
                                float dv = v - s.tv[sb];
                                if (dv < 0.0f) dv = 0.0f;
                                lc = ((s.tx[sb] - kLen) - x) >= (kS0 + dv);
                            }
                        }
                    }
                }
                // teleport (SUMO --time-to-teleport): the head, standing for >= teleport seconds, whose way is still blocked
                // (merge slot, capacity, no room behind the target's tail) leaves the network where it stands.  SUMO would
                // move it along its route and its trip would end later: the surrogate TRUNCATES the trip, so it is counted
                // as a teleport, not as an arrival, and its trip row carries a negative arrival second
                if (i == 0 && !can_cross && !lc && (tl >= 0 || sb >= 0) && w >= P.teleport) {
                    ++tele;
                    if constexpr (REC) {
                        const int k = atomicAdd(&P.n_trips[e], 1);
                        if (k < P.trip_cap) {
                            int *tr = P.trips + ((size_t)e * P.trip_cap + k) * 6;
                            tr[0] = r; tr[1] = (int)(cur.r0 >> 16); tr[2] = (int)(cur.r0 & 0xFFFFu); tr[3] = -(t + 1);
                            tr[4] = (int)(cur.r1 & 0xFFFFu); tr[5] = (int)(cur.r1 >> 16);
                        }
                    }
                    ++ncross;
                    pox = x; pov = v;
                    cur = nxt;
                    continue;
                }
                // (rule 10: a lane change is a hand-off to the sibling lane `sb` at distance 0 instead of L - x, with the stop line
                //  closed -- no lane change and junction crossing in one second; it shares the evaluation below)
                const bool line_block = lc || (all_crossed ? !can_cross : !open);
                const bool tgt_lead = lc ? s.n[sb] > 0 : (can_cross && !sink && s.n[tl] > 0);
                const int tq = tgt_lead ? (lc ? sb : tl) : 0;      // the lane whose old tail is the leader of the lane's head
                const float Lq = lc ? 0.0f : L;                    // distance to that lane's start
                // leader: vehicle ahead (old state), else the old tail of the target lane, else free road; the stop line
                // is a second leader when the link is closed (evaluating both branch-free was measured no faster and
                // costs 40 VGPRs)
                const bool has_lead = i > 0 || tgt_lead;
                const float lg = i > 0 ? (pox - kLen) - x : (Lq - x) + (s.tx[tq] - kLen);
                const float lvl = i > 0 ? pov : s.tv[tq];
                float vn = follow(v, v0, has_lead, lg, lvl, kS0);
                if (line_block) {
                    const float v2 = follow(v, v0, true, L - x, 0.0f, 0.0f);
                    if (v2 < vn) vn = v2;
                }
                // rule 10: the head of a lane that has to move over lines up BEHIND the sibling's queue instead of driving past
                // it: the sibling's old tail is a third leader while it is still ahead
                if (sb >= 0 && !lc && i == 0 && s.n[sb] > 0) {
                    const float g3 = (s.tx[sb] - kLen) - x;
                    if (g3 >= 0.0f) {
                        const float v3 = follow(v, v0, true, g3, s.tv[sb], kS0);
                        if (v3 < vn) vn = v3;
                    }
                }
                float xn = x + vn;
                if (!HELP && !all_crossed) {
                    // behind the first vehicle that stays: x' = min(x + v', L, K - 5 i), K = exclusive prefix-min of the
                    // keys a_j + 5 j (the closed form of x'_i = min(a_i, x'_{i-1} - 5); the flat phase scans it)
                    const float xf = xn;
                    float a = xf;
                    if (a > L) a = L;
                    const float lim = K - (float)(5 * i);
                    xn = a < lim ? a : lim;
                    if (xn < x) xn = x;
                    if (xn != xf) vn = xn - x;
                    const float key = a + (float)(5 * i);
                    if (key < K) K = key;
                } else {
                    bool clamped = false;
                    if (xn > pnx - kLen) { xn = pnx - kLen; clamped = true; }
                    if (tgt_lead) {
                        const float lim = Lq + (s.tx[tq] - kLen);
                        if (xn > lim) { xn = lim; clamped = true; }
                    }
                    if (can_cross && !sink) {                          // target lane shorter than one step's travel
                        const float far = L + s.len[tl];
                        if (xn > far) { xn = far; clamped = true; }
                    }
                    if (!can_cross && xn > L) { xn = L; clamped = true; }
                    if (xn < x) { xn = x; clamped = true; }
                    if (clamped) vn = xn - x;
                }
                uint32_t r1n = cur.r1;
                if constexpr (REC) {                   // tripinfo waitingTime / waitingCount
                    if (vn < kHalt) r1n = (r1n + 1u) + (w == 0 ? 0x10000u : 0u);
                }
                w = (vn < kHalt) ? w + 1 : 0;
                pnx = xn; pox = x; pov = v;
                const uint32_t nmeta = (uint32_t)w | ((uint32_t)r << 16);
                if (lc) {                                   // rule 10: over to the sibling lane, position kept
                    const int o = nsent * NLA + l;
                    s.ox[o] = xn; s.ov[o] = vn; s.osf[o] = sf; s.om[o] = nmeta; s.oto[o] = sb;
                    if constexpr (REC) { s.or0[o] = cur.r0; s.or1[o] = r1n; }
                    ++nsent; ++ncross;
                } else if (can_cross && xn >= L) {
                    if (!sink) {
                        const int o = nsent * NLA + l;
                        const float ex = xn - L, Lt = s.len[tl];          // rounding of (L + Lt) - L
                        s.ox[o] = ex > Lt ? Lt : ex; s.ov[o] = vn; s.osf[o] = sf; s.om[o] = nmeta; s.oto[o] = tl;
                        if constexpr (REC) { s.or0[o] = cur.r0; s.or1[o] = r1n; }
                        ++nsent;
                    } else {
                        ++arrived;
                        if constexpr (REC) {
                            ++rq_arr;
                            const int k = atomicAdd(&P.n_trips[e], 1);
                            if (k < P.trip_cap) {
                                int *tr = P.trips + ((size_t)e * P.trip_cap + k) * 6;
                                tr[0] = r; tr[1] = (int)(cur.r0 >> 16); tr[2] = (int)(cur.r0 & 0xFFFFu); tr[3] = t + 1;
                                tr[4] = (int)(r1n & 0xFFFFu); tr[5] = (int)(r1n >> 16);
                            }
                        }
                    }
                    ++ncross;
                } else {
                    if (all_crossed) K = xn + (float)(5 * i);      // the first vehicle that stays seeds the chain
                    all_crossed = false;
                    if constexpr (HELP) {
                        has_first = true; i0 = i; fxn = xn; fvn = vn; fsf = sf; fmeta = nmeta;
                        if (last && xn >= det) { ++d_wave; if (vn < kHalt) ++d_halt; }
                    } else {
                        keep(xn, vn, sf, nmeta, cur.r0, r1n);
                    }
                }
                cur = nxt;
            }
            if constexpr (HELP) {
                if (!has_first) i0 = n;                            // everybody crossed
                s.nc[l] = i0;
                s.seed[l] = has_first ? fxn + (float)(5 * i0) : INFINITY;
            }
            s.nout[l] = nsent;
        }
        TSC_STAMP();
        if constexpr (!HELP) __syncthreads();          // (HELP: the barrier sits inside the flat phase, behind its first half)
        TSC_STAMP();
        // ================= phase F (K2, HELP): every vehicle behind the first stayer, one per thread slot =============
        // Nobody behind a vehicle that stays can cross, so such a vehicle's new speed depends only on OLD state (itself, the
        // vehicle ahead, the signal) and its new position on the chain clamp x'_i = min(a_i, x'_{i-1} - 5), a_i = min(x_i +
        // v'_i, L).  The clamp is an exclusive prefix-min over the keys a_j + 5 j within a lane (MICROSIM_SPEC.md rule 4): all
        // queued vehicles of the instance are laid out flat over the workgroup (prefix sum of the lane counts, lane marks +
        // a prefix maximum to find a flat index's lane), kF consecutive ones per thread, and the chain is a segmented
        // min-scan -- in the thread, then over the wavefront on the DPP path, then across wavefronts through one LDS word
        // per wave (a lane's <= 27 queued vehicles touch at most two wavefronts).  Replaces the A1 scratch round trip
        // through HBM and the 28-deep sequential lane walk of round 1.
        if constexpr (HELP) {
            const int nseg = NLA >> 6;
            int total = 0;
            for (int w = 0; w < nseg; ++w) total += s.wtot[w];
            constexpr int kF = KF;
            // flat-phase thread index: rotated so that the lane wavefronts (busy with phase H first) own the LAST flat indices
            const int lf = (int)blockDim.x > NLA ? (l >= NLA ? l - NLA : l + ((int)blockDim.x - NLA)) : l;
            const int wv = lf >> 6, wl = lf & 63, nwv = (int)blockDim.x >> 6;
            int round = 0;
            for (int base = 0; base < total; base += kF * (int)blockDim.x, ++round) {
                const float carry_round = round > 0 ? s.wtail[((round - 1) & 1) * 16 + nwv - 1] : INFINITY;
                // A wavefront whose first flat index is past the total has no vehicle in this (last) super-round: it only
                // keeps the barrier.  The kernel is VALU-issue bound and the four workgroups of a CU share its SIMDs, so the
                // ~600 instructions such a wave would run on clamped indices are real time for the others (on average half a
                // super-round: 17-25 % of the phase at 2-3 super-rounds).
                const bool wave_on = base + kF * (int)((unsigned)lf & ~63u) < total;
                int eq[kF], ei[kF];
                float x[kF], v[kF], sf[kF], px[kF], pv[kF];
                uint32_t m[kF];
                bool act[kF];
                float vn[kF], a[kF], key[kF], loc[kF];
                bool run0[kF];
                float pvs = INFINITY;
                int pfs = 1;
#pragma unroll
                for (int u = 0; u < kF; ++u) {
                    eq[u] = 0; ei[u] = 0; x[u] = v[u] = sf[u] = px[u] = pv[u] = 0.0f; m[u] = 0u; act[u] = false;
                    vn[u] = a[u] = 0.0f; key[u] = loc[u] = INFINITY; run0[u] = true;
                }
                if (wave_on) {
                    // flat index -> (lane, slot): the owner of index k is the last lane whose mark lies at or before k.  Marks hold
                    // the lane's index within its wavefront (a byte); the wavefront comes from the segment the index falls into, so
                    // the owners are increasing along the flat order and a prefix maximum finds them.
                    int kks[kF], val[kF];
    #pragma unroll
                    for (int u = 0; u < kF; ++u) {
                        const int k = base + kF * lf + u;
                        act[u] = k < total;
                        const int kk = act[u] ? k : total - 1;        // clamped: locate and load unconditionally
                        kks[u] = kk;
                        int w = 0, kr = kk;
                        for (; w < nseg - 1; ++w) {
                            const int c = s.wtot[w];
                            if (kr < c) break;
                            kr -= c;
                        }
                        const int m = (int)s.mark[kk];
                        val[u] = m ? (w << 6) + m : 0;
                    }
                    int lmax[kF], rmax = 0;
    #pragma unroll
                    for (int u = 0; u < kF; ++u) { rmax = max(rmax, val[u]); lmax[u] = rmax; }
                    const int incl = wave_scan_max(rmax);
                    int excl = TSC_DPP_I(0, incl, kDppWaveShr1, 0xF);                      // lanes before me; lane 0: none
                    excl = max(excl, (int)s.bound[(base >> 6) + kF * (lf >> 6)]);           // owner of this wavefront's first index
    #pragma unroll
                    for (int u = 0; u < kF; ++u) {
                        const int q = max(lmax[u], excl) - 1;
                        const int i = kks[u] - s.pre[q] + 1;
                        eq[u] = q; ei[u] = i;
                        const unsigned ob = (unsigned)vslot(i, q, NLP) * 4u, pb = (unsigned)vslot(i - 1, q, NLP) * 4u;
                        const float4 a4 = ldg(S, 4u * ob);
                        const float2 p2 = ldg((const float2 *)S, 4u * pb);
                        x[u] = a4.x; v[u] = a4.y; sf[u] = a4.z; m[u] = __float_as_uint(a4.w);
                        px[u] = p2.x; pv[u] = p2.y;
                    }
                    if (round > 0 && lf == 0 && ei[0] > 1) { px[0] = s.hz[0]; pv[0] = s.hz[1]; }   // overwritten by the previous super-round
    #pragma unroll
                    for (int u = 0; u < kF; ++u) {
                        const int q = eq[u];
                        const float Lq = s.len[q];
                        const int mvp = s.mv[q * NR + (int)(m[u] >> 16)];
                        const float v0 = s.vmax[q] * sf[u];
                        const bool open = sig_open(mv_tl(mvp), mv_k(mvp), s.node[q], (int)(m[u] & 0xFFFFu), x[u], v[u], Lq, link, P.KMAX, P.teleport);
                        float vv = follow(v[u], v0, true, (px[u] - kLen) - x[u], pv[u], kS0);
                        if (!open) {
                            const float v2 = follow(v[u], v0, true, Lq - x[u], 0.0f, 0.0f);
                            if (v2 < vv) vv = v2;
                        }
                        vn[u] = vv;
                        float aa = x[u] + vv;
                        if (aa > Lq) aa = Lq;
                        a[u] = aa;
                    }
                }
                // ---- everything above used OLD state only; what follows needs phase H's results (nc, seed) of this second
                if (round == 0) __syncthreads();
                if (wave_on) {
                    float run = INFINITY;
                    bool first_run = true;
    #pragma unroll
                    for (int u = 0; u < kF; ++u) {
                        const int q = eq[u];
                        const float aa = a[u];
                        const bool live = act[u] && ei[u] > s.nc[q];
                        act[u] = live;
                        key[u] = live ? aa + (float)(5 * ei[u]) : INFINITY;
                        if (u > 0 && eq[u] != eq[u - 1]) { run = INFINITY; first_run = false; }
                        loc[u] = run; run0[u] = first_run;
                        run = fminf(run, key[u]);
                    }
                    // segmented inclusive min-scan of the threads' last runs over the wavefront
                    float sv = run;
                    int sfl = (!first_run || ei[0] == 1) ? 1 : 0;
                    wave_seg_scan_min(sv, sfl);
                    pvs = TSC_DPP_F(sv, sv, kDppWaveShr1, 0xF);       // lane 0 keeps its own value (not read: its carry is the
                    pfs = TSC_DPP_I(sfl, sfl, kDppWaveShr1, 0xF);     // previous wavefront's tail)
                    if (wl == 63) s.wtail[(round & 1) * 16 + wv] = sv;
                }
                if (lf == (int)blockDim.x - 1) { s.hz[2] = x[kF - 1]; s.hz[3] = v[kF - 1]; }
                __syncthreads();
                if (lf == 0) { s.hz[0] = s.hz[2]; s.hz[1] = s.hz[3]; }      // read by flat thread 0 after the next barrier only
                const float prev_tail = wv > 0 ? s.wtail[(round & 1) * 16 + wv - 1] : carry_round;
                const float carry = ei[0] > 1 ? (wl == 0 ? prev_tail : (pfs ? pvs : fminf(pvs, prev_tail))) : INFINITY;
#pragma unroll
                for (int u = 0; u < kF; ++u) {
                    if (!act[u]) continue;
                    const int q = eq[u], i = ei[u];
                    const float Kp = fminf(s.seed[q], fminf(run0[u] ? carry : INFINITY, loc[u]));
                    const float xf = x[u] + vn[u];
                    const float lim = Kp - (float)(5 * i);
                    float xn = a[u] < lim ? a[u] : lim;
                    if (xn < x[u]) xn = x[u];
                    float vv = vn[u];
                    if (xn != xf) vv = xn - x[u];
                    const uint32_t w = (vv < kHalt) ? (m[u] & 0xFFFFu) + 1u : 0u;
                    const uint32_t nmeta = w | (m[u] & 0xFFFF0000u);
                    const int shift = s.nc[q];
                    const unsigned ob = (unsigned)vslot(i - shift, q, NLP) * 4u;
                    stg(S, 4u * ob, make_float4(xn, vv, sf[u], __uint_as_float(nmeta)));
                    if (i == s.n[q] - 1) { s.tx[q] = xn; s.tv[q] = vv; }
                    if (last && xn >= P.lane_det[q]) { atomicAdd(&s.wave[q], 1); if (vv < kHalt) atomicAdd(&s.halt[q], 1); }
                }
                if (base + kF * (int)blockDim.x < total) __syncthreads();     // next super-round reads what this one stored
            }
            TSC_STAMP();
            __syncthreads();
            TSC_STAMP();
            {   // this second's marks all lie below `total`: clear them for the next one (which sets its own after the next barrier)
                uint32_t *mk = (uint32_t *)s.mark;
                for (int q = l; q < (total + 3) / 4; q += blockDim.x) mk[q] = 0u;
            }
        }
        // ================= phase B (K3): gather hand-offs from feeder lanes, then demand =========
        if (lane) {
            if constexpr (HELP) {
                // the first vehicle that stays becomes slot 0 now that nobody reads its old slot any more; the rest of the
                // lane was compacted behind it by the flat phase
                kept = has_first ? n - i0 : 0;
                if (has_first) {
                    const unsigned ob0 = (unsigned)vslot(0, l, NLP) * 4u;
                    stg(S, 4u * ob0, make_float4(fxn, fvn, fsf, __uint_as_float(fmeta)));
                    hx = fxn; hv = fvn; hsf = fsf; hm = fmeta;
                    if (kept >= 2) { tx = s.tx[l]; tv = s.tv[l]; } else { tx = fxn; tv = fvn; }
                }
            }
            n = kept;
            uint32_t ups = 0xFFFFFFFFu;
            if constexpr (SPEC > 0) ups = s.up4[l];          // the feeders as bytes (0xFF = none): four registers less across the flat phase
            for (int u = 0; u < kMaxUp; ++u) {
                int src;
                if constexpr (SPEC > 0) { src = (int)((ups >> (8 * u)) & 0xFFu); if (src == 0xFF) src = -1; }
                else src = u == 0 ? up0 : u == 1 ? up1 : u == 2 ? up2 : up3;
                if (src < 0) continue;
                const int cnt = s.nout[src];
                for (int j = 0; j < cnt; ++j) {
                    const int o = j * NLA + src;
                    if (s.oto[o] == l && n < kCap) {
                        const int d = vslot(n, l, NLP);
                        float ax = s.ox[o];
                        const float av = s.ov[o];
                        const uint32_t am = s.om[o];
                        if (n > 0 && ax > tx - kLen) {                   // arrivals of two feeders in one second
                            ax = tx - kLen;
                            if (ax < 0.0f) ax = 0.0f;
                        }
                        S[d] = make_float4(ax, av, s.osf[o], __uint_as_float(am));
                        if constexpr (REC) { R0[d] = s.or0[o]; R1[d] = s.or1[o]; tally(ax, av, am); }
                        if (n == 0) { hx = ax; hv = av; hsf = s.osf[o]; hm = am; }
                        tx = ax; tv = av;
                        ++n;
                        if (last && ax >= det) { ++d_wave; if (av < kHalt) ++d_halt; }
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < kMaxEntry; ++q) {               // my entry routes, ascending (packed: 0xFF ends the list)
                const int r = (int)(((q < 4 ? myr_lo : myr_hi) >> (8 * (q & 3))) & 0xFFu);
                if (r == 0xFF) break;
                int pend = s.pend[r] + (int)s.emit[r * 8 + sub];
                int ser = s.ser[r];
                if (pend > 0 && n < kCap) {
                    const float L = HELP ? s.len[l] : L_;
                    const float xt = n > 0 ? tx : (L + kLen) + kS0;
                    float xmax = (xt - kLen) - kS0;
                    float xlo = kLen;
                    if (SPEC != 1 && P.sorigin) {                    // the SUMO entry lane is one piece of this (contracted) lane
                        xlo = P.sorigin[2 * r] + kLen;
                        const float hi = P.sorigin[2 * r + 1];
                        if (xmax > hi) xmax = hi;
                    }
                    if (!(xmax < xlo)) {
                        const float u0 = u01(hash32(seed, (uint32_t)r, (uint32_t)ser, 0));
                        const float u1 = u01(hash32(seed, (uint32_t)r, (uint32_t)ser, 1));
                        const float u2 = u01(hash32(seed, (uint32_t)r, (uint32_t)ser, 2));
                        const float ax = xlo + u0 * (xmax - xlo);
                        const float asf = 1.0f + 0.2f * ((u1 + u2) - 1.0f);
                        int rt = r;                                  // the vehicle's route: the stream itself, ...
                        if (SPEC == 0 && P.sroute) {                 // (the specialised instantiations have one-to-one streams)
                            const int md = P.smode[r];
                            rt = P.sroute[r];                        // ... the stream's fixed route, ...
                            if (md == 2) rt = P.iroute[(size_t)e * NS + r];      // ... this episode's draw of the host, ...
                            if (md == 1) {                           // ... or this vehicle's draw from the turn ratios
                                const int u16 = (int)(hash32(seed, (uint32_t)r, (uint32_t)ser, 3) >> 16);
                                const int iv = t / P.ilen < P.NI ? t / P.ilen : P.NI - 1;
                                const int *ch = P.schoice + ((size_t)r * P.NI + iv) * P.KC * 2;
                                rt = ch[0];
                                for (int c = 0; c < P.KC; ++c) {
                                    if (ch[2 * c] < 0) break;
                                    rt = ch[2 * c];
                                    if (u16 < ch[2 * c + 1]) break;
                                }
                            }
                        }
                        const uint32_t am = (uint32_t)rt << 16;
                        const int d = vslot(n, l, NLP);
                        S[d] = make_float4(ax, 0.0f, asf, __uint_as_float(am));
                        if constexpr (REC) { R0[d] = (uint32_t)t | ((uint32_t)ser << 16); R1[d] = 0u; tally(ax, 0.0f, am); ++rq_dep; }
                        if (n == 0) { hx = ax; hv = 0.0f; hsf = asf; hm = am; }
                        tx = ax; tv = 0.0f;
                        ++n;
                        if (last && ax >= det) { ++d_wave; ++d_halt; }
                        --pend;
                        ++ser;
                    }
                }
                s.pend[r] = pend; s.ser[r] = ser;               // a route has exactly one entry lane: no race
            }
        }
        publish();
        if constexpr (REC) {
            if (lthr) { s.rq[l] = rq_halt; s.rsp[l] = rq_speed; }
            if (lane) {
                atomicAdd((unsigned long long *)&s.rint[0], (unsigned long long)n); atomicAdd((unsigned long long *)&s.rint[1], (unsigned long long)rq_dep);
                atomicAdd((unsigned long long *)&s.rint[2], (unsigned long long)rq_arr); atomicAdd((unsigned long long *)&s.rint[3], (unsigned long long)rq_wait);
            }
        }
        TSC_STAMP();
        __syncthreads();
        TSC_STAMP();
        if constexpr (REC) {
            // flush second `sub` of this control step: integers, speed (per-lane partial sums added in lane order: the
            // oracle's association), halting vehicles of every incoming lane in (agent, lane) order
            const size_t row = (size_t)e * 8 + sub;
            if (l < 4) P.rec_int[row * 4 + l] = s.rint[l];
            if (l == 0) {
                double sp = 0.0;
                for (int q = 0; q < P.NU; ++q) sp += s.rsp[q];
                P.rec_speed[row] = sp;
            }
            for (int p2 = l; p2 < P.A * P.LMAX; p2 += blockDim.x) {
                const int ln = P.agent_lanes[p2];
                P.rec_queue[row * (P.A * P.LMAX) + p2] = (ln >= 0 && ln < P.NU) ? s.rq[ln] : 0;
            }
            __syncthreads();
            if (l < 4) s.rint[l] = 0;
        }
    }

    // ---- K4: detectors (envs/env.py:325-407): wave, halting, wait of the front-most vehicle
    {   // window-mean live vehicles: one atomic per wavefront
        int tot = lane ? n : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) tot += __shfl_down(tot, o, 64);
        if ((l & 63) == 0 && tot) atomicAdd(&P.live_acc[e], (unsigned long long)tot);
    }
    if (lane) {
        P.N[(size_t)e * NLP + l] = n;
        // counts were taken while the last simulated second wrote the vehicles; the front-most vehicle is slot 0
        const int hw = (n > 0 && hx >= det && hx > 0.0f) ? (int)(hm & 0xFFFFu) : 0;
        s.wave[l] += d_wave; s.halt[l] += d_halt; s.hwait[l] = hw;       // the flat phase added its vehicles with LDS atomics
    }
    for (int r = l; r < NS; r += blockDim.x) { P.pending[(size_t)e * NS + r] = s.pend[r]; P.serial[(size_t)e * NS + r] = s.ser[r]; }
    if (arrived) atomicAdd(&P.arrived[e], (unsigned long long)arrived);
    if (tele) atomicAdd(&P.teleported[e], (unsigned long long)tele);
    if (l == 0) { P.tsec[e] = t; done[e] = t >= P.episode ? 1 : 0; }
    TSC_STAMP();
    __syncthreads();
    TSC_STAMP();

    // ---- K5: observations.  The float64 normalisation (envs/env.py:439-442, a division) is done once per lane,
    // not once per observation entry; the entries then only gather.  The outbox arrays are dead by now.
    float *o_wave = s.ox, *o_coop = s.ov, *o_wait = s.osf;        // [NLP] each (kMaxCross * NLA >= NLP)
    int *qacc = (int *)s.om, *wacc = qacc + P.A;                   // per-agent queue / wait sums (integers: exact in any order)
    for (int q = l; q < NLP; q += blockDim.x) {
        const double wv = norm_clip((double)s.wave[q], P.norm_wave, P.clip_wave);
        o_wave[q] = (float)wv;
        o_coop[q] = (float)(wv * P.coop_gamma);
        o_wait[q] = (float)norm_clip((double)s.hwait[q], P.norm_wait, P.clip_wait);
    }
    for (int a = l; a < 2 * P.A; a += blockDim.x) qacc[a] = 0;
    __syncthreads();
    {
        const int tot = P.A * P.SMAX;
        const float *fp = (P.fp_bound ? P.fp_bound : P.fp) + (size_t)e * P.A * P.PMAX;
        float *ob = obs + (size_t)e * tot;
        constexpr int kOB = 6;                                     // entries per thread and round, loads first
        for (int base = l; base < tot; base += kOB * (int)blockDim.x) {
            int ks[kOB];
#pragma unroll
            for (int u = 0; u < kOB; ++u) { const int idx = base + u * (int)blockDim.x; ks[u] = P.obs_ks[idx < tot ? idx : tot - 1]; }
#pragma unroll
            for (int u = 0; u < kOB; ++u) {
                const int idx = base + u * (int)blockDim.x;
                if (idx >= tot) continue;
                const int kind = ks[u] >> 16, src = ks[u] & 0xFFFF;
                float o = 0.0f;
                if (kind == 1) o = o_wave[src];
                else if (kind == 2) o = o_coop[src];
                else if (kind == 3) o = o_wait[src];
                else if (kind == 4) o = fp[src];
                ob[idx] = o;
            }
        }
    }
    TSC_STAMP();
    // ---- K6: reward (envs/env.py:356-367) and shaping (:580,:590-631), float64.  Queue and wait of an agent are
    // sums of small integers: accumulated with LDS atomics over all (agent, lane) pairs in parallel.
    for (int p2 = l; p2 < P.A * P.LMAX; p2 += blockDim.x) {
        const int ln = P.agent_lanes[p2];
        const int a = p2 / P.LMAX;
        if (ln >= 0 && p2 - a * P.LMAX < P.agent_nlane[a]) {
            int q = s.halt[ln];
            if (P.queue_cap >= 0 && q > P.queue_cap) q = P.queue_cap;
            atomicAdd(&qacc[a], q);
            atomicAdd(&wacc[a], s.hwait[ln]);
        }
    }
    __syncthreads();
    for (int a = l; a < P.A; a += blockDim.x) {
        const long long queue = qacc[a];
        const double wsum = (double)wacc[a];
        double r;
        if (P.objective == TSC_OBJ_QUEUE) r = (double)(-queue);
        else if (P.objective == TSC_OBJ_WAIT) r = -wsum;
        else r = (double)(-queue) - P.coef_wait * wsum;
        s.r[a] = r;
    }
    __syncthreads();
    if (l == 0) { double g = np_sum(s.r, P.A); s.r[P.A] = g; greward[e] = g; P.reward_acc[e] += g; }
    __syncthreads();
    for (int a = l; a < P.A; a += blockDim.x) {
        const double g = s.r[P.A];
        double out;
        if (!train_mode) {
            out = s.r[a];
        } else if (P.agent_kind == TSC_AGENT_GREEDY) {
            out = g;
        } else if (P.agent_kind == TSC_AGENT_GLOBAL) {
            out = P.realnet_scale ? g / (double)(P.A * 20) : g;
        } else {
            double cur = s.r[a];
            int deg = 0;
            for (int j = 0; j < P.NBR; ++j) {
                const int nb = P.nbr[a * P.NBR + j];
                if (nb < 0) break;
                cur += P.coop_gamma * s.r[nb];
                ++deg;
            }
            out = P.realnet_scale ? cur / (double)((1 + deg) * 20) : cur;
        }
        reward[(size_t)e * P.A + a] = out;
    }
    TSC_STAMP();
#ifdef TSC_ENV_PHASE_SUMS
    if (wsum) {
#pragma unroll
        for (int b_ = 0; b_ < 5; ++b_) P.dbg[64 + 2 * (size_t)P.E + 5 * (size_t)blockIdx.x + b_] = pacc_[b_];
    }
#endif
    if (stamp) P.dbg[63] = nstamp;
    if (P.dbg && threadIdx.x == 0) {
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        // end time in the low 48 bits, XCC id and HW_ID (cu / se / simd / wave slot) on top
        P.dbg[64 + 2 * blockIdx.x + 1] = (long long)(wall_clock64() & 0xFFFFFFFFFFFFll) | ((long long)(xcc & 0xF) << 60) | ((long long)(hwid & 0xFFF) << 48);
    }
#undef TSC_STAMP
}

}  // namespace

// ------------------------------------------------------------------------------------------------
struct tsc_env {
    EnvDev P;
    int device;
    hipStream_t stream;
    std::vector<void *> allocs;
    size_t smem;
    int threads;                    // workgroup size of step_kernel
    int kf;                         // flat-phase vehicles per thread (measurement knob TSC_ENV_KF)
    int spec;                       // 1: the scenario has the large_grid table dimensions -> specialised step_kernel (TSC_ENV_SPEC=0: off)
    uint32_t *d_seeds;
    std::vector<int> h_mode, h_sroute;      // host copies of the stream tables (tsc_env_set_stream_routes)
    int *order_buf = nullptr;       // tsc_env_set_block_order
    bool auto_threads;              // the workgroup size follows the number of instances resident on the device (pick_workgroup)
    int kf_default;
};

// Workgroup size / flat-phase width of the specialised step kernels for `n_resident` env instances on the device (this handle's
// and whatever shares the GPU with it); TSC_ENV_THREADS / TSC_ENV_KF override (measurement / test knobs).
static void pick_workgroup(tsc_env *h, int n_resident) {
    if (h->auto_threads) {
        int dev_cus = 256;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, h->device) == hipSuccess && prop.multiProcessorCount > 0) dev_cus = prop.multiProcessorCount;
        h->threads = 256; h->kf = h->kf_default;
        if (n_resident <= dev_cus) { h->threads = 1024; h->kf = 1; }
        else if (n_resident <= 2 * dev_cus) { h->threads = 512; h->kf = 2; }  // (Monaco, E = 512, saturated: 59.6 us with 2, 67.1 with 1)
    }
    if (const char *ev = getenv("TSC_ENV_THREADS")) {
        const int tv = atoi(ev);
        if (tv >= h->P.NLA && tv <= 1024 && tv % 64 == 0) h->threads = tv;
    }
    if (const char *ev = getenv("TSC_ENV_KF")) { const int kv = atoi(ev); h->kf = (kv == 2 || kv == 4) ? kv : 1; }
}

namespace tsc {
ProfState &prof() { static ProfState p; return p; }
}  // namespace tsc

extern "C" {

const char *tsc_last_error(void) { return tsc::err_buf(); }

int tsc_profile_enable(int32_t on) { tsc::prof().on = on != 0; tsc::prof().stride = on > 1 ? on : 1; return 0; }
int tsc_profile_select(uint64_t mask) { tsc::prof().only = mask ? mask : ~0ull; return 0; }

static int prof_fold() {
    tsc::ProfState &p = tsc::prof();
    for (auto &r : p.recs) {
        TSC_HIP(hipEventSynchronize(r.b));
        float ms = 0.f;
        TSC_HIP(hipEventElapsedTime(&ms, r.a, r.b));
        p.total_ms[r.id] += ms; p.count[r.id] += 1;
        p.pool.push_back(r.a); p.pool.push_back(r.b);
    }
    p.recs.clear();
    return 0;
}

int tsc_profile_reset(void) {
    if (prof_fold()) return 1;
    tsc::ProfState &p = tsc::prof();
    for (int i = 0; i < tsc::KID_COUNT; ++i) { p.total_ms[i] = 0; p.count[i] = 0; p.seq[i] = 0; }
    return 0;
}

int tsc_profile_read(int32_t kernel_id, double *total_ms, int64_t *count) {
    if (kernel_id < 0 || kernel_id >= tsc::KID_COUNT || !total_ms || !count) return tsc::fail("tsc_profile_read: bad arguments");
    if (prof_fold()) return 1;
    const tsc::ProfState &p = tsc::prof();
    const long long timed = p.count[kernel_id], all = p.seq[kernel_id];
    *total_ms = timed ? p.total_ms[kernel_id] / (double)timed * (double)all : 0.0;     // average x launches
    *count = all;
    return 0;
}

const char *tsc_profile_name(int32_t id) {
    static const char *names[] = {"env_step", "fc_gemm", "zx_gemm", "lstm_fwd", "head_fwd", "sample", "add_transition",
                                  "returns", "head_bwd", "lstm_bwd", "dwo_gemm", "dwh_gemm", "dwx_gemm", "dx1_gemm",
                                  "dw1_gemm", "grad_norm", "rmsprop", "transpose_wx", "fingerprint", "policy_fwd_fused",
                                  "iql_act", "iql_grad", "iql_reduce", "iql_sample", "iql_add", "iql_adam"};
    static_assert(sizeof(names) / sizeof(names[0]) == tsc::KID_COUNT, "one name per kernel id");
    return (id >= 0 && id < tsc::KID_COUNT) ? names[id] : "";
}

int tsc_version(void) { return 105; }      // 1.05: round 5 (tsc_env_set_greedy / tsc_env_greedy_actions); 1.04: tsc_env_counters, negative arrival = truncated trip

#define UP(field, T, src, count)                                                 \
    do {                                                                         \
        T *d_ = nullptr;                                                         \
        TSC_HIP(tsc::upload<T>(&d_, (const T *)(src), (size_t)(count)));         \
        h->allocs.push_back(d_);                                                 \
        P.field = d_;                                                            \
    } while (0)
#define ALLOC(field, T, count)                                                   \
    do {                                                                         \
        T *d_ = nullptr;                                                         \
        TSC_HIP(hipMalloc((void **)&d_, sizeof(T) * (size_t)(count)));           \
        TSC_HIP(hipMemset(d_, 0, sizeof(T) * (size_t)(count)));                  \
        h->allocs.push_back(d_);                                                 \
        P.field = d_;                                                            \
    } while (0)

int tsc_env_create(const tsc_scenario *sc, int32_t n_env, int32_t device, tsc_env **out) {
    if (!sc || !out || n_env <= 0) return tsc::fail("tsc_env_create: bad arguments");
    if (sc->n_lane > 1024) return tsc::fail("tsc_env_create: n_lane %d > 1024 unsupported", sc->n_lane);
    if (sc->n_route > 254) return tsc::fail("tsc_env_create: n_route %d > 254 unsupported", sc->n_route);
    TSC_HIP(hipSetDevice(device));
    tsc_env *h = new tsc_env();
    tsc::CreateGuard<tsc_env, tsc_env_destroy> guard(h);        // an error return below frees the handle and its buffers
    h->device = device;
    h->stream = nullptr;
    EnvDev &P = h->P;
    P.NL = sc->n_lane; P.NLP = (sc->n_lane + 63) / 64 * 64; P.NR = sc->n_route; P.A = sc->n_agent;
    P.NF = sc->n_flow; P.KMAX = sc->k_max; P.PMAX = sc->p_max; P.LMAX = sc->l_max; P.SMAX = sc->s_max;
    P.NBR = sc->nbr_max; P.E = n_env;
    P.ctrl = sc->control_interval_sec; P.yellow = sc->yellow_interval_sec; P.episode = sc->episode_length_sec;
    P.teleport = sc->teleport_sec; P.queue_cap = sc->queue_cap; P.objective = sc->objective;
    P.agent_kind = sc->agent_kind; P.realnet_scale = sc->realnet_scale;
    P.coop_gamma = sc->coop_gamma; P.norm_wave = sc->norm_wave; P.norm_wait = sc->norm_wait;
    P.clip_wave = sc->clip_wave; P.clip_wait = sc->clip_wait; P.coef_wait = sc->coef_wait;

    const int NL = P.NL, NR = P.NR, A = P.A;
    bool any_zip = false;                   // some lane has more than one feeder without a right-of-way table (zipper merge)
    // insertion streams: none declared -> every route is its own stream
    const bool streams = sc->n_stream > 0;
    const int NS = streams ? sc->n_stream : NR, KC = streams ? sc->k_choice : 1;
    P.NS = NS; P.KC = KC;
    if (NS > 254) return tsc::fail("tsc_env_create: n_stream %d > 254 unsupported", NS);
    if (streams && (!sc->stream_entry || !sc->stream_mode || !sc->stream_choice || KC < 1 || sc->n_interval < 1 || sc->choice_interval_sec < 1))
        return tsc::fail("tsc_env_create: incomplete stream tables");
    const int NI = streams ? sc->n_interval : 1;
    P.NI = NI; P.ilen = streams ? sc->choice_interval_sec : 1 << 30;
    auto stream_entry = [&](int s_) { return streams ? sc->stream_entry[s_] : sc->route_entry[s_]; };
    // Lanes that can ever hold a vehicle: the chains route entry lane -> mv_next[lane][route] -> ...  (the movement
    // table also has rows for the sibling lanes of every edge a route passes, which no vehicle of that route is
    // ever put on).  After load sorting they are a prefix; only they get a thread and LDS rows.
    std::vector<char> reach((size_t)NL, 0);
    {
        int nu = 0;
        for (int s_ = 0; s_ < NS; ++s_)
            for (int c = 0; c < KC * NI; ++c) {
                const int r = streams ? sc->stream_choice[((size_t)s_ * KC * NI + c) * 2] : s_;
                if (r < 0) continue;
                if (r >= NR) return tsc::fail("tsc_env_create: stream %d names route %d of %d", s_, r, NR);
                int l = stream_entry(s_);
                for (int hops = 0; l >= 0 && l < NL && hops <= 2 * NL; ++hops) {
                    reach[l] = 1;
                    if (l + 1 > nu) nu = l + 1;
                    int nx = sc->mv_next[(size_t)l * NR + r];
                    if (nx < -1 && sc->lane_sib && sc->lane_sib[l] >= 0 && sc->lane_sib[l] < NL &&
                        sc->mv_next[(size_t)sc->lane_sib[l] * NR + r] >= -1) nx = sc->lane_sib[l];     // rule 10: the vehicle moves over
                    l = nx;
                }
            }
        P.NU = nu < 1 ? 1 : nu;
        P.NLA = (P.NU + 63) / 64 * 64;
    }
    UP(lane_len, float, sc->lane_len, NL); UP(lane_vmax, float, sc->lane_vmax, NL);
    UP(lane_det, float, sc->lane_det_start, NL);
    UP(lane_node, int, sc->lane_node, NL);
    if (sc->lane_sib) {
        for (int l = 0; l < NL; ++l)
            if (sc->lane_sib[l] >= NL || sc->lane_sib[l] == l) return tsc::fail("tsc_env_create: lane %d names sibling lane %d of %d", l, sc->lane_sib[l], NL);
        UP(lane_sib, int, sc->lane_sib, NL);
    }
    {
        std::vector<int> up(sc->lane_up, sc->lane_up + (size_t)NL * kMaxUp);
        for (int &u : up) if (u >= 0 && (u >= NL || !reach[u])) u = -1;        // a feeder that is never occupied never sends
        UP(lane_up, int, up.data(), up.size());
    }
    if (NL > 0xFFD) return tsc::fail("tsc_env_create: n_lane %d > 4093 unsupported", NL);
    std::vector<int> mv((size_t)NL * NR);
    for (int i = 0; i < NL * NR; ++i) {
        const int nx = sc->mv_next[i], lk = sc->mv_link[i], pr = sc->mv_prio[i];
        const int yl = (sc->mv_yield[i] >= 0 && reach[sc->mv_yield[i]]) ? sc->mv_yield[i] : -1;   // an empty lane has no right of way to give
        if (lk > 62) return tsc::fail("tsc_env_create: signal link index %d > 62 unsupported", lk);
        const unsigned a = nx == -1 ? 0xFFFu : nx < -1 ? 0xFFEu : (unsigned)nx;
        const unsigned b = lk < 0 ? 63u : (unsigned)lk;
        const unsigned c = yl < 0 ? 0xFFFu : (unsigned)yl;
        mv[i] = (int)(a | (b << 12) | (c << 18) | ((pr ? 1u : 0u) << 30));
        if (mv_tl(mv[i]) != (nx < -1 ? -2 : nx) || mv_k(mv[i]) != (lk < 0 ? -1 : lk) || mv_yield(mv[i]) != yl)
            return tsc::fail("tsc_env_create: movement table overflow");
    }
    UP(mv, int, mv.data(), NL * NR);
    {
        std::vector<uint8_t> zp(((size_t)NL * NR + 3) / 4 * 4, 0);
        for (int i = 0; i < NL * NR; ++i) {
            const int z = sc->mv_zip[i], rank = z & 0xFF, cnt = z >> 8;
            if (rank > 15 || cnt > 15) return tsc::fail("tsc_env_create: zipper slot overflow");
            zp[i] = (uint8_t)(rank | (cnt << 4));
        }
        UP(zip, uint8_t, zp.data(), zp.size());
        for (uint8_t v : zp) if (v >> 4 > 1) any_zip = true;
        // the merge arbitration packs a lane's feeders into bytes (Smem::up4, 0xFF = none): a feeder index >= 255 would be
        // left out of the winner search and wait for the teleport
        if (any_zip && NL > 255)
            return tsc::fail("tsc_env_create: zipper merges need lane indices < 255 (feeders are bytes on the device), the scenario has %d lanes", NL);
    }
    {
        std::vector<int> se(NS);
        for (int s_ = 0; s_ < NS; ++s_) se[s_] = stream_entry(s_);
        UP(route_entry, int, se.data(), NS);
    }
    P.sroute = nullptr; P.smode = nullptr; P.schoice = nullptr; P.sorigin = nullptr; P.iroute = nullptr;
    if (streams) {
        std::vector<int> sr(NS);
        bool any_origin = false, identity = NS == NR;
        for (int s_ = 0; s_ < NS; ++s_) {
            const int md = sc->stream_mode[s_];
            if (md < 0 || md > 2) return tsc::fail("tsc_env_create: stream %d has mode %d", s_, md);
            sr[s_] = sc->stream_choice[(size_t)s_ * NI * KC * 2];
            if (sr[s_] < 0) return tsc::fail("tsc_env_create: stream %d has no route", s_);
            if (md != 0 || sr[s_] != s_) identity = false;
            if (sc->stream_origin && sc->stream_origin[s_] != 0.0f) any_origin = true;
            if (sc->stream_limit && sc->stream_limit[s_] < sc->lane_len[stream_entry(s_)]) any_origin = true;
        }
        h->h_mode.assign(sc->stream_mode, sc->stream_mode + NS); h->h_sroute = sr;
        if (!identity) {                              // fixed one-to-one streams keep the round-2 fast path (sroute == null)
            UP(sroute, int, sr.data(), NS);
            UP(smode, int, sc->stream_mode, NS);
            UP(schoice, int, sc->stream_choice, (size_t)NS * NI * KC * 2);
            ALLOC(iroute, int, (size_t)n_env * NS);
            std::vector<int> ir((size_t)n_env * NS);
            for (size_t i = 0; i < ir.size(); ++i) ir[i] = sr[i % NS];
            TSC_HIP(hipMemcpy(P.iroute, ir.data(), sizeof(int) * ir.size(), hipMemcpyHostToDevice));
        }
        if (any_origin) {
            std::vector<float> so((size_t)NS * 2);
            for (int s_ = 0; s_ < NS; ++s_) {
                so[2 * s_] = sc->stream_origin ? sc->stream_origin[s_] : 0.0f;
                so[2 * s_ + 1] = sc->stream_limit ? sc->stream_limit[s_] : INFINITY;
            }
            UP(sorigin, float, so.data(), so.size());
        }
    }
    // flows sorted by stream (stable) + CSR
    std::vector<int> fl; std::vector<int> ptr(NS + 1, 0);
    for (int r = 0; r < NS; ++r) {
        ptr[r] = (int)fl.size() / 4;
        for (int f = 0; f < sc->n_flow; ++f)
            if (sc->flows[f * 4 + 3] == r) fl.insert(fl.end(), sc->flows + f * 4, sc->flows + f * 4 + 4);
    }
    ptr[NS] = (int)fl.size() / 4;
    UP(flows, int, fl.data(), fl.size());
    UP(flow_ptr, int, ptr.data(), NS + 1);
    {   // per-lane entry routes (<= 2) and per-route emission table (MICROSIM_SPEC.md, rule 6)
        std::vector<int> lr((size_t)NL * kMaxEntry, -1);
        for (int r = 0; r < NS; ++r) {
            const int l = stream_entry(r);
            if (l < 0 || l >= NL) return tsc::fail("tsc_env_create: stream %d has no entry lane", r);
            int q = 0;
            while (q < kMaxEntry && lr[l * kMaxEntry + q] >= 0) ++q;
            if (q == kMaxEntry) return tsc::fail("tsc_env_create: more than %d routes enter lane %d", kMaxEntry, l);
            lr[l * kMaxEntry + q] = r;
        }
        std::vector<uint32_t> lrp((size_t)NL * 2, 0xFFFFFFFFu);         // one byte per route (n_route <= 255 checked above)
        for (int l = 0; l < NL; ++l)
            for (int q = 0; q < kMaxEntry; ++q)
                if (lr[l * kMaxEntry + q] >= 0)
                    lrp[l * 2 + q / 4] = (lrp[l * 2 + q / 4] & ~(0xFFu << (8 * (q % 4)))) | ((uint32_t)lr[l * kMaxEntry + q] << (8 * (q % 4)));
        UP(lane_routes, uint32_t, lrp.data(), lrp.size());
        P.emit_len = sc->episode_length_sec + 64;
        std::vector<uint8_t> em((size_t)NS * P.emit_len, 0);
        for (int f = 0; f < sc->n_flow; ++f) {
            const long long b = sc->flows[f * 4], en = sc->flows[f * 4 + 1], vph = sc->flows[f * 4 + 2];
            const int r = sc->flows[f * 4 + 3];
            if (r < 0 || r >= NS) return tsc::fail("tsc_env_create: flow %d names stream %d of %d", f, r, NS);
            for (long long t = b; t < en && t < P.emit_len; ++t) {
                const long long tau = t - b;
                const long long c = (((tau + 1) * vph + 3599) / 3600) - ((tau * vph + 3599) / 3600);
                const long long v = em[(size_t)r * P.emit_len + t] + c;
                if (v > 255) return tsc::fail("tsc_env_create: flow %d emits more than 255 vehicles per second", f);
                em[(size_t)r * P.emit_len + t] = (uint8_t)v;
            }
        }
        UP(emit_tab, uint8_t, em.data(), em.size());
    }
    if (sc->control_interval_sec > 8) return tsc::fail("tsc_env_create: control interval > 8 s unsupported");
    UP(agent_lanes, int, sc->agent_lanes, A * P.LMAX);
    UP(agent_nlane, int, sc->agent_nlane, A); UP(agent_nlink, int, sc->agent_nlink, A);
    UP(agent_nphase, int, sc->agent_nphase, A);
    UP(green_tab, uint8_t, sc->green_tab, (size_t)A * P.PMAX * P.KMAX);
    UP(yellow_tab, uint8_t, sc->yellow_tab, (size_t)A * P.PMAX * P.PMAX * P.KMAX);
    UP(nbr, int, sc->nbr, A * P.NBR);
    UP(obs_kind, int, sc->obs_kind, A * P.SMAX); UP(obs_src, int, sc->obs_src, A * P.SMAX);
    {
        std::vector<int> ks((size_t)A * P.SMAX);
        for (size_t i = 0; i < ks.size(); ++i) {
            if (sc->obs_src[i] < 0 || sc->obs_src[i] > 0xFFFF) { ks[i] = 0; continue; }
            ks[i] = (sc->obs_kind[i] << 16) | sc->obs_src[i];
        }
        UP(obs_ks, int, ks.data(), ks.size());
    }

    const size_t slots = (size_t)n_env * kCap * P.NLP;
    ALLOC(S, float4, slots);
    ALLOC(N, int, (size_t)n_env * P.NLP);
    ALLOC(pending, int, (size_t)n_env * NS); ALLOC(serial, int, (size_t)n_env * NS);
    ALLOC(tsec, int, n_env); ALLOC(seed, uint32_t, n_env);
    ALLOC(prev_action, int, (size_t)n_env * A);
    ALLOC(fp, float, (size_t)n_env * A * P.PMAX);
    ALLOC(arrived, unsigned long long, n_env);
    ALLOC(teleported, unsigned long long, n_env);
    ALLOC(reward_acc, double, n_env);
    ALLOC(n_trips, int, n_env); ALLOC(live_acc, unsigned long long, n_env);
    {
        std::vector<float> org((size_t)NL, 0.0f);
        if (sc->lane_origin) org.assign(sc->lane_origin, sc->lane_origin + NL);
        UP(lane_origin, float, org.data(), NL);
    }
    P.rec = 0; P.trip_cap = 0; P.R0 = P.R1 = nullptr; P.rec_int = nullptr; P.rec_speed = nullptr; P.rec_queue = nullptr;
    P.trips = nullptr;
    P.fp_bound = nullptr;
    P.dbg = nullptr;
    P.order = nullptr;
    TSC_HIP(hipMalloc((void **)&h->d_seeds, sizeof(uint32_t) * n_env));
    h->allocs.push_back(h->d_seeds);
    {   // phase A1 (helper threads) unless switched off for A/B measurements
        const char *ev = getenv("TSC_ENV_HELP");
        P.help = (ev && ev[0] == '0') ? 0 : 1;
    }
    if (kMaxCross * P.NLA < P.NLP || kMaxCross * P.NLA < 2 * P.A)
        return tsc::fail("tsc_env_create: too few live lanes (%d of %d) for the observation scratch", P.NU, P.NL);
    h->smem = smem_bytes(P);
    if (h->smem > 160 * 1024) return tsc::fail("tsc_env_create: LDS need %zu B > 160 KiB", h->smem);
    h->threads = (P.help && P.NLA < 256) ? 256 : P.NLA;
    // the reference's large_grid / Monaco get the instantiation with compile-time table dimensions (TSC_ENV_SPEC=0: off)
    h->spec = 0;
    for (int k = 1; k < 3; ++k) {
        const SpecDims &D = kSpec[k];
        // (NU: the scenario's live lanes are a prefix of the instantiation's; the lanes in between stay empty)
        if (P.NS == P.NR && P.NLP == D.NLP && P.NLA == D.NLA && P.NU <= D.NU && D.NU <= P.NL && P.NR == D.NR && P.A == D.A && P.KMAX == D.KMAX && P.PMAX == D.PMAX &&
            P.LMAX == D.LMAX && P.NBR == D.NBR && P.ctrl == D.ctrl && P.yellow == D.yellow && P.teleport == D.teleport &&
            (k != 1 || (!any_zip && !P.sorigin)) && !P.sroute) { h->spec = k; P.NU = D.NU; h->smem = smem_bytes(P); }
    }
    if (const char *ev = getenv("TSC_ENV_SPEC")) if (!atoi(ev)) h->spec = 0;
    // vehicles per thread and super-round of the flat phase (TSC_ENV_KF = 1 / 2 / 4 for A/B runs).  With runtime dimensions
    // 1 is best (296 M env-steps/s; 2: 294, 4: 284 -- fewer barriers do not pay for the registers); the specialised kernel
    // has the registers for 2 (env step 11.5 -> 10.7 ms per rollout; 4 spills: 11.4; round 5, rule-10 traffic, episode average:
    // 3 -- 128 VGPRs, 8 spilled dwords, bit-exact -- 94.9 / 95.3 us per control step against 92.7 / 92.8 with 2)
    h->kf = h->spec == 1 ? 2 : 1;                            // (Monaco, spec 2: 330 M env-steps/s with 1, 323 M with 2)
    if (const char *ev = getenv("TSC_ENV_KF")) { const int kv = atoi(ev); h->kf = (kv == 2 || kv == 4) ? kv : 1; }
    if (h->spec && h->kf == 4) h->spec = 0;                  // (no specialised instantiation of the 4-wide variant)
#define TSC_ATTR(MT, KF, SP) TSC_HIP(hipFuncSetAttribute((const void *)step_kernel<MT, true, false, KF, SP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->smem))
    TSC_ATTR(256, 1, 1); TSC_ATTR(256, 2, 1); TSC_ATTR(256, 1, 2); TSC_ATTR(256, 2, 2);
    TSC_ATTR(512, 1, 1); TSC_ATTR(512, 2, 1); TSC_ATTR(512, 1, 2); TSC_ATTR(512, 2, 2);
    TSC_ATTR(1024, 1, 1); TSC_ATTR(1024, 2, 1); TSC_ATTR(1024, 1, 2); TSC_ATTR(1024, 2, 2);
#undef TSC_ATTR
    TSC_HIP(hipFuncSetAttribute((const void *)step_kernel<256, true, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->smem));
    TSC_HIP(hipFuncSetAttribute((const void *)step_kernel<256, true, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->smem));
    // Workgroup size of the specialised kernels: a CU holds 16 wavefronts of this kernel (128 VGPRs), i.e. four workgroups of 256
    // threads.  With fewer instances than that the flat phase -- a latency chain per wavefront -- is spread over more wavefronts of
    // the same instance instead of leaving the slots idle (sim only, saturated large_grid: E = 256: 63.0 -> 52.5 us per control
    // step with 1024 threads, E = 512: 70.9 -> 64.8 us with 512; Monaco 69.6 -> 59.9 / 77.9 -> 71.5; E = 1024 wants 256).
    // TSC_ENV_THREADS overrides (parity tests run every size).
    // What counts is how many instances share the DEVICE, not this handle's own: tsc_env_set_resident_instances (half-batches on
    // separate streams, ranks sharing a GPU).
    h->auto_threads = h->spec && P.help && h->threads == 256;
    h->kf_default = h->kf;
    pick_workgroup(h, n_env);
    TSC_HIP(hipFuncSetAttribute((const void *)step_kernel<256, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->smem));
    TSC_HIP(hipFuncSetAttribute((const void *)step_kernel<1024, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->smem));
    TSC_HIP(hipFuncSetAttribute((const void *)step_kernel<256, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->smem));
    TSC_HIP(hipFuncSetAttribute((const void *)step_kernel<1024, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->smem));
    TSC_HIP(hipFuncSetAttribute((const void *)reset_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->smem));
    *out = guard.release();
    return 0;
}

int tsc_env_record(tsc_env *h, int32_t enable, int32_t trip_cap) {
    if (!h || trip_cap < 0) return tsc::fail("tsc_env_record: bad arguments");
    EnvDev &P = h->P;
    TSC_HIP(hipStreamSynchronize(h->stream));
    if (enable && !P.R0) {
        const size_t slots = (size_t)P.E * kCap * P.NLP;
        ALLOC(R0, uint32_t, slots); ALLOC(R1, uint32_t, slots);
        ALLOC(rec_int, long long, (size_t)P.E * 8 * 4); ALLOC(rec_speed, double, (size_t)P.E * 8);
        ALLOC(rec_queue, int, (size_t)P.E * 8 * P.A * P.LMAX);
        P.trip_cap = trip_cap > 0 ? trip_cap : 8192;
        ALLOC(trips, int, (size_t)P.E * P.trip_cap * 6);
    }
    P.rec = enable ? 1 : 0;
    h->smem = smem_bytes(P);
    if (h->smem > 160 * 1024) return tsc::fail("tsc_env_record: LDS need %zu B > 160 KiB", h->smem);
    TSC_HIP(hipFuncSetAttribute((const void *)step_kernel<256, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->smem));
    TSC_HIP(hipFuncSetAttribute((const void *)step_kernel<1024, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->smem));
    TSC_HIP(hipFuncSetAttribute((const void *)step_kernel<256, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->smem));
    TSC_HIP(hipFuncSetAttribute((const void *)step_kernel<256, true, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->smem));
    TSC_HIP(hipFuncSetAttribute((const void *)step_kernel<256, true, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->smem));
    TSC_HIP(hipFuncSetAttribute((const void *)step_kernel<1024, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->smem));
    TSC_HIP(hipFuncSetAttribute((const void *)step_kernel<256, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->smem));
    TSC_HIP(hipFuncSetAttribute((const void *)step_kernel<1024, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->smem));
    TSC_HIP(hipFuncSetAttribute((const void *)reset_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->smem));
    return 0;
}

int tsc_env_read_record(tsc_env *h, int64_t *ints_host, double *speed_host, int32_t *queue_host) {
    if (!h || !h->P.rec_int || !ints_host || !speed_host || !queue_host) return tsc::fail("tsc_env_read_record: recording is off");
    const EnvDev &P = h->P;
    TSC_HIP(hipStreamSynchronize(h->stream));
    TSC_HIP(hipMemcpy(ints_host, P.rec_int, sizeof(long long) * (size_t)P.E * 8 * 4, hipMemcpyDeviceToHost));
    TSC_HIP(hipMemcpy(speed_host, P.rec_speed, sizeof(double) * (size_t)P.E * 8, hipMemcpyDeviceToHost));
    TSC_HIP(hipMemcpy(queue_host, P.rec_queue, sizeof(int) * (size_t)P.E * 8 * P.A * P.LMAX, hipMemcpyDeviceToHost));
    return 0;
}

int tsc_env_read_trips(tsc_env *h, int32_t e, int32_t *trips_host, int32_t max_trips, int32_t *count) {
    if (!h || !h->P.trips || e < 0 || e >= h->P.E || !trips_host || !count) return tsc::fail("tsc_env_read_trips: recording is off");
    const EnvDev &P = h->P;
    TSC_HIP(hipStreamSynchronize(h->stream));
    int n = 0;
    TSC_HIP(hipMemcpy(&n, P.n_trips + e, sizeof(int), hipMemcpyDeviceToHost));
    *count = n;
    if (n > P.trip_cap) n = P.trip_cap;
    if (n > max_trips) n = max_trips;
    if (n > 0) TSC_HIP(hipMemcpy(trips_host, P.trips + (size_t)e * P.trip_cap * 6, sizeof(int) * (size_t)n * 6, hipMemcpyDeviceToHost));
    return 0;
}

int tsc_env_counters(tsc_env *h, uint64_t *arrived_host, uint64_t *teleported_host) {
    if (!h || (!arrived_host && !teleported_host)) return tsc::fail("tsc_env_counters: bad arguments");
    TSC_HIP(hipStreamSynchronize(h->stream));
    if (arrived_host) TSC_HIP(hipMemcpy(arrived_host, h->P.arrived, sizeof(uint64_t) * h->P.E, hipMemcpyDeviceToHost));
    if (teleported_host) TSC_HIP(hipMemcpy(teleported_host, h->P.teleported, sizeof(uint64_t) * h->P.E, hipMemcpyDeviceToHost));
    return 0;
}

int tsc_env_live_sum(tsc_env *h, double *sum_host, int32_t reset) {
    if (!h || !sum_host) return tsc::fail("tsc_env_live_sum: bad arguments");
    TSC_HIP(hipStreamSynchronize(h->stream));
    std::vector<unsigned long long> acc(h->P.E);
    TSC_HIP(hipMemcpy(acc.data(), h->P.live_acc, sizeof(unsigned long long) * h->P.E, hipMemcpyDeviceToHost));
    double t = 0;
    for (unsigned long long v : acc) t += (double)v;
    *sum_host = t;
    if (reset) TSC_HIP(hipMemset(h->P.live_acc, 0, sizeof(unsigned long long) * h->P.E));
    return 0;
}

int tsc_env_destroy(tsc_env *h) {
    if (!h) return 0;
    (void)hipSetDevice(h->device);
    for (void *p : h->allocs) (void)hipFree(p);
    delete h;
    return 0;
}

int tsc_env_set_resident_instances(tsc_env *h, int32_t n_resident) {
    if (!h || n_resident <= 0) return tsc::fail("tsc_env_set_resident_instances: bad arguments");
    pick_workgroup(h, n_resident > h->P.E ? n_resident : h->P.E);
    return 0;
}

int tsc_env_set_stream(tsc_env *h, void *hip_stream) {
    if (!h) return tsc::fail("null handle");
    h->stream = (hipStream_t)hip_stream;
    return 0;
}

int tsc_env_reset(tsc_env *h, const uint32_t *seeds_host, float *obs_dev) {
    if (!h || !seeds_host || !obs_dev) return tsc::fail("tsc_env_reset: bad arguments");
    h->P.fp_bound = nullptr;                     // reset(): fingerprints <- uniform policy (envs/env.py:556-557)
    TSC_HIP(hipMemcpyAsync(h->d_seeds, seeds_host, sizeof(uint32_t) * h->P.E, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(reset_kernel, dim3(h->P.E), dim3(h->P.NLP), h->smem, h->stream, h->P, h->d_seeds, obs_dev);
    TSC_HIP(hipGetLastError());
    TSC_HIP(hipStreamSynchronize(h->stream));          // seeds_host may be reused by the caller
    return 0;
}

int tsc_env_set_stream_routes(tsc_env *h, const int32_t *routes_host) {
    if (!h || !routes_host) return tsc::fail("tsc_env_set_stream_routes: bad arguments");
    const EnvDev &P = h->P;
    if (!P.iroute) return tsc::fail("tsc_env_set_stream_routes: the scenario has no per-episode stream routes");
    std::vector<int> ir((size_t)P.E * P.NS);
    for (int e = 0; e < P.E; ++e)
        for (int s_ = 0; s_ < P.NS; ++s_) {
            int r = h->h_sroute[s_];
            if (h->h_mode[s_] == 2) {
                r = routes_host[(size_t)e * P.NS + s_];
                if (r < 0 || r >= P.NR) return tsc::fail("tsc_env_set_stream_routes: instance %d stream %d: route %d of %d", e, s_, r, P.NR);
            }
            ir[(size_t)e * P.NS + s_] = r;
        }
    TSC_HIP(hipMemcpyAsync(P.iroute, ir.data(), sizeof(int) * ir.size(), hipMemcpyHostToDevice, h->stream));
    TSC_HIP(hipStreamSynchronize(h->stream));
    return 0;
}

int tsc_env_bind_fingerprint(tsc_env *h, const float *pi_dev) {
    if (!h) return tsc::fail("null handle");
    h->P.fp_bound = pi_dev;                      // null: back to the copied fingerprints
    return 0;
}

int tsc_env_reward_sum(tsc_env *h, double *sum_host, int32_t reset) {
    if (!h || !sum_host) return tsc::fail("tsc_env_reward_sum: bad arguments");
    TSC_HIP(hipStreamSynchronize(h->stream));
    std::vector<double> acc(h->P.E);
    TSC_HIP(hipMemcpy(acc.data(), h->P.reward_acc, sizeof(double) * h->P.E, hipMemcpyDeviceToHost));
    double t = 0;
    for (double v : acc) t += v;
    *sum_host = t;
    if (reset) TSC_HIP(hipMemset(h->P.reward_acc, 0, sizeof(double) * h->P.E));
    return 0;
}

int tsc_env_set_fingerprint(tsc_env *h, const float *pi_dev) {
    if (!h || !pi_dev) return tsc::fail("tsc_env_set_fingerprint: bad arguments");
    h->P.fp_bound = nullptr;
    size_t tot = (size_t)h->P.E * h->P.A * h->P.PMAX;
    tsc::ProfScope ps(tsc::KID_FINGERPRINT, h->stream);
    hipLaunchKernelGGL(fingerprint_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream, h->P, pi_dev);
    TSC_HIP(hipGetLastError());
    return 0;
}

int tsc_env_set_greedy(tsc_env *h, int32_t n_cand_max, int32_t n_term_max, const int32_t *n_cand, const int32_t *term,
                       const int32_t *cand_action) {
    if (!h || !n_cand || !term || !cand_action || n_cand_max <= 0 || n_term_max <= 0)
        return tsc::fail("tsc_env_set_greedy: bad arguments");
    EnvDev &P = h->P;
    std::vector<int8_t> t8((size_t)P.A * n_cand_max * n_term_max);
    for (int a = 0; a < P.A; ++a) {
        if (n_cand[a] <= 0 || n_cand[a] > n_cand_max) return tsc::fail("tsc_env_set_greedy: agent %d has %d candidates of %d", a, n_cand[a], n_cand_max);
        for (int c = 0; c < n_cand_max; ++c) {
            if (c < n_cand[a] && (cand_action[a * n_cand_max + c] < 0 || cand_action[a * n_cand_max + c] >= P.PMAX))
                return tsc::fail("tsc_env_set_greedy: agent %d candidate %d: action %d of %d", a, c, cand_action[a * n_cand_max + c], P.PMAX);
            for (int k = 0; k < n_term_max; ++k) {
                const int j = term[((size_t)a * n_cand_max + c) * n_term_max + k];
                if (j >= P.SMAX || j > 127) return tsc::fail("tsc_env_set_greedy: agent %d candidate %d: observation index %d of %d", a, c, j, P.SMAX);
                t8[((size_t)a * n_cand_max + c) * n_term_max + k] = (int8_t)(j < 0 ? -1 : j);
            }
        }
    }
    (void)hipSetDevice(h->device);
    TSC_HIP(hipStreamSynchronize(h->stream));                  // a running greedy_kernel may still read the old tables
    for (const void *old : {(const void *)P.g_ncand, (const void *)P.g_term, (const void *)P.g_action}) {      // a second call replaces them
        if (!old) continue;
        for (auto it = h->allocs.begin(); it != h->allocs.end(); ++it)
            if (*it == old) { (void)hipFree(*it); h->allocs.erase(it); break; }
    }
    P.g_ncand = nullptr; P.g_term = nullptr; P.g_action = nullptr;
    UP(g_ncand, int, n_cand, P.A);
    UP(g_term, int8_t, t8.data(), t8.size());
    UP(g_action, int, cand_action, (size_t)P.A * n_cand_max);
    P.GC = n_cand_max; P.GT = n_term_max;
    return 0;
}

int tsc_env_greedy_actions(tsc_env *h, const float *obs_dev, int32_t *action_dev) {
    if (!h || !obs_dev || !action_dev) return tsc::fail("tsc_env_greedy_actions: bad arguments");
    if (!h->P.g_ncand) return tsc::fail("tsc_env_greedy_actions: no controller tables (tsc_env_set_greedy)");
    const int tot = h->P.E * h->P.A;
    hipLaunchKernelGGL(greedy_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream, h->P, obs_dev, action_dev);
    TSC_HIP(hipGetLastError());
    return 0;
}

int tsc_env_step(tsc_env *h, const int32_t *action_dev, float *obs_dev, double *reward_dev,
                 double *global_reward_dev, uint8_t *done_dev, int32_t train_mode) {
    if (!h || !action_dev || !obs_dev || !reward_dev || !global_reward_dev || !done_dev)
        return tsc::fail("tsc_env_step: bad arguments");
    tsc::ProfScope ps(tsc::KID_ENV_STEP, h->stream);
#define TSC_STEP(MAXT, HELP)                                                                                       \
    hipLaunchKernelGGL((step_kernel<MAXT, HELP>), dim3(h->P.E), dim3(h->threads), h->smem, h->stream, h->P, action_dev, \
                       obs_dev, reward_dev, global_reward_dev, done_dev, (int)train_mode)
#define TSC_STEP_REC(MAXT)                                                                                         \
    hipLaunchKernelGGL((step_kernel<MAXT, false, true>), dim3(h->P.E), dim3(h->threads), h->smem, h->stream, h->P, action_dev, \
                       obs_dev, reward_dev, global_reward_dev, done_dev, (int)train_mode)
    if (h->P.rec) { if (h->threads <= 256) TSC_STEP_REC(256); else TSC_STEP_REC(1024); }
#define TSC_STEP_KF(KF)                                                                                           \
    hipLaunchKernelGGL((step_kernel<256, true, false, KF>), dim3(h->P.E), dim3(h->threads), h->smem, h->stream, h->P, action_dev, \
                       obs_dev, reward_dev, global_reward_dev, done_dev, (int)train_mode)
#define TSC_STEP_SPEC(MT, KF, SP)                                                                                   \
    hipLaunchKernelGGL((step_kernel<MT, true, false, KF, SP>), dim3(h->P.E), dim3(h->threads), h->smem, h->stream, h->P, action_dev, \
                       obs_dev, reward_dev, global_reward_dev, done_dev, (int)train_mode)
    // specialised instantiations: workgroup size 256 (four workgroups per CU: E > 512), 512 or 1024 (fewer instances than
    // workgroup slots: the flat phase spreads over more wavefronts instead of leaving the CU idle)
    else if (h->P.help && h->spec && (h->kf == 1 || h->kf == 2) && (h->threads == 256 || h->threads == 512 || h->threads == 1024)) {
        const int key = h->threads * 100 + h->kf * 10 + h->spec;
        switch (key) {
            case 25611: TSC_STEP_SPEC(256, 1, 1); break;   case 25621: TSC_STEP_SPEC(256, 2, 1); break;
            case 25612: TSC_STEP_SPEC(256, 1, 2); break;   case 25622: TSC_STEP_SPEC(256, 2, 2); break;
            case 51211: TSC_STEP_SPEC(512, 1, 1); break;   case 51221: TSC_STEP_SPEC(512, 2, 1); break;
            case 51212: TSC_STEP_SPEC(512, 1, 2); break;   case 51222: TSC_STEP_SPEC(512, 2, 2); break;
            case 102411: TSC_STEP_SPEC(1024, 1, 1); break; case 102421: TSC_STEP_SPEC(1024, 2, 1); break;
            case 102412: TSC_STEP_SPEC(1024, 1, 2); break; case 102422: TSC_STEP_SPEC(1024, 2, 2); break;
            default: return tsc::fail("tsc_env_step: no instantiation for %d threads, kf %d, spec %d", h->threads, h->kf, h->spec);
        }
    }
#undef TSC_STEP_SPEC
    else if (h->threads <= 256 && h->P.help && h->kf == 1) TSC_STEP_KF(1);
    else if (h->threads <= 256 && h->P.help && h->kf == 2) TSC_STEP_KF(2);
#undef TSC_STEP_KF
    else if (h->threads <= 256) { if (h->P.help) TSC_STEP(256, true); else TSC_STEP(256, false); }
    else { if (h->P.help) TSC_STEP(1024, true); else TSC_STEP(1024, false); }
#undef TSC_STEP_REC
#undef TSC_STEP
    TSC_HIP(hipGetLastError());
    return 0;
}

int tsc_env_get_state(tsc_env *h, int32_t e, int32_t *n, float *x, float *v, float *sf, int32_t *w, int32_t *r,
                      int32_t *pending, int32_t *serial, int32_t *time_sec) {
    if (!h || e < 0 || e >= h->P.E) return tsc::fail("tsc_env_get_state: bad arguments");
    const EnvDev &P = h->P;
    TSC_HIP(hipStreamSynchronize(h->stream));
    const size_t slab = (size_t)kCap * P.NLP;
    std::vector<float4> hs4(slab);
    std::vector<int> hn(P.NLP);
    TSC_HIP(hipMemcpy(hs4.data(), P.S + e * slab, slab * sizeof(float4), hipMemcpyDeviceToHost));
    TSC_HIP(hipMemcpy(hn.data(), P.N + (size_t)e * P.NLP, P.NLP * 4, hipMemcpyDeviceToHost));
    for (int l = 0; l < P.NL; ++l) {
        n[l] = hn[l];
        for (int i = 0; i < kCap; ++i) {
            const bool live = i < hn[l];
            const size_t s = (size_t)vslot(i, l, P.NLP), d = (size_t)l * kCap + i;
            uint32_t mb = 0u;
            if (live) memcpy(&mb, &hs4[s].w, 4);
            x[d] = live ? hs4[s].x : 0.0f; v[d] = live ? hs4[s].y : 0.0f; sf[d] = live ? hs4[s].z : 0.0f;
            w[d] = (int)(mb & 0xFFFFu); r[d] = (int)(mb >> 16);
        }
    }
    if (pending) TSC_HIP(hipMemcpy(pending, P.pending + (size_t)e * P.NS, P.NS * 4, hipMemcpyDeviceToHost));
    if (serial) TSC_HIP(hipMemcpy(serial, P.serial + (size_t)e * P.NS, P.NS * 4, hipMemcpyDeviceToHost));
    if (time_sec) TSC_HIP(hipMemcpy(time_sec, P.tsec + e, 4, hipMemcpyDeviceToHost));
    return 0;
}

int tsc_env_debug_clock(tsc_env *h, int32_t enable, int64_t *stamps64_host) {
    if (!h) return tsc::fail("null handle");
    TSC_HIP(hipStreamSynchronize(h->stream));
    if (enable && !h->P.dbg) {
        long long *d = nullptr;
        TSC_HIP(hipMalloc((void **)&d, (64 + 7 * (size_t)h->P.E) * sizeof(long long)));
        TSC_HIP(hipMemset(d, 0, (64 + 7 * (size_t)h->P.E) * sizeof(long long)));
        h->allocs.push_back(d);
        h->P.dbg = d;
    }
    if (stamps64_host && h->P.dbg)
        TSC_HIP(hipMemcpy(stamps64_host, h->P.dbg, (enable == 3 ? 64 + 7 * (size_t)h->P.E : enable == 2 ? 64 + 2 * (size_t)h->P.E : 64) * sizeof(long long), hipMemcpyDeviceToHost));
    return 0;
}

int tsc_env_vehicle_counts(tsc_env *h, int32_t *counts_host) {
    if (!h || !counts_host) return tsc::fail("tsc_env_vehicle_counts: bad arguments");
    const EnvDev &P = h->P;
    TSC_HIP(hipStreamSynchronize(h->stream));
    std::vector<int> hn((size_t)P.E * P.NLP);
    TSC_HIP(hipMemcpy(hn.data(), P.N, hn.size() * 4, hipMemcpyDeviceToHost));
    for (int e = 0; e < P.E; ++e) {
        int tot = 0;
        for (int l = 0; l < P.NL; ++l) tot += hn[(size_t)e * P.NLP + l];
        counts_host[e] = tot;
    }
    return 0;
}

int tsc_env_set_block_order(tsc_env *h, const int32_t *order_host) {
    if (!h) return tsc::fail("tsc_env_set_block_order: null handle");
    TSC_HIP(hipStreamSynchronize(h->stream));
    if (!order_host) { h->P.order = nullptr; return 0; }
    std::vector<char> seen(h->P.E, 0);
    for (int b = 0; b < h->P.E; ++b) {
        const int e = order_host[b];
        if (e < 0 || e >= h->P.E || seen[e]) return tsc::fail("tsc_env_set_block_order: not a permutation of the %d instances", h->P.E);
        seen[e] = 1;
    }
    if (!h->order_buf) {
        TSC_HIP(hipMalloc((void **)&h->order_buf, sizeof(int) * (size_t)h->P.E));
        h->allocs.push_back(h->order_buf);
    }
    TSC_HIP(hipMemcpy(h->order_buf, order_host, sizeof(int) * (size_t)h->P.E, hipMemcpyHostToDevice));
    h->P.order = h->order_buf;
    return 0;
}

int tsc_env_live_vehicles(tsc_env *h, double *mean_live) {
    if (!h || !mean_live) return tsc::fail("tsc_env_live_vehicles: bad arguments");
    const EnvDev &P = h->P;
    TSC_HIP(hipStreamSynchronize(h->stream));
    std::vector<int> hn((size_t)P.E * P.NLP);
    TSC_HIP(hipMemcpy(hn.data(), P.N, hn.size() * 4, hipMemcpyDeviceToHost));
    double tot = 0;
    for (int e = 0; e < P.E; ++e)
        for (int l = 0; l < P.NL; ++l) tot += hn[(size_t)e * P.NLP + l];
    *mean_live = tot / P.E;
    return 0;
}

}  // extern "C"
