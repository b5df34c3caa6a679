/*
 * tsc.h -- C ABI of the MI355X-native traffic-signal-control hot path.
 *
 * The reference (cts198859/deeprl_signal_control) has no FFI: its boundary is two
 * Python duck-types (env + model) plus the TraCI wire protocol.
 * This header is what a binding for that boundary would bind; every entry point
 * names the reference interface it replaces.  The Python host side that presents
 * the reference's own method names on top of it lives in
 * deeprl_signal_control_amd/{env,agents,trainer}.py (ctypes), and INTEGRATION.md
 * shows the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on error; tsc_last_error()
 *     gives the message (thread-local).
 *   - "dev" pointers are device (HBM) pointers owned by the caller (e.g. torch
 *     tensors); "host" pointers are ordinary host memory.  No torch types.
 *   - one HIP stream per handle (tsc_*_set_stream); a handle is not thread-safe,
 *     handles are independent across threads / GPUs.
 *   - E = number of parallel env instances on this GPU, A = agents
 *     (intersections), SMAX / AMAX = padded observation / action widths.
 */
#ifndef TSC_H_
#define TSC_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TSC_MAX_UP     4   /* feeder lanes per lane                         */
#define TSC_MAX_CROSS  4   /* vehicles leaving one lane per simulated second */
#define TSC_LANE_CAP   28  /* vehicle slots per lane                         */

enum { TSC_AGENT_GREEDY = 0, TSC_AGENT_GLOBAL = 1 /* ia2c, iql */, TSC_AGENT_MA2C = 2 };
enum { TSC_OBJ_QUEUE = 0, TSC_OBJ_WAIT = 1, TSC_OBJ_HYBRID = 2 };

/* Dense scenario tables (host pointers, copied at create time).  Produced by
 * deeprl_signal_control_amd/scenario.py; meaning and reference provenance of every
 * table is documented there (large_grid/data/build_file.py, envs/large_grid_env.py,
 * envs/env.py:207-254,303-323). */
typedef struct tsc_scenario {
    int32_t n_lane, n_route, n_agent, n_flow;
    int32_t k_max;        /* signal links per agent (padded)   */
    int32_t p_max;        /* phases per agent (padded) = AMAX  */
    int32_t l_max;        /* incoming lanes per agent (padded) */
    int32_t s_max;        /* observation width (padded)        */
    int32_t nbr_max;      /* neighbours per agent (padded)     */
    const float   *lane_len, *lane_vmax, *lane_det_start;      /* [n_lane] */
    const int32_t *lane_node;                                  /* [n_lane] downstream agent or -1 */
    const int32_t *lane_up;                                    /* [n_lane, TSC_MAX_UP] */
    const int32_t *mv_next, *mv_link;                          /* [n_lane, n_route] */
    const int32_t *mv_yield, *mv_prio;                         /* [n_lane, n_route] right of way */
    const int32_t *mv_zip;                                     /* [n_lane, n_route] zipper slot: rank | count << 8 */
    const int32_t *route_entry;                                /* [n_route] */
    const int32_t *flows;                                      /* [n_flow, 4] begin,end,vph,route */
    const int32_t *agent_lanes;                                /* [n_agent, l_max] */
    const int32_t *agent_nlane, *agent_nlink, *agent_nphase;   /* [n_agent] */
    const uint8_t *green_tab;                                  /* [n_agent, p_max, k_max] */
    const uint8_t *yellow_tab;                                 /* [n_agent, p_max(prev), p_max(new), k_max] */
    const int32_t *nbr;                                        /* [n_agent, nbr_max], -1 padded */
    const int32_t *obs_kind, *obs_src;                         /* [n_agent, s_max] */
    /* ENV_CONFIG (config/config_*.ini) */
    int32_t control_interval_sec, yellow_interval_sec, episode_length_sec, teleport_sec;
    int32_t queue_cap;        /* real_net: min(10, halting) (envs/env.py:332-333); -1 = none */
    int32_t objective;        /* TSC_OBJ_*   */
    int32_t agent_kind;       /* TSC_AGENT_* */
    int32_t realnet_scale;    /* envs/env.py:599-601,625-629 */
    double coop_gamma, norm_wave, norm_wait, clip_wave, clip_wait, coef_wait;
    const float *lane_origin;  /* [n_lane] or NULL: where the SUMO lane begins inside a contracted lane chain (scenario.py
                                * contract_chains); the lane.* getters of the recording path count from here */
    /* Insertion streams (round 3).  A stream is one <flow> source: an entry lane, its flow elements (flows[.][3] is then
     * a STREAM index) and the route its vehicles take.  n_stream == 0: every route is its own stream (route_entry[] is the
     * entry lane, the layout of rounds 1-2).  Otherwise, per stream: stream_entry[s] = entry lane; stream_origin[s] = start
     * of the insertion window on it (metres; the SUMO entry lane may be an inner piece of a contracted chain);
     * stream_mode[s]: 0 = fixed route stream_choice[s][0][0];  1 = every vehicle draws its route from stream_choice[s]
     * ({route, cumulative weight of 65536} pairs, k_choice per stream and interval, route -1 pads) with the counter-based hash of
     * (seed, stream, serial) -- SUMO's jtrrouter turn ratios (small_grid/data/build_file.py:223-335);  2 = the route is
     * given per env instance and episode by tsc_env_set_stream_routes() -- the sinks large_grid's init_routes() draws
     * from np.random (large_grid/data/build_file.py:223-266); stream_choice[s] then lists the routes it may take. */
    int32_t n_stream, k_choice;
    const int32_t *stream_entry;   /* [n_stream] */
    const float   *stream_origin;  /* [n_stream] or NULL */
    const float   *stream_limit;   /* [n_stream] or NULL */
    const int32_t *stream_mode;    /* [n_stream] */
    const int32_t *stream_choice;  /* [n_stream, n_interval, k_choice, 2]; the choices in force at second t: interval
                                    * min(t / choice_interval_sec, n_interval - 1) (time-variant turn ratios) */
    int32_t n_interval, choice_interval_sec;
    /* Lane changing on two-lane streets (round 5; MICROSIM_SPEC.md rule 10; SUMO's lane-change model behind simulationStep,
     * envs/env.py:464, with the reference's connection table large_grid/data/build_file.py:107-124): lane_sib[l] = the other lane
     * of lane l's edge or -1; NULL = none.  mv_next then names the lane a junction's CONNECTION enters; a vehicle standing on a
     * lane whose mv_next entry for its route is "not served" (< -1) while lane_sib[l] serves it moves over as a hand-off that keeps
     * its position (lane_up must list the sibling FIRST: lane changers are gathered before the junction's arrivals). */
    const int32_t *lane_sib;       /* [n_lane] or NULL */
} tsc_scenario;

typedef struct tsc_env tsc_env;

const char *tsc_last_error(void);
int tsc_version(void);            /* 100 * major + minor; 105: tsc_env_set_greedy / tsc_env_greedy_actions; 104: tsc_env_counters, truncated trips flagged in tsc_env_read_trips */

/* Per-kernel timing with HIP events on the launch stream (bench.py's live roofline figure; the
 * reference has no equivalent).  Off by default; read() synchronises the recorded events.
 * enable(on): 0 = off, 1 = time every launch, n > 1 = time every n-th launch of the kernels that run once per
 * control step (an event pair serialises dependent kernels for a few microseconds); read() then returns
 * total_ms = average of the timed launches x all launches, count = all launches. */
int tsc_profile_enable(int32_t on);
int tsc_profile_reset(void);
/* Time only the kernel ids whose bit is set (0 = all, the default): an event pair between two dependent launches costs the
 * FOLLOWING launch up to ~17 us (the packets behind a kernel that leaves much dirty data), so a kernel's own duration is
 * measured with only ITS launches bracketed; read() of an unselected id returns count without time. */
int tsc_profile_select(uint64_t mask);
int tsc_profile_read(int32_t kernel_id, double *total_ms, int64_t *count);
const char *tsc_profile_name(int32_t kernel_id);   /* "" past the last id */

/* ---- env: replaces TrafficSimulator (envs/env.py:82-635) + SUMO/TraCI ------------------ */

/* TrafficSimulator.__init__ (envs/env.py:83-110) for E parallel instances. */
int tsc_env_create(const tsc_scenario *scn, int32_t n_env, int32_t device, tsc_env **out);
int tsc_env_destroy(tsc_env *h);                               /* terminate(), envs/env.py:563 */
int tsc_env_set_stream(tsc_env *h, void *hip_stream);
/* How many env instances share the DEVICE with this handle's (other handles of the process -- half-batches on separate
 * streams -- or other ranks on the same GPU), this handle's included.  The step kernel's workgroup size follows the device's
 * load, not the handle's: with fewer instances than workgroup slots an instance is spread over 512 / 1024 threads, with a full
 * device it runs 256.  Default: the handle's own n_env.  (No reference counterpart: one SUMO process per env, main.py:93.) */
int tsc_env_set_resident_instances(tsc_env *h, int32_t n_resident);

/* reset(), envs/env.py:544-561.  seeds: host [E] (the caller does the reference's
 * `seed += 1` bookkeeping); obs: dev float32 [E, A, SMAX] = float32(state) at t = 0. */
int tsc_env_reset(tsc_env *h, const uint32_t *seeds_host, float *obs_dev);

/* Routes of the mode-2 streams for the NEXT reset(): host int32 [E, n_stream] (entries of other streams are ignored) --
 * what gen_rou_file(seed) would write for every instance's episode seed (the caller draws them; env.py does it with
 * numpy's RandomState(seed), the generator the reference's np.random.seed(seed) + np.random.choice uses). */
int tsc_env_set_stream_routes(tsc_env *h, const int32_t *routes_host);

/* update_fingerprint(policy), envs/env.py:633-635.  pi: dev float32 [E, A, AMAX];
 * entries k >= n_a - 1 are ignored. */
int tsc_env_set_fingerprint(tsc_env *h, const float *pi_dev);

/* Zero-copy variant of update_fingerprint: the env gathers fingerprints straight from the caller's policy
 * buffer (dev float32 [E, A, AMAX]) at the next step(), which is when the reference reads them; the buffer
 * must stay untouched until then (true for Trainer.explore, utils.py:148-160).  reset() and
 * tsc_env_set_fingerprint() unbind. */
int tsc_env_bind_fingerprint(tsc_env *h, const float *pi_dev);

/* The reference's greedy controllers -- LargeGridController (envs/large_grid_env.py:45-60), RealNetController
 * (envs/real_net_env.py:78-111), SmallGridController (envs/small_grid_env.py:40-55) -- as one rule over tables (what their
 * constructors hold: node names, phase strings, STATE_PHASE_MAP): agent a compares n_cand[a] candidate flows; candidate c is
 * the float64 sum, from 0 and in table order, of the observation entries term[a][c][.] (indices into the agent's
 * observation row, -1 ends the list); np.argmax keeps the first maximum; the winner stands for action cand_action[a][c].
 * Host pointers, int32: n_cand [A], term [A, n_cand_max, n_term_max], cand_action [A, n_cand_max]; copied.
 * deeprl_signal_control_amd/scenario.py:Scenario.greedy_controller_tables compiles them. */
int tsc_env_set_greedy(tsc_env *h, int32_t n_cand_max, int32_t n_term_max, const int32_t *n_cand, const int32_t *term,
                       const int32_t *cand_action);
/* Controller.forward(obs) for every instance (envs/large_grid_env.py:50-54): obs dev float32 [E, A, SMAX] as reset() /
 * step() wrote it, action dev int32 [E, A].  The controllers read the env's float64 state; the kernel recovers it from the
 * float32 entries (a wave entry is a vehicle count / norm_wave, clipped: envs/env.py:439-442), so ties fall as in numpy. */
int tsc_env_greedy_actions(tsc_env *h, const float *obs_dev, int32_t *action_dev);

/* Sum over instances and control steps of the global reward since the last reset of the accumulator
 * (what Trainer logs per episode, utils.py:161,296-305).  Synchronises. */
int tsc_env_reward_sum(tsc_env *h, double *sum_host, int32_t reset);

/* step(action), envs/env.py:566-631: yellow FSM -> 2 sim-steps -> green -> 3 sim-steps
 * -> detectors -> obs -> reward -> shaping.  action: dev int32 [E, A];
 * obs: dev float32 [E, A, SMAX]; reward: dev float64 [E, A]; global_reward: dev float64 [E];
 * done: dev uint8 [E].  train_mode = 0 returns local rewards (envs/env.py:590-592). */
int tsc_env_step(tsc_env *h, const int32_t *action_dev, float *obs_dev, double *reward_dev,
                 double *global_reward_dev, uint8_t *done_dev, int32_t train_mode);

/* Debug / parity access: vehicle state of env `e` as dense host arrays [n_lane, TSC_LANE_CAP]
 * (front vehicle first) + counts [n_lane] + per-stream pending/serial [n_stream, = n_route without stream tables].
 * Synchronises. */
int tsc_env_get_state(tsc_env *h, int32_t e, int32_t *n, float *x, float *v, float *sf,
                      int32_t *w, int32_t *r, int32_t *pending, int32_t *serial, int32_t *time_sec);

/* Tuning aid: shader-clock stamps (s_memtime) taken by workgroup 0 / thread 0 of the last step at
 * every phase boundary; stamps[63] = number of stamps. enable != 0 allocates the buffer; enable == 2 also
 * returns [64 + 2e], [64 + 2e + 1] = constant-rate (100 MHz) start / end time of workgroup e (host buffer of
 * 64 + 2E entries); enable == 3 also [64 + 2E + 5e + b] = shader-clock cycles workgroup e spent in phase kind b (0 prologue +
 * epilogue, 1 head walk, 2 flat phase, 3 gather + demand, 4 the barriers between them; host buffer of 64 + 7E entries; zeros unless
 * the library was built with -DTSC_ENV_PHASE_SUMS, tools/build_variant.sh). */
int tsc_env_debug_clock(tsc_env *h, int32_t enable, int64_t *stamps64_host);

/* Tuning aids of the simulator's launch geometry.  tsc_env_vehicle_counts: vehicles in the network of every instance (host int32
 * [E]; synchronises).  tsc_env_set_block_order: workgroup b of tsc_env_step simulates instance order[b] (host int32 [E], a
 * permutation; null = identity).  The order changes which instances share a CU, never a result. */
int tsc_env_vehicle_counts(tsc_env *h, int32_t *counts_host);
int tsc_env_set_block_order(tsc_env *h, const int32_t *order_host);

/* Per-instance counters of the running episode, uint64 [E] each (either pointer may be null): vehicles that reached the
 * end of their route (simulation.getArrivedNumber summed over the episode, envs/env.py:413) and vehicles the teleport
 * surrogate removed (SUMO --time-to-teleport, envs/env.py:283-284; they are NOT arrivals).  Host pointers. Synchronises. */
int tsc_env_counters(tsc_env *h, uint64_t *arrived_host, uint64_t *teleported_host);
/* Mean number of live vehicles per env (roofline bookkeeping). Synchronises. */
int tsc_env_live_vehicles(tsc_env *h, double *mean_live);
/* sum over instances and control steps since the last reset of the accumulator of the vehicles in
 * the network at the end of the step: window-mean V = sum / (steps * E)  (SURVEY.md 8d); reset() does not clear it.
 * Synchronises. */
int tsc_env_live_sum(tsc_env *h, double *sum_host, int32_t reset);

/* Evaluation recording = init_data(is_record=True) (envs/env.py:517-528): the following steps also keep, per simulated
 * second, _measure_traffic_step's inputs (envs/env.py:409-437) and a log of finished trips (the --tripinfo-output file
 * collect_tripinfo parses, :498-515).  Off on the training path (separate kernel instantiation).  Call before reset(). */
int tsc_env_record(tsc_env *h, int32_t enable, int32_t trip_cap);
/* Rows of the LAST step: ints [E, 8, 4] = vehicles in the network, departed, arrived, sum of waiting times of second q <
 * control_interval_sec; speed [E, 8] = sum of speeds; queue [E, 8, A * l_max] = lane.getLastStepHaltingNumber of every
 * incoming lane in (agent, ild) order (-1-padded lanes report 0).  Host pointers. Synchronises. */
int tsc_env_read_record(tsc_env *h, int64_t *ints_host, double *speed_host, int32_t *queue_host);
/* Finished trips of instance e since reset(): rows {route, serial within the route, depart_sec, arrival_sec, waiting
 * seconds, waiting count}; *count = trips finished (may exceed max_trips / the capacity given to tsc_env_record).
 * A row with a NEGATIVE arrival_sec (= -second) is a trip the teleport surrogate truncated (MICROSIM_SPEC.md rule 1): SUMO
 * would have moved that vehicle on and written its tripinfo later, so collect_tripinfo (envs/env.py:498-515) must not
 * count it as a finished trip. */
int tsc_env_read_trips(tsc_env *h, int32_t e, int32_t *trips_host, int32_t max_trips, int32_t *count);

/* ---- model: replaces IA2C / MA2C (agents/models.py:132-262) + LstmACPolicy / FPLstmACPolicy
 *      (agents/policies.py:75-211) + OnPolicyBuffer (agents/utils.py:182-228) + the TF1 runtime ---- */

typedef struct tsc_model_cfg {
    int32_t n_agent;          /* A                                                        */
    int32_t s_max, a_max;     /* padded obs / action widths (= tsc_scenario s_max, p_max) */
    const int32_t *n_wave;    /* [A] wave inputs   (policy n_s = env n_s - n_w - n_f)      */
    const int32_t *n_wait;    /* [A] wait inputs   (n_w)                                   */
    const int32_t *n_fp;      /* [A] fingerprint inputs (n_f; 0 for IA2C)                 */
    const int32_t *n_act;     /* [A] actions (n_a)                                        */
    int32_t n_fc_wave, n_fc_wait, n_fc_fp, n_lstm;   /* num_fw, num_ft, num_fp, num_lstm  */
    int32_t n_step;           /* batch_size: control steps per update                     */
    double gamma, reward_norm, reward_clip, value_coef, max_grad_norm, rmsp_alpha, rmsp_epsilon;
    int32_t policy_kind;      /* 0 = LstmACPolicy / FPLstmACPolicy (agents/policies.py:75-211),
                                 1 = FcACPolicy (agents/policies.py:214-256; fingerprints unsupported there) */
} tsc_model_cfg;

typedef struct tsc_model tsc_model;

/* Parameter layout.  Agent-tower group g = 2*agent + tower (0 = pi, 1 = v); every group owns
 * `stride` consecutive floats: W1[s_max][H] (block-diagonal fcw|fcf|fct, obs rows in env order),
 * b1[H], Wx[H][4L], Wh[L][4L], bl[4L], Wo[L][8], bo[8]   (H = n_fc_wave+n_fc_fp+n_fc_wait,
 * L = n_lstm, LSTM gate order i,f,o,u as agents/utils.py:107).  FC policy: Wx -> Wfc[H][L], no Wh, bl -> bfc[L].  out[] = {G, stride, H, L,
 * off_W1, off_b1, off_Wx, off_Wh, off_bl, off_Wo, off_bo, out_pad(8)}. */
int tsc_model_create(const tsc_model_cfg *cfg, int32_t n_env, int32_t device, tsc_model **out);
int tsc_model_destroy(tsc_model *m);
int tsc_model_set_stream(tsc_model *m, void *hip_stream);
int tsc_model_layout(tsc_model *m, int64_t out[12]);
int tsc_model_set_params(tsc_model *m, const float *params_host);     /* optimizer state untouched */
int tsc_model_reset_opt_state(tsc_model *m);                         /* RMSProp ms <- 1 (TF1 slot init) */
int tsc_model_get_params(tsc_model *m, float *params_host);
int tsc_model_get_opt_state(tsc_model *m, float *ms_host);
int tsc_model_set_opt_state(tsc_model *m, const float *ms_host);

/* IA2C.reset (agents/models.py:218-220): zero the forward and backward LSTM states. */
int tsc_model_reset(tsc_model *m);

/* IA2C.forward (agents/models.py:185-200 -> policies.py:125-136).  obs: dev f32 [E,A,SMAX];
 * done: dev u8 [E] (pre-decision done, resets the LSTM state); pi: dev f32 [E,A,AMAX] (padded
 * actions get 0), v: dev f32 [E,A].  advance = 1 <=> out_type contains 'p' (state is advanced);
 * advance = 0 <=> out_type 'v' (bootstrap value, state untouched, policies.py:127-135). */
int tsc_model_forward(tsc_model *m, const float *obs_dev, const uint8_t *done_dev, float *pi_dev,
                      float *v_dev, int32_t advance);

/* forward(out_type='pv') + tsc_model_sample in one call (the action is drawn inside the fused forward).
 * t_slot = index of the transition this forward belongs to (0 .. n_step-1, the slot the following
 * tsc_model_add_transition fills) lets the kernel keep the step's activations for the update, so that
 * tsc_model_compute_grads need not re-evaluate the forward graph (agents/policies.py:94-96 builds it twice);
 * t_slot = -1 disables that.  The cache is used only if all n_step slots were filled in order. */
int tsc_model_forward_sample(tsc_model *m, const float *obs_dev, const uint8_t *done_dev, float *pi_dev,
                             float *v_dev, int32_t *action_dev, uint64_t seed, uint64_t step, int32_t t_slot);

/* np.random.choice(n_a, p=pi) per agent (utils.py:155-157) with a counter-based generator:
 * u = U(seed, step, e, a); action = searchsorted(cumsum(pi)/sum, u, right). action: dev i32 [E,A]. */
int tsc_model_sample(tsc_model *m, const float *pi_dev, int32_t *action_dev, uint64_t seed, uint64_t step);

/* IA2C.add_transition (agents/models.py:222-229): reward / reward_norm, clip, store
 * (obs, action, reward, value, done) at slot t of the on-policy buffer.  reward: dev f64 [E,A]. */
int tsc_model_add_transition(tsc_model *m, int32_t t, const float *obs_dev, const uint8_t *done_pre_dev,
                             const int32_t *action_dev, const double *reward_dev, const float *value_dev,
                             const uint8_t *done_post_dev);

/* Zero-copy rollouts: device pointers of transition slot t of the on-policy buffer, ptrs[6] = {obs f32 [E,A,SMAX],
 * action i32 [E,A], value f32 [E,A], reward f64 [E,A] (raw; normalised / clipped when the returns are computed), done before
 * the step u8 [E], done after the step u8 [E]}.  A caller that lets tsc_model_forward_sample write action / value and
 * tsc_env_step write reward / done / the next observation (slot t + 1; slot n_step exists for that, and is copied to slot
 * 0 by tsc_model_apply_grads) straight into the slots has nothing left for tsc_model_add_transition to do. */
int tsc_model_rollout_slot(tsc_model *m, int32_t t, void *ptrs[6]);

/* IA2C.backward part 1 (agents/models.py:174-183 -> utils.py:202-228 -> policies.py:41-57,138-152):
 * n-step returns/advantages (float64, bootstrap R_boot dev f32 [E,A], ignored where the last
 * transition was terminal), BPTT through the n_step-unrolled LSTM from the saved backward state,
 * loss, gradients.  Gradients land in the contiguous buffer tsc_model_grad_buffer() returns
 * (same layout as the parameters) so the caller can all-reduce it over RCCL. */
int tsc_model_compute_grads(tsc_model *m, const float *R_boot_dev, double entropy_beta);
int tsc_model_grad_buffer(tsc_model *m, float **grad_dev, int64_t *count);

/* IA2C.backward part 2 (policies.py:57-61): per-agent clip_by_global_norm(max_grad_norm) on
 * grad * grad_scale, TF1 RMSPropOptimizer step, states_bw <- states_fw (policies.py:153),
 * buffer reset carrying the last done (utils.py:227).  stats_host (nullable): per agent
 * {policy_loss, value_loss, entropy_loss, grad_norm} float64 [A,4]. */
int tsc_model_apply_grads(tsc_model *m, double lr, double grad_scale, double *stats_host);

/* Tuning aid for the fused rollout forward: like tsc_env_debug_clock (phase stamps of one workgroup in
 * [0,63), per-workgroup 100 MHz start / end from index 64). */
int tsc_model_debug_clock(tsc_model *m, int32_t enable, int64_t *stamps_host, int32_t count);

/* Debug / parity access: the float32 returns and advantages [n_step, E, A] the last
 * tsc_model_compute_grads() fed to the loss (agents/utils.py:223-224). Synchronises. */
int tsc_model_get_returns(tsc_model *m, float *Rs_host, float *Advs_host);

/* Debug / parity access to the training activations of agent-tower g, rows [row0, row0 + nrows) of the
 * [n_step * E] sample axis (row = t * E + e): what = 0 X1 [H], 1 gates i|f|o|u (dz after compute_grads) [256],
 * 2 h [64], 3 c [64], 4 masked h_prev [64], 5 dH [64].  After n_step tsc_model_forward_sample calls with
 * t_slot = 0..n_step-1 these are the rows the fused forward cached for the update.  Synchronises. */
int tsc_model_debug_read(tsc_model *m, int32_t what, int32_t g, int64_t row0, int64_t nrows, float *out_host);

/* ---- IQL: replaces IQL (agents/models.py:264-376) + LRQPolicy / DeepQPolicy (agents/policies.py:285-389) +
 *      ReplayBuffer (agents/utils.py:231-263) + the TF1 runtime (Adam) ------------------------------------------- */

typedef struct tsc_iql_cfg {
    int32_t n_agent, s_max, a_max;
    const int32_t *n_wave;    /* [A] wave inputs = n_s - n_w  (DeepQPolicy's n_s, agents/models.py:301) */
    const int32_t *n_wait;    /* [A] wait inputs (n_w)                                                  */
    const int32_t *n_act;     /* [A] actions (n_a)                                                      */
    int32_t kind;             /* 0 = LRQPolicy (model_type 'lr'), 1 = DeepQPolicy ('dqn')               */
    int32_t n_fc0, n_h;       /* num_fc, num_h (dqn; the wait FC is n_fc0 / 4 wide, policies.py:359)     */
    int32_t batch_size;       /* minibatch rows per env instance (= n_step, agents/models.py:275)        */
    int32_t buffer_size;      /* replay ring per env instance and agent                                  */
    double gamma, reward_norm, reward_clip, max_grad_norm;
} tsc_iql_cfg;

typedef struct tsc_iql tsc_iql;

/* Parameter layout, `stride` floats per agent:  dqn: W1[s_max][H1] | b1[H1] | W2[H1][H2] | b2[H2] | Wq[H2][8] | bq[8]
 * (H1 = n_fc0 + n_fc0/4, W1 block-diagonal q_fcw | q_fct over the obs in env order);  lr: Wq[s_max][8] | bq[8].
 * out[] = {A, stride, H1, H2, off_W1, off_b1, off_W2, off_b2, off_Wq, off_bq, out_pad(8), kind}. */
int tsc_iql_create(const tsc_iql_cfg *cfg, int32_t n_env, int32_t device, tsc_iql **out);
int tsc_iql_destroy(tsc_iql *h);
int tsc_iql_set_stream(tsc_iql *h, void *hip_stream);
int tsc_iql_layout(tsc_iql *h, int64_t out[12]);
int tsc_iql_set_params(tsc_iql *h, const float *params_host);
int tsc_iql_get_params(tsc_iql *h, float *params_host);
int tsc_iql_get_opt_state(tsc_iql *h, float *m_host, float *v_host, int64_t *adam_t);   /* Adam moments + step count */
int tsc_iql_set_opt_state(tsc_iql *h, const float *m_host, const float *v_host, int64_t adam_t);

/* IQL.forward (agents/models.py:332-348).  obs: dev f32 [E,A,SMAX]; q: dev f32 [E,A,AMAX] (padded actions 0);
 * action: dev i32 [E,A].  mode 0 = argmax ('act'), 1 = 'explore' (u0 < eps ? floor(u1 * n_a) : argmax),
 * 2 = stochastic (qs / sum(qs) -> np.random.choice).  Uniforms: u_s = U(seed, step, 2 * (e * A + a) + s), the
 * counter-based generator of tsc_model_sample. */
int tsc_iql_forward(tsc_iql *h, const float *obs_dev, float *q_dev, int32_t *action_dev, int32_t mode, double eps,
                    uint64_t seed, uint64_t step);

/* IQL.add_transition (agents/models.py:354-361): reward / reward_norm, clip, append (obs, a, r, next_obs, done) to
 * every instance's ring (slot = transitions so far % buffer_size).  reward: dev f64 [E,A]; done: dev u8 [E]. */
int tsc_iql_add_transition(tsc_iql *h, const float *obs_dev, const int32_t *action_dev, const double *reward_dev,
                           const float *next_obs_dev, const uint8_t *done_dev);
int tsc_iql_replay_size(tsc_iql *h, int64_t *size, int64_t *cum_size);       /* ReplayBuffer.size / cum_size */

/* One minibatch step of IQL.backward (agents/models.py:319-330 -> policies.py:305-328), first half: draw batch_size
 * distinct transitions per (instance, agent) -- Floyd's algorithm on U(seed, update_index, (e * A + a) * B + i) --
 * and leave the gradient of mean((Q(s)[a] - stop_grad(done ? r : r + gamma max Q(s')))^2) over the E * batch_size rows
 * of every agent in the contiguous buffer tsc_iql_grad_buffer() returns (parameter layout). */
int tsc_iql_compute_grads(tsc_iql *h, uint64_t seed, uint64_t update_index);
/* The same with the caller's draw instead of the built-in one: idx dev i32 [E, A, batch_size], every entry a ring slot
 * in [0, replay size) -- what ReplayBuffer.sample_transition's random.sample picked (agents/utils.py:252-258); an entry
 * outside that range is clamped into it (never an out-of-bounds read; validate on the host if it matters).  Lets a
 * host that owns the sampling (or a recorded reference run, tests/test_refnet_iql_gpu.py) drive the update. */
int tsc_iql_compute_grads_at(tsc_iql *h, const int32_t *idx_dev);
int tsc_iql_grad_buffer(tsc_iql *h, float **grad_dev, int64_t *count);
/* ... second half: per-agent clip_by_global_norm(max_grad_norm) on grad * grad_scale, TF1 AdamOptimizer step
 * (beta1 .9, beta2 .999, epsilon 1e-8).  stats_host (nullable): per agent {loss, grad_norm} float64 [A,2]. */
int tsc_iql_apply_grads(tsc_iql *h, double lr, double grad_scale, double *stats_host);
/* Debug / parity access: the replay indices [E, A, batch_size] of the last tsc_iql_compute_grads. Synchronises. */
int tsc_iql_debug_batch(tsc_iql *h, int32_t *idx_host);
/* Which learner the handle runs: *fused = 1 when tsc_iql_forward / tsc_iql_compute_grads are the one-kernel DeepQPolicy path
 * (csrc/tsc_iql_fused.h: num_fc 128, num_h 64, s_max <= 48 -- the reference's configurations), 0 for the grouped-GEMM path
 * (IQL-LR, other widths, or TSC_IQL_FUSED=0 in the environment when the handle was created). */
int tsc_iql_path(tsc_iql *h, int32_t *fused);
/* Measurement hook of the fused learner (tools/bench_iql.py --stamps): enable != 0 allocates the stamp buffer; the next
 * tsc_iql_compute_grads then records, for every workgroup, its start / end on the 100-MHz wall clock ([64 + 2 b],
 * [64 + 2 b + 1]) and -- in a measurement build with -DTSC_IQL_STAMPS (tools/build_variant.sh; the stamps' branches are kept
 * out of the product kernel) -- for workgroup 0 eleven shader-clock stamps per wavefront around the phases of its third 64-row
 * chunk ([16 w + k], csrc/tsc_iql_fused.h QSTAMP).  stamps_host (nullable) receives min(count, 64 + 2 x workgroups) values.
 * Synchronises. */
int tsc_iql_debug_clock(tsc_iql *h, int32_t enable, int64_t *stamps_host, int32_t count);

/* Test hook: the grouped fp32 MFMA GEMM used by every layer.  form: 0 = NN, 1 = TN; epi as
 * csrc/tsc_gemm.h.  All pointers device; strides in elements.  A non-null split-K workspace lets
 * the TN form cut its reduction into deterministic chunks (as the weight-gradient GEMMs do). */
int tsc_gemm_grouped_f32(int32_t form, int32_t epi, int32_t groups, int32_t M, int32_t N, int32_t K,
                         const float *A, int64_t sA, int32_t lda, const float *B, int64_t sB, int32_t ldb,
                         float *C, int64_t sC, int32_t ldc, const float *bias, const float *aux,
                         const int16_t *rowrange, float *colsum, float *splitk_ws, int64_t ws_floats,
                         float *splitk_wsc, int64_t wsc_floats, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* TSC_H_ */
