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
    const int32_t *lane_node, *lane_opp;                       /* [n_lane] */
    const int32_t *lane_up;                                    /* [n_lane, TSC_MAX_UP] */
    const int32_t *mv_next, *mv_link;                          /* [n_lane, n_route] */
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
} tsc_scenario;

typedef struct tsc_env tsc_env;

const char *tsc_last_error(void);
int tsc_version(void);

/* ---- env: replaces TrafficSimulator (envs/env.py:82-635) + SUMO/TraCI ------------------ */

/* TrafficSimulator.__init__ (envs/env.py:83-110) for E parallel instances. */
int tsc_env_create(const tsc_scenario *scn, int32_t n_env, int32_t device, tsc_env **out);
int tsc_env_destroy(tsc_env *h);                               /* terminate(), envs/env.py:563 */
int tsc_env_set_stream(tsc_env *h, void *hip_stream);

/* reset(), envs/env.py:544-561.  seeds: host [E] (the caller does the reference's
 * `seed += 1` bookkeeping); obs: dev float32 [E, A, SMAX] = float32(state) at t = 0. */
int tsc_env_reset(tsc_env *h, const uint32_t *seeds_host, float *obs_dev);

/* update_fingerprint(policy), envs/env.py:633-635.  pi: dev float32 [E, A, AMAX];
 * entries k >= n_a - 1 are ignored. */
int tsc_env_set_fingerprint(tsc_env *h, const float *pi_dev);

/* step(action), envs/env.py:566-631: yellow FSM -> 2 sim-steps -> green -> 3 sim-steps
 * -> detectors -> obs -> reward -> shaping.  action: dev int32 [E, A];
 * obs: dev float32 [E, A, SMAX]; reward: dev float64 [E, A]; global_reward: dev float64 [E];
 * done: dev uint8 [E].  train_mode = 0 returns local rewards (envs/env.py:590-592). */
int tsc_env_step(tsc_env *h, const int32_t *action_dev, float *obs_dev, double *reward_dev,
                 double *global_reward_dev, uint8_t *done_dev, int32_t train_mode);

/* Debug / parity access: vehicle state of env `e` as dense host arrays [n_lane, TSC_LANE_CAP]
 * (front vehicle first) + counts [n_lane] + per-route pending/serial [n_route]. Synchronises. */
int tsc_env_get_state(tsc_env *h, int32_t e, int32_t *n, float *x, float *v, float *sf,
                      int32_t *w, int32_t *r, int32_t *pending, int32_t *serial, int32_t *time_sec);

/* Mean number of live vehicles per env (roofline bookkeeping). Synchronises. */
int tsc_env_live_vehicles(tsc_env *h, double *mean_live);

#ifdef __cplusplus
}
#endif
#endif /* TSC_H_ */
