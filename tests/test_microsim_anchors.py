"""Statistical gate of the microsimulator spec against the only published anchors of the
reference's SUMO runs: the greedy controllers' mean step reward.  SUMO itself is absent, the
dynamics are this repo's spec -- the bands below are what the spec produces today (regression gate), NOT SUMO's
numbers; the distance to the anchors is asserted as such so that nobody reads "calibrated" into it:

  large_grid greedy   published -972.28 (result_plot.ipynb:188)      this spec ~ -66   (all vehicles arrive)
  Monaco greedy       published  -41.8  (real_net_experimental_data)  this spec ~ -137  (congested, but it flows: round 3)

Round 3 changed the spec: merge arbitration by readiness, a teleport surrogate that removes a
blocked head after time-to-teleport, headway 1.0 s.  Monaco under the reference's greedy controller went from a
permanent gridlock after t = 1300 s (355 trips, 1.96 m/s, 395 s mean wait) to 1486 completed trips (489 of them ended by
the teleport surrogate), 2.8 m/s and 81 s mean wait.  DESIGN.md section 3 ("calibration") records what was measured
and why the anchors stay out of reach."""
import numpy as np

from deeprl_signal_control_amd.scenario import build_large_grid, build_real_net
from deeprl_signal_control_amd.trainer import greedy_actions


def _episode(scn, seed, act):
    from oracle.env_oracle import OracleEnv
    env = OracleEnv(scn, seed=seed, train_mode=False, test_seeds=(seed,))
    ob = env.reset(0)
    rs, live = [], []
    while True:
        ob, r, done, g = env.step(act(ob))
        rs.append(g); live.append(env.ms.totals()['live'])
        assert env.ms.check() == 0
        if done:
            break
    return float(np.mean(rs)), max(live), env.ms.totals()


def test_large_grid_greedy_band():
    from oracle.env_oracle import greedy_large_grid
    scn = build_large_grid('greedy', norm_wave=1.0, norm_wait=1.0, clip_wave=-1.0, clip_wait=-1.0)
    r, peak, tot = _episode(scn, 10000, lambda ob: [greedy_large_grid(o[:6]) for o in ob])
    assert -120.0 < r < -30.0 and 350 < peak < 700                    # today: -66.5, 518 concurrent vehicles
    assert tot['teleported'] == 0                                     # nobody stands for 600 s on the grid
    assert tot['departed'] == tot['arrived'] == 3717 and tot['pending'] == 0     # demand of A.3 served completely
    assert 150 < tot['sum_trip'] / tot['arrived'] < 350               # mean trip ~230 s (free flow ~110 s)
    assert r / -972.28 < 0.15                                         # an order of magnitude less congested than SUMO


def test_monaco_greedy_band():
    scn = build_real_net('greedy', norm_wave=1.0, clip_wave=-1.0)
    L = scn.agent_lanes.shape[1]

    def act(ob):
        w = np.zeros((scn.n_agent, L))
        for a, o in enumerate(ob):
            w[a, :len(o)] = o
        return list(greedy_actions(scn, w))
    r, peak, tot = _episode(scn, 10000, act)
    assert -200.0 < r < -90.0 and 538 <= peak <= 734                  # today -137.0, 608: peak inside the published 538-734
    assert tot['departed'] + tot['pending'] > 2300                    # ~2383 vehicles demanded (A.4)
    assert tot['arrived'] > 1000 and tot['departed'] > 1500           # today 1486 of 1950 inserted (round 2: 355 of 999)
    assert 200 < tot['teleported'] < 600                              # 489: the greedy controller starves shared lanes
    assert r / -41.8 > 2.0                                            # still several times MORE congested than SUMO's greedy run


# Aggregates of real_net_experimental_data/eva_data/real_net_greedy_{traffic,trip,control}.csv (10 evaluation episodes of
# the authors' SUMO run): the statistical counterpart of the evaluation tables this repo writes.
PUBLISHED_MONACO_GREEDY = dict(avg_queue=0.51, avg_speed_mps=6.06, avg_wait_sec=65.5, peak_cars=322, trips=1945,
                               trip_duration_sec=238.0, reward=-41.8)


def test_monaco_greedy_eval_tables_vs_published():
    """The recorded evaluation tables (envs/env.py:409-437,498-542 schema) of one greedy Monaco episode under this spec,
    next to the published ones: every aggregate says the same thing as the reward anchor -- the spec's Monaco jams."""
    from oracle.env_oracle import OracleEnv
    scn = build_real_net('greedy', norm_wave=1.0, clip_wave=-1.0)
    L = scn.agent_lanes.shape[1]
    env = OracleEnv(scn, seed=10000, train_mode=False, test_seeds=(10000,), is_record=True)
    ob = env.reset(0)
    while True:
        w = np.zeros((scn.n_agent, L))
        for a, o in enumerate(ob):
            w[a, :len(o)] = o
        ob, r, done, g = env.step(list(greedy_actions(scn, w)))
        if done:
            break
    env.collect_tripinfo()
    traffic, trips = env.traffic_data, env.trip_data
    assert len(traffic) == 3600 and set(traffic[0]) >= {'avg_queue', 'avg_speed_mps', 'avg_wait_sec', 'number_total_car'}
    ours = dict(avg_queue=np.mean([t['avg_queue'] for t in traffic]), avg_speed_mps=np.mean([t['avg_speed_mps'] for t in traffic]),
                avg_wait_sec=np.mean([t['avg_wait_sec'] for t in traffic]), peak_cars=max(t['number_total_car'] for t in traffic),
                trips=len(trips))
    # today: queue 1.32 veh/lane, 2.79 m/s, 81 s mean wait, 608 concurrent vehicles, 1486 completed trips
    # (round 2: 1.71, 1.96 m/s, 395 s, 644, 355)
    assert 0.8 < ours['avg_queue'] < 2.0 and 2.0 < ours['avg_speed_mps'] < 4.0 and 50 < ours['avg_wait_sec'] < 150
    assert 500 < ours['peak_cars'] < 800 and 1000 < ours['trips'] < 1700
    pub = PUBLISHED_MONACO_GREEDY
    assert ours['avg_queue'] > 2 * pub['avg_queue'] and ours['avg_speed_mps'] < 0.6 * pub['avg_speed_mps']
    assert 0.5 * pub['trips'] < ours['trips'] < 0.8 * pub['trips']   # SUMO's run completes ~1945 trips per episode
