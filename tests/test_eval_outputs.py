"""Evaluation outputs: per-second network statistics, control log and trip info of the reference's
recording path (envs/env.py:409-437, :498-542, :581-588).  Fixtures tests/golden/{large_grid,real_net}_eval.npz were
recorded from the reference's OWN classes with is_record=True over the fake TraCI backend (tools/make_golden.py
eval_fixtures; the backend writes the tripinfo file SUMO would); the oracle restatement (CPU) and the HIP recording
kernel (GPU) must reproduce them: integers and strings exactly, float means to 1e-12 (np.mean's pairwise sum vs the
per-lane partial sums both restatements use)."""
import os

import numpy as np
import pytest

from deeprl_signal_control_amd.scenario import build_large_grid, build_real_net
from deeprl_signal_control_amd.trainer import greedy_actions

KW = dict(norm_wave=1.0, norm_wait=1.0, clip_wave=1000.0, clip_wait=1000.0, coop_gamma=0.75)


def _scn(name):
    if name == 'large_grid':
        return build_large_grid('greedy', episode_length_sec=1200, **KW)
    return build_real_net('greedy', episode_length_sec=900, **KW)


def _greedy(scn):
    from oracle.env_oracle import greedy_large_grid
    L = scn.agent_lanes.shape[1]

    def act(ob):
        if scn.name == 'large_grid':
            return [greedy_large_grid(o[:6]) for o in ob]
        w = np.zeros((scn.n_agent, L))
        for a, o in enumerate(ob):
            w[a, :len(o)] = o
        return [int(x) for x in greedy_actions(scn, w)]
    return act


def _run(env, scn):
    env.train_mode = False
    ob = env.reset(test_ind=0)
    act = _greedy(scn)
    while True:
        ob, _, done, _ = env.step(act(ob))
        if done:
            break
    env.collect_tripinfo()


def _check(g, traffic, control, trips):
    for k in ('episode', 'time_sec', 'number_total_car', 'number_departed_car', 'number_arrived_car'):
        np.testing.assert_array_equal(np.array([r[k] for r in traffic]), g['traffic_' + k], err_msg=k)
    for k in ('avg_wait_sec', 'avg_speed_mps', 'std_queue', 'avg_queue'):
        np.testing.assert_allclose(np.array([r[k] for r in traffic], np.float64), g['traffic_' + k], rtol=1e-12, atol=1e-12, err_msg=k)
    for k in ('episode', 'time_sec', 'step', 'reward'):
        np.testing.assert_array_equal(np.array([r[k] for r in control]), g['control_' + k], err_msg=k)
    assert [r['action'] for r in control] == list(g['control_action'])
    want = sorted(zip(*(g['trip_' + k] for k in ('arrival_sec', 'id', 'depart_sec', 'duration_sec', 'wait_step', 'wait_sec'))),
                  key=lambda t: (float(t[0]), t[1]))
    got = sorted(((r['arrival_sec'], r['id'], r['depart_sec'], r['duration_sec'], r['wait_step'], r['wait_sec']) for r in trips),
                 key=lambda t: (float(t[0]), t[1]))
    assert len(got) == len(want) > 100 and got == [tuple(str(x) for x in t) for t in want]
    assert all(r['episode'] == 1 for r in trips)


@pytest.mark.parametrize('name', ['large_grid', 'real_net'])
def test_oracle_recording_matches_reference(name, golden_dir):
    from oracle.env_oracle import OracleEnv
    scn = _scn(name)
    env = OracleEnv(scn, seed=42, test_seeds=(10000,), is_record=True)
    _run(env, scn)
    _check(np.load(os.path.join(golden_dir, name + '_eval.npz')), env.traffic_data, env.control_data, env.trip_data)


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['large_grid', 'real_net'])
def test_gpu_recording_matches_reference_and_writes_csv(name, golden_dir, tmp_path):
    import pandas as pd
    from deeprl_signal_control_amd.env import TrafficEnv
    scn = _scn(name)
    out = str(tmp_path) + '/'
    env = TrafficEnv(scn, output_path=out, is_record=True, seed=42, test_seeds=(10000,))
    _run(env, scn)
    g = np.load(os.path.join(golden_dir, name + '_eval.npz'))
    _check(g, env.traffic_data, env.control_data, env.trip_data)
    env.output_data()
    # the three tables of envs/env.py:534-542, columns as in real_net_experimental_data/eva_data/*.csv
    t = pd.read_csv(out + '%s_greedy_traffic.csv' % name, index_col=0)
    assert list(t.columns) == ['avg_queue', 'avg_speed_mps', 'avg_wait_sec', 'episode', 'number_arrived_car',
                               'number_departed_car', 'number_total_car', 'std_queue', 'time_sec']
    assert len(t) == scn.episode_length_sec and t.time_sec.iloc[-1] == scn.episode_length_sec
    c = pd.read_csv(out + '%s_greedy_control.csv' % name, index_col=0)
    assert list(c.columns) == ['action', 'episode', 'reward', 'step', 'time_sec'] and len(c) == scn.episode_length_sec // 5
    assert len(c.action.iloc[0].split(',')) == scn.n_agent
    tr = pd.read_csv(out + '%s_greedy_trip.csv' % name, index_col=0)
    assert list(tr.columns) == ['arrival_sec', 'depart_sec', 'duration_sec', 'episode', 'id', 'wait_sec', 'wait_step']
    assert len(tr) == len(g['trip_id']) and (tr.duration_sec == tr.arrival_sec - tr.depart_sec).all()
    # a second, unrecorded env on the training kernel sees the same traffic (recording does not perturb the dynamics)
    env2 = TrafficEnv(scn, seed=42, test_seeds=(10000,))
    env2.train_mode = False
    ob = env2.reset(test_ind=0)
    act = _greedy(scn)
    for row in env.control_data[:60]:
        a = act(ob)
        assert ','.join('%d' % x for x in a) == row['action']
        ob, _, _, gr = env2.step(a)
        assert gr == row['reward']
    env.close(); env2.close()


@pytest.mark.gpu
def test_window_mean_live_vehicles():
    import torch
    from deeprl_signal_control_amd.env import VecTrafficEnv
    scn = build_large_grid('ma2c')
    env = VecTrafficEnv(scn, 5, seed=3)
    env.reset()
    g = torch.Generator(device='cuda'); g.manual_seed(0)
    tot = 0.0
    for t in range(40):
        env.step(torch.randint(0, 5, (5, 25), generator=g, device='cuda', dtype=torch.int32))
        tot += env.mean_live_vehicles()
    assert abs(env.live_vehicle_mean(40) - tot / 40) < 1e-9
    assert env.live_vehicle_mean(1) == 0.0                    # accumulator was reset
    env.close()


@pytest.mark.gpu
def test_teleported_trips_are_counted_apart_from_arrivals():
    """A full greedy Monaco episode starves a few heads for time-to-teleport seconds (MICROSIM_SPEC.md): the teleport surrogate
    takes them out of the network.  They are truncated trips, not arrivals: the device counters, the per-second
    number_arrived_car and the trip table (like SUMO's tripinfo file) leave them out and report them apart; the oracle
    agrees on every figure."""
    from deeprl_signal_control_amd.env import TrafficEnv
    from oracle.env_oracle import OracleEnv
    scn = build_real_net('greedy', norm_wave=1.0, clip_wave=-1.0)
    env = TrafficEnv(scn, is_record=True, seed=42, test_seeds=(10000,))
    orc = OracleEnv(scn, seed=42, train_mode=False, test_seeds=(10000,), is_record=True)
    _run(env, scn)
    _run(orc, scn)
    tot = orc.ms.totals()
    arrived, teleported = env.vec.counters()
    assert 50 < tot['teleported'] < 300 and tot['arrived'] > 2000
    assert (int(arrived[0]), int(teleported[0])) == (tot['arrived'], tot['teleported'])
    assert env.vec.teleported_trips[0] == orc.teleported_trips == tot['teleported']
    assert len(env.trip_data) == len(orc.trip_data) == tot['arrived']
    key = lambda r: (float(r['arrival_sec']), r['id'])
    assert sorted(env.trip_data, key=key) == sorted(orc.trip_data, key=key)
    # the truncated trips keep their rows in a table of their own (they are the worst-delayed trips of the episode --
    # every one stood for time-to-teleport seconds -- and must not silently vanish from the evaluation output)
    kc = lambda r: (float(r['removed_sec']), r['id'])
    assert len(env.truncated_trip_data) == tot['teleported']
    assert sorted(env.truncated_trip_data, key=kc) == sorted(orc.truncated_trip_data, key=kc)
    assert min(float(r['wait_sec']) for r in env.truncated_trip_data) >= scn.teleport_sec
    assert np.mean([float(r['duration_sec']) for r in env.truncated_trip_data]) > np.mean([float(r['duration_sec']) for r in env.trip_data])
    assert sum(r['number_arrived_car'] for r in env.traffic_data) == tot['arrived']
    assert [r['number_arrived_car'] for r in env.traffic_data] == [r['number_arrived_car'] for r in orc.traffic_data]
    env.close()
