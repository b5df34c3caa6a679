"""small_grid (BASELINE configs[0], envs/small_grid_env.py + small_grid/data/build_file.py): tables pinned against the
reference's own SmallGridEnv / SmallGridController run unmodified over the fake TraCI (tools/make_golden.py
small_grid_fixtures; greedy is the only agent the reference can run there), the oracle and the HIP env on the
same fixture, and the IA2C learner ('npc' dropped from the neighbour lists) on the device."""
import json
import os

import numpy as np
import pytest

from deeprl_signal_control_amd.scenario import build_small_grid, small_grid_demand
from deeprl_signal_control_amd.trainer import greedy_actions

KW = dict(norm_wave=1.0, norm_wait=1.0, clip_wave=1000.0, clip_wait=1000.0, coop_gamma=0.75)


def test_static_tables_and_demand(golden_dir):
    st = json.load(open(os.path.join(golden_dir, 'small_grid_static.json')))
    scn = build_small_grid('greedy', **KW)
    assert scn.node_names == st['node_names'] and scn.n_a_ls == st['n_a_ls'] == [3, 2, 2, 2, 2, 2]
    assert scn.n_s_ls == st['n_s_ls'] and scn.n_w_ls == st['n_w_ls'] and st['T'] == 720
    for a, n in enumerate(scn.node_names):
        assert [scn.lane_names[l] for l in scn.agent_lanes[a, :scn.agent_nlane[a]]] == st['ilds_in'][n]
        assert [scn.lane_names[l] for l in scn.link_lane[a, :scn.agent_nlink[a]]] == st['lanes_in'][n]
        for p in range(scn.n_a_ls[a]):
            for q in range(scn.n_a_ls[a]):
                assert bytes(scn.yellow_tab[a, p, q, :scn.agent_nlink[a]]).decode() == st['yellow']['%s:%d->%d' % (n, p, q)]
    # demand: every source flow element of build_file.py:191-210 is split over its paths without losing a vehicle
    src = {'np1_nt1': [500, 100, 700, 800, 550, 550], 'np2_nt1': [600, 700, 100, 200, 50, 100], 'np3_nt1': [100, 400, 400, 200, 600, 550],
           'np8_nt4': [100, 200, 300, 300, 300, 400], 'np9_nt4': [600, 400, 400, 600, 800, 400]}
    dem = small_grid_demand(1000)
    for s, vols in src.items():
        for i, v in enumerate(vols):
            assert sum(vph for path, tb, te, vph in dem if path[0] == s and tb == 600 * i) == v
    mf = [d for d in dem if not d[0][0].startswith('np')]
    assert len(mf) == 9 and all(v == 1008 and te - tb == 1200 for _, tb, te, v in mf)       # probability "0.28" per second
    assert scn.n_route == 32
    # the env draws every vehicle's turns (random_turns, the default): one flow element per source flow of the reference
    # file; the expectation routing of round 2 (random_turns=False) has one element per (source flow, path)
    assert len(scn.flows) == 5 * 6 + 9 and scn.n_stream == 5 + 6
    assert len(build_small_grid('greedy', random_turns=False, **KW).flows) == len(dem)
    # MARL agents: 'npc' is not a signal node (the reference crashes on it) -> dropped
    ia = build_small_grid('ia2c')
    assert ia.neighbors == [[1, 5], [0, 2], [1, 3], [2, 4], [3, 5], [0, 4]] and ia.n_s_ls == [10, 9, 8, 8, 8, 9] and ia.n_f_ls == [0] * 6


def _replay(env, g):
    env.train_mode = False
    ob = env.reset(test_ind=0)
    np.testing.assert_array_equal(np.concatenate(ob), g['obs'][0].astype(np.float32))
    scn = env.scn
    for t in range(len(g['actions'])):
        o = np.zeros((scn.n_agent, scn.s_max))
        for a, x in enumerate(ob):
            o[a, :len(x)] = x
        act = greedy_actions(scn, o)
        np.testing.assert_array_equal(act, g['actions'][t], err_msg='SmallGridController t=%d' % t)
        ob, r, done, gr = env.step(list(act))
        np.testing.assert_array_equal(np.concatenate(ob).astype(np.float32), g['obs'][t + 1].astype(np.float32), err_msg='obs t=%d' % t)
        np.testing.assert_array_equal(np.asarray(r, np.float64), g['reward'][t])
        assert gr == g['global_reward'][t] and bool(done) == bool(g['done'][t])


def test_oracle_matches_reference_env(golden_dir):
    from oracle.env_oracle import OracleEnv
    scn = build_small_grid('greedy', **KW)
    _replay(OracleEnv(scn, seed=42, test_seeds=(10000, 20000, 30000)), np.load(os.path.join(golden_dir, 'small_grid_greedy.npz')))


@pytest.mark.gpu
def test_gpu_golden_small_grid(golden_dir):
    from deeprl_signal_control_amd.env import TrafficEnv
    scn = build_small_grid('greedy', **KW)
    env = TrafficEnv(scn, seed=42, test_seeds=(10000, 20000, 30000))
    _replay(env, np.load(os.path.join(golden_dir, 'small_grid_greedy.npz')))
    env.close()


@pytest.mark.gpu
def test_gpu_ia2c_small_grid_env_and_learner():
    """IA2C on small_grid (BASELINE configs[0] names it): batched HIP env vs the oracle (obs, rewards, vehicle state) under
    the policy's own sampled actions, then two A2C updates vs the float64 learner oracle."""
    import torch
    from deeprl_signal_control_amd.agents import VecA2C
    from deeprl_signal_control_amd.env import VecTrafficEnv
    from oracle.env_oracle import OracleEnv
    from oracle.nets_oracle import OracleA2C
    scn = build_small_grid('ia2c')
    E, T, A = 6, 10, 6
    env = VecTrafficEnv(scn, E, seed=70, test_seeds=(10000,))
    orc = [OracleEnv(scn, seed=70 + e) for e in range(E)]
    m = VecA2C(scn.n_s_ls, scn.n_a_ls, scn.n_w_ls, scn.n_f_ls, E, scn.s_max, 3, dict(batch_size=T, reward_norm=3000.0), seed=1, name='ia2c')
    o = OracleA2C(m.get_tower_params(), m.n_wave_ls, m.n_w_ls, m.n_f_ls, m.n_a_ls, E, reward_norm=3000.0)
    assert m.H == 160 and m.s_max == 12
    ob = env.reset()
    for x in orc:
        x.reset()
    m.reset(); o.reset()
    done = torch.ones(E, dtype=torch.uint8, device='cuda')
    for it in range(2):
        for t in range(T):
            pi, v, act = m.forward_sample(ob, done)
            o.forward(ob.cpu().numpy(), done.cpu().numpy(), 'pv')
            nob, r, dpost, g = env.step(act)
            a_np = act.cpu().numpy()
            for e in range(E):
                oo, orr, od, og = orc[e].step(list(a_np[e]))
                for a in range(A):
                    np.testing.assert_array_equal(nob[e, a, :scn.n_s_ls[a]].cpu().numpy(), oo[a].astype(np.float32))
                np.testing.assert_array_equal(r[e].cpu().numpy(), orr)
            m.add_transition(ob, done, act, r, v, dpost)
            o.add_transition(ob.cpu().numpy(), done.cpu().numpy(), a_np, r.cpu().numpy(), v.cpu().numpy(), dpost.cpu().numpy())
            ob, done = nob, dpost
        R = m.forward(ob, False, 'v').clone()
        m.backward(R)
        og, _ = o.compute_grads(R.cpu().numpy(), 0.01)
        o.apply_grads(og, 5e-4)
        p, op = m.get_tower_params(), o.tower_params()
        for t in range(m.G):
            for k in op[t]:
                np.testing.assert_allclose(p[t][k], op[t][k], atol=3e-5, err_msg='it=%d tower=%d %s' % (it, t, k))
    for e in (0, E - 1):
        st, sn = env.get_state(e), orc[e].ms.snapshot()
        for k in ('n', 'x', 'v', 'sf', 'w', 'r'):
            np.testing.assert_array_equal(st[k], sn[k])
    env.close(); m.close()
