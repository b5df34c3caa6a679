"""Monaco (real_net): tables compiled from most.net.xml and the oracle restatement, pinned against
fixtures recorded from the reference's own RealNetEnv (tools/make_golden.py, envs/real_net_env.py
unmodified over oracle/fake_traci.py); GPU parity of the HIP env path on the same fixtures."""
import json
import os

import numpy as np
import pytest

from deeprl_signal_control_amd.scenario import build_real_net


def _replay(env, g, n_a_ls, test_ind=None, f32=False):
    acts, pols = g['actions'], g['policies']
    ob = env.reset() if test_ind is None else env.reset(test_ind=test_ind)
    cast = (lambda x: x.astype(np.float32)) if f32 else (lambda x: x)
    np.testing.assert_array_equal(np.concatenate(ob), cast(g['obs'][0]))
    for t in range(len(acts)):
        if env.agent == 'ma2c':
            env.update_fingerprint([pols[t][a, :n] for a, n in enumerate(n_a_ls)])
        ob, r, done, gr = env.step(list(acts[t]))
        np.testing.assert_array_equal(np.concatenate(ob), cast(g['obs'][t + 1]), err_msg='obs t=%d' % t)
        np.testing.assert_array_equal(np.asarray(r, np.float64), g['reward'][t], err_msg='reward t=%d' % t)
        assert gr == g['global_reward'][t] and bool(done) == bool(g['done'][t])


def test_static_tables(golden_dir):
    st = json.load(open(os.path.join(golden_dir, 'real_net_static.json')))
    scn = build_real_net('ma2c')
    assert scn.node_names == st['node_names'] and len(scn.node_names) == 28      # 28, not 30
    assert scn.n_s_ls == st['n_s_ls'] and scn.n_a_ls == st['n_a_ls']
    assert scn.n_w_ls == st['n_w_ls'] == [0] * 28 and scn.n_f_ls == st['n_f_ls']
    assert build_real_net('ia2c').n_s_ls == st['ia2c_n_s_ls']
    for a, n in enumerate(scn.node_names):
        assert [scn.node_names[j] for j in scn.neighbors[a]] == st['neighbors'][n]
        assert [scn.lane_names[l] for l in scn.agent_lanes[a, :scn.agent_nlane[a]]] == st['ilds_in'][n]
        assert [scn.lane_names[l] for l in scn.link_lane[a, :scn.agent_nlink[a]]] == st['lanes_in'][n]
        k = scn.n_a_ls[a]
        for p in range(k):
            for q in range(k):
                assert bytes(scn.yellow_tab[a, p, q, :scn.agent_nlink[a]]).decode() == st['yellow']['%s:%d->%d' % (n, p, q)]
    # demand: 88 flow elements at flow_rate veh/h, same (from, to) and windows as the reference generator
    assert len(scn.flows) == len(st['flows']) == 88
    for (b, e, vph, r), (frm, to, via, b2, e2, v2) in zip(scn.flows, st['flows']):
        assert (b, e, vph) == (b2, e2, v2) and scn.route_names[r] == (frm, to)
        assert all(v in scn.extra['routes'][r] for v in via.split())


def test_oracle_matches_reference_env(golden_dir):
    from oracle.env_oracle import OracleEnv
    scn = build_real_net('ma2c')
    _replay(OracleEnv(scn, seed=42, test_seeds=(10000, 20000, 30000)),
            np.load(os.path.join(golden_dir, 'real_net_ma2c.npz')), scn.n_a_ls)
    scn = build_real_net('ia2c')
    _replay(OracleEnv(scn, seed=42, test_seeds=(10000, 20000, 30000)),
            np.load(os.path.join(golden_dir, 'real_net_ia2c.npz')), scn.n_a_ls)
    scn = build_real_net('ma2c')
    env = OracleEnv(scn, seed=42, test_seeds=(10000, 20000, 30000), train_mode=False)
    _replay(env, np.load(os.path.join(golden_dir, 'real_net_ma2c_test.npz')), scn.n_a_ls, test_ind=2)


def test_microsim_invariants_monaco():
    from oracle.microsim import MicroSim
    scn = build_real_net('ma2c')
    m = MicroSim(scn)
    m.reset(5)
    rng = np.random.RandomState(1)
    for t in range(1800):
        if t % 5 == 0:
            for a in range(28):
                m.set_links(a, scn.phases[a][rng.randint(scn.n_a_ls[a])])
        m.step()
        assert m.check() == 0, t
    tot = m.totals()
    assert tot['departed'] == tot['arrived'] + tot['live'] and tot['arrived'] > 100


@pytest.mark.gpu
def test_gpu_golden_monaco(golden_dir):
    from deeprl_signal_control_amd.env import TrafficEnv
    scn = build_real_net('ma2c')
    env = TrafficEnv(scn, seed=42, test_seeds=(10000, 20000, 30000))
    _replay(env, np.load(os.path.join(golden_dir, 'real_net_ma2c.npz')), scn.n_a_ls, f32=True)
    env.close()
    scn = build_real_net('ia2c')
    env = TrafficEnv(scn, seed=42, test_seeds=(10000, 20000, 30000))
    _replay(env, np.load(os.path.join(golden_dir, 'real_net_ia2c.npz')), scn.n_a_ls, f32=True)
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize('threads', ['256', '512', '1024'])
def test_gpu_batched_vs_oracle_monaco(threads, monkeypatch):
    import torch
    monkeypatch.setenv('TSC_ENV_THREADS', threads)           # every workgroup size of the Monaco instantiation (spec 2)
    from deeprl_signal_control_amd.env import VecTrafficEnv
    from oracle.env_oracle import OracleEnv
    scn = build_real_net('ma2c')
    E, A = 12, 28
    env = VecTrafficEnv(scn, E, seed=300, test_seeds=(10000, 20000, 30000))
    orc = [OracleEnv(scn, seed=300 + e, test_seeds=(10000, 20000, 30000)) for e in range(E)]
    env.reset()
    for o in orc:
        o.reset()
    rng = np.random.RandomState(4)
    act = np.zeros((E, A), np.int32)
    for t in range(400):
        pol = np.zeros((E, A, 6), np.float32)
        for a, n in enumerate(scn.n_a_ls):
            pol[:, a, :n] = rng.dirichlet(np.ones(n), size=E)
            change = rng.rand(E) < 0.25
            act[change, a] = rng.randint(0, n, change.sum())
        env.update_fingerprint(torch.from_numpy(pol).cuda())
        o, r, d, g = env.step(torch.from_numpy(act).cuda())
        o, r, g = o.cpu().numpy(), r.cpu().numpy(), g.cpu().numpy()
        for e in range(E):
            orc[e].update_fingerprint([pol[e, a, :n] for a, n in enumerate(scn.n_a_ls)])
            oo, orr, od, og = orc[e].step(list(act[e]))
            for a in range(A):
                np.testing.assert_array_equal(o[e, a, :scn.n_s_ls[a]], oo[a].astype(np.float32), err_msg='t=%d e=%d a=%d' % (t, e, a))
            np.testing.assert_array_equal(r[e], orr, err_msg='t=%d e=%d' % (t, e))
            assert g[e] == og
    for e in (0, 5, 11):
        st, sn = env.get_state(e), orc[e].ms.snapshot()
        for k in ('n', 'x', 'v', 'sf', 'w', 'r'):
            np.testing.assert_array_equal(st[k], sn[k], err_msg='state %s e=%d' % (k, e))
    env.close()


def test_greedy_controller_matches_reference(golden_dir):
    """The controller tables (Scenario.greedy_controller_tables: phase -> lanes with a 'G' link in link order, every lane
    once) through the host restatement trainer.greedy_actions (wave sums in table order -> first argmax) against
    RealNetController.greedy of the reference (tools/make_golden.py:greedy_fixtures).  The device kernel reads the same
    tables: tests/test_binding_gpu.py::test_greedy_controllers_on_device.  (large_grid has its own hard-coded controller,
    pinned by tests/golden/large_grid_greedy.npz.)"""
    from deeprl_signal_control_amd.trainer import greedy_actions
    g = np.load(os.path.join(golden_dir, 'real_net_greedy_controller.npz'))
    scn = build_real_net('greedy')
    np.testing.assert_array_equal(greedy_actions(scn, g['wave']), g['action'])


def test_lane_chain_contraction_preserves_routes_and_detectors():
    """scenario.contract_chains: merged lanes keep the SUMO lane's name / signal / detector (x >= det_start is the original
    lane), every route keeps its length and its signalised links, nothing longer than LANE_CAP - MAX_CROSS vehicles."""
    from deeprl_signal_control_amd.scenario import LANE_CAP, MAX_CROSS
    raw, scn = build_real_net('ma2c', contract=False, sort_lanes=False), build_real_net('ma2c', sort_lanes=False)
    assert scn.n_s_ls == raw.n_s_ls and scn.n_lane < raw.n_lane and len(scn.extra['contracted']) == raw.n_lane - scn.n_lane
    rid = {n: i for i, n in enumerate(raw.lane_names)}

    def walk(s, r):
        l, tot, links, names = int(s.route_entry_lane[r]), 0.0, [], []
        while l >= 0:
            tot += float(s.lane_len[l]); names.append(s.lane_names[l])
            if s.lane_node[l] >= 0 and s.mv_link[l, r] >= 0:
                links.append((int(s.lane_node[l]), int(s.mv_link[l, r])))
            l = int(s.mv_next[l, r])
        return tot, links, names
    for r in range(scn.n_route):
        t0, k0, n0 = walk(raw, r)
        t1, k1, n1 = walk(scn, r)
        assert abs(t0 - t1) < 1e-2 and k0 == k1 and [n for n in n0 if n not in scn.extra['contracted']] == n1
    for l, name in enumerate(scn.lane_names):
        assert abs((scn.lane_len[l] - scn.lane_det_start[l]) - raw.lane_len[rid[name]]) < 1e-3      # detector = the SUMO lane
        assert scn.lane_len[l] <= (LANE_CAP - MAX_CROSS) * 7.5 or scn.lane_det_start[l] == 0
    for a in range(scn.n_agent):                                      # observed / signalised lanes are never merged away
        assert [scn.lane_names[l] for l in scn.agent_lanes[a, :scn.agent_nlane[a]]] == \
               [raw.lane_names[l] for l in raw.agent_lanes[a, :raw.agent_nlane[a]]]
