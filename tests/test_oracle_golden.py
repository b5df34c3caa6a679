"""Pin the oracle restatement (oracle/env_oracle.py + oracle/microsim.c) against
fixtures recorded from the REFERENCE's own env classes (tools/make_golden.py ran
/root/reference/envs/large_grid_env.py unmodified over oracle/fake_traci.py)."""
import json
import os

import numpy as np
import pytest

from deeprl_signal_control_amd.scenario import build_large_grid, yellow_phase, LARGE_GRID_PHASES
from oracle.env_oracle import OracleEnv, numpy_pairwise_sum


def _replay(env, g, prefix='', test_ind=None):
    acts, pols = g[prefix + 'actions'], g[prefix + 'policies']
    ob = env.reset() if test_ind is None else env.reset(test_ind=test_ind)
    np.testing.assert_array_equal(np.concatenate(ob), g[prefix + 'obs'][0])
    for t in range(len(acts)):
        if env.agent == 'ma2c':
            env.update_fingerprint(list(pols[t]))
        ob, r, done, gr = env.step(list(acts[t]))
        np.testing.assert_array_equal(np.concatenate(ob), g[prefix + 'obs'][t + 1], err_msg='obs t=%d' % t)
        np.testing.assert_array_equal(np.asarray(r, np.float64), g[prefix + 'reward'][t], err_msg='reward t=%d' % t)
        assert gr == g[prefix + 'global_reward'][t]
        assert bool(done) == bool(g[prefix + 'done'][t])


def test_static_tables(golden_dir):
    st = json.load(open(os.path.join(golden_dir, 'large_grid_static.json')))
    scn = build_large_grid('ma2c')
    assert scn.node_names == st['node_names']
    assert scn.n_s_ls == st['n_s_ls'] and scn.n_a_ls == st['n_a_ls']
    assert scn.n_w_ls == st['n_w_ls'] and scn.n_f_ls == st['n_f_ls']
    assert build_large_grid('ia2c').n_s_ls == st['ia2c_n_s_ls']
    for a, n in enumerate(scn.node_names):
        assert [scn.node_names[j] for j in scn.neighbors[a]] == st['neighbors'][n]
        assert [scn.lane_names[l] for l in scn.agent_lanes[a]] == st['ilds_in'][n]
        assert [scn.lane_names[l] for l in scn.link_lane[a]] == st['lanes_in'][n]
    for p in range(5):
        assert bytes(scn.green_tab[0, p]).decode() == st['green'][p]
        for q in range(5):
            assert bytes(scn.yellow_tab[0, p, q]).decode() == st['yellow']['%d->%d' % (p, q)]
            if p != q:
                assert yellow_phase(LARGE_GRID_PHASES[p], LARGE_GRID_PHASES[q]) == st['yellow']['%d->%d' % (p, q)]
    flows = [[e1, e2, b, e, v] for (e1, e2, b, e, v) in scn.extra['demand']]
    assert flows == st['flows']                       # 84 elements incl. the %d truncation (461)
    assert len(flows) == 84


def test_ma2c_full_episode_and_reseed(golden_dir):
    g = np.load(os.path.join(golden_dir, 'large_grid_ma2c.npz'))
    env = OracleEnv(build_large_grid('ma2c'), seed=12)
    _replay(env, g, 'ep1_')
    assert g['ep1_done'][-1] and len(g['ep1_done']) == 720
    _replay(env, g, 'ep2_')                            # second reset: seed 13 (env.py:560)


def test_ia2c_global_reward_broadcast(golden_dir):
    g = np.load(os.path.join(golden_dir, 'large_grid_ia2c.npz'))
    _replay(OracleEnv(build_large_grid('ia2c'), seed=12), g)


def test_ma2c_test_mode_local_rewards(golden_dir):
    g = np.load(os.path.join(golden_dir, 'large_grid_ma2c_test.npz'))
    env = OracleEnv(build_large_grid('ma2c'), seed=12, test_seeds=(10000, 20000), train_mode=False)
    _replay(env, g, test_ind=1)


def test_greedy_state(golden_dir):
    g = np.load(os.path.join(golden_dir, 'large_grid_greedy.npz'))
    scn = build_large_grid('greedy', norm_wave=1.0, norm_wait=1.0, clip_wave=1000.0, clip_wait=1000.0,
                           coop_gamma=0.75)
    env = OracleEnv(scn, seed=42, test_seeds=(10000, 20000, 30000), train_mode=False)
    _replay(env, g, test_ind=0)


@pytest.mark.parametrize('tag,kw', [('queue', dict(objective='queue')), ('wait', dict(objective='wait')),
                                    ('norms', dict(norm_wave=3.0, norm_wait=40.0, clip_wave=1.5, clip_wait=1.0, coop_gamma=0.5,
                                                   coef_wait=0.5))])
def test_reward_objectives_and_normalisation_constants(golden_dir, tag, kw):
    """envs/env.py:356-367 'queue' / 'wait' objectives and non-default norm / clip / cooperation constants, recorded from
    the reference LargeGridEnv with those [ENV_CONFIG] values."""
    g = np.load(os.path.join(golden_dir, 'large_grid_ma2c_%s.npz' % tag))
    _replay(OracleEnv(build_large_grid('ma2c', **kw), seed=12), g)
    assert np.abs(g['reward']).max() > 0


def test_demand_tables_for_other_peak_flows(golden_dir):
    """large_grid/data/build_file.py:268-337 with other [ENV_CONFIG] peak flows: every flow element incl. the `%d`
    truncation of the scaled volumes, as the reference generator wrote them."""
    ref = json.load(open(os.path.join(golden_dir, 'large_grid_flow_variants.json')))
    assert len(ref) == 3
    for key, flows in ref.items():
        p1, p2 = (int(v) for v in key.split(','))
        scn = build_large_grid('ma2c', peak_flow1=p1, peak_flow2=p2)
        assert [[e1, e2, b, e, v] for (e1, e2, b, e, v) in scn.extra['demand']] == flows


def test_iql_agents_see_the_ia2c_env(golden_dir):
    """config_iqll_large.ini (agent = iqll): undiscounted neighbour waves, no fingerprints, global reward -- recorded from
    the reference LargeGridEnv; the scenario compiler builds the same tables for iqll / iqld as for ia2c."""
    g = np.load(os.path.join(golden_dir, 'large_grid_iqll.npz'))
    for agent in ('iqll', 'iqld'):
        _replay(OracleEnv(build_large_grid(agent), seed=12), g)


def test_numpy_sum_order():
    rng = np.random.RandomState(0)
    for n in (2, 6, 25, 28):
        for _ in range(300):
            a = -(rng.randint(0, 40, n) + 0.2 * rng.randint(0, 300, n)).astype(np.float64)
            assert numpy_pairwise_sum(a) == np.sum(a)


def test_microsim_invariants():
    """Spacing >= vehicle length, positions inside the lane, conservation of vehicles."""
    from oracle.microsim import MicroSim
    scn = build_large_grid('ma2c')
    m = MicroSim(scn)
    m.reset(3)
    rng = np.random.RandomState(0)
    for t in range(1500):
        if t % 5 == 0:
            for a in range(25):
                m.set_links(a, scn.phases[a][rng.randint(5)])
        m.step()
        assert m.check() == 0
    tot = m.totals()
    assert tot['departed'] == tot['arrived'] + tot['live']
    assert tot['live'] > 100


@pytest.mark.parametrize('name', ['large_grid', 'real_net'])
def test_live_lanes_are_a_prefix_after_load_sorting(name):
    """csrc/tsc_env.hip gives threads / LDS rows only to the lanes a route can put a vehicle on (entry lane ->
    mv_next chain).  That set must be exactly the lanes with static load > 0 and a prefix [0, NU) of the sorted
    numbering; unreachable lanes may be referenced as yield lanes / feeders (the loader drops those references)
    but never as the target of a reachable lane's movement."""
    from deeprl_signal_control_amd.scenario import build_scenario, lane_load
    scn = build_scenario(name, 'ma2c')
    NR = scn.mv_next.shape[1]
    def step(l, r):
        """next lane of a vehicle of route r on lane l: its movement, or (rule 10) the sibling lane that serves it"""
        t = int(scn.mv_next[l][r])
        if t < -1 and scn.lane_sib is not None and scn.lane_sib[l] >= 0 and scn.mv_next[scn.lane_sib[l]][r] >= -1:
            t = int(scn.lane_sib[l])
        return t
    reach = set()
    for r in range(NR):
        l, hops = int(scn.route_entry_lane[r]), 0
        while l >= 0:
            reach.add(l)
            l = step(l, r)
            hops += 1
            assert hops <= 2 * len(scn.lane_len), 'route %d loops' % r
    nu = max(reach) + 1
    assert reach == set(range(nu))
    load = lane_load(scn)
    assert set(np.nonzero(load > 0)[0].tolist()) == reach
    # large_grid: 81 lanes served the routes until round 4; with rule 10 two connection lanes more carry vehicles on their way to
    # the sibling lane (nt2_nt3_1, nt4_nt5_0); real_net: 160 SUMO lanes, 1-to-1 chains contracted
    assert nu == {'large_grid': 83, 'real_net': 113}[name]
    for l in reach:
        for r in range(NR):
            t = int(scn.mv_next[l][r])
            if t >= 0 and any(int(scn.route_entry_lane[q]) >= 0 for q in (r,)):
                # a movement of a route that actually passes l leads to a reachable lane
                chain, c = set(), int(scn.route_entry_lane[r])
                while c >= 0:
                    chain.add(c)
                    c = step(c, r)
                if l in chain:
                    assert t in reach


def test_chain_clamp_closed_form_equals_the_recurrence():
    """MICROSIM_SPEC.md rule 4 / ADVICE r02: the queue clamp behind the first vehicle that stays is evaluated as an
    exclusive prefix-min over keys (x'_i = max(x_i, min(a_i, K_{i-1} - 5 i)), K_i = min(K_{i-1}, a_i + 5 i)) so that the HIP
    kernel can scan it.  On random queues it equals the sequential recurrence x'_i = max(x_i, min(a_i, x'_{i-1} - 5)): with the
    spec's invariant (old spacing >= 5 m, nobody moves backward) the no-backward clamp never binds -- x'_{i-1} - 5 >= x_{i-1} - 5
    >= x_i -- which is the only place where the two forms could part; the test checks that too."""
    rng = np.random.RandomState(4)
    worst = 0.0
    for _ in range(2000):
        n = rng.randint(2, 27)
        gaps = 5.0 + rng.exponential(1.5, n) * (rng.rand(n) < 0.7)
        x = np.float32(180.0) - np.cumsum(gaps).astype(np.float32)          # front first, spacing >= 5 m
        a = np.minimum(x + (rng.rand(n) * 6).astype(np.float32), np.float32(200.0))
        first = np.float32(min(a[0], 200.0))
        seq = [first]                                                        # the sequential recurrence
        for i in range(1, n):
            inner = min(a[i], np.float32(seq[-1] - np.float32(5.0)))
            assert inner >= x[i] - 1e-3                                      # the no-backward clamp cannot bind
            seq.append(max(x[i], inner))
        K = np.float32(first + np.float32(0.0))                              # the closed form (oracle/microsim.c, csrc/tsc_env.hip)
        cf = [first]
        for i in range(1, n):
            cf.append(max(x[i], min(a[i], np.float32(K - np.float32(5 * i)))))
            K = min(K, np.float32(a[i] + np.float32(5 * i)))
        seq, cf = np.array(seq, np.float64), np.array(cf, np.float64)
        assert np.all(cf[:-1] - cf[1:] >= 5.0 - 1e-3)                        # spacing kept
        worst = max(worst, np.abs(cf - seq).max())
    assert worst < 1e-3                                                      # float32 rounding of the keys only
