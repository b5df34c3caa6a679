"""GPU parity of the HIP env path (csrc/tsc_env.hip through the C-ABI) against
(a) fixtures recorded from the reference's own env classes and (b) the CPU oracle.
Bit-exact: obs == float32(reference float64 obs); rewards identical float64."""
import os

import numpy as np
import pytest
import torch

from deeprl_signal_control_amd.scenario import build_large_grid

pytestmark = pytest.mark.gpu


def _replay(env, g, prefix='', test_ind=None):
    acts, pols = g[prefix + 'actions'], g[prefix + 'policies']
    ob = env.reset() if test_ind is None else env.reset(test_ind=test_ind)
    np.testing.assert_array_equal(np.concatenate(ob), g[prefix + 'obs'][0].astype(np.float32))
    for t in range(len(acts)):
        if env.agent == 'ma2c':
            env.update_fingerprint(list(pols[t]))
        ob, r, done, gr = env.step(list(acts[t]))
        np.testing.assert_array_equal(np.concatenate(ob), g[prefix + 'obs'][t + 1].astype(np.float32),
                                      err_msg='obs t=%d' % t)
        np.testing.assert_array_equal(np.asarray(r, np.float64), g[prefix + 'reward'][t], err_msg='reward t=%d' % t)
        assert gr == g[prefix + 'global_reward'][t], t
        assert bool(done) == bool(g[prefix + 'done'][t])


@pytest.mark.parametrize('threads', ['256', '512', '1024'])
def test_golden_ma2c_full_episode(golden_dir, threads, monkeypatch):
    """Every workgroup size of the specialised step kernel (csrc/tsc_env.hip picks 1024 / 512 / 256 threads per instance for
    E <= 256 / <= 512 / more; 256 is what the benchmarked 1024 instances run) replays the reference's episode bit-exactly."""
    from deeprl_signal_control_amd.env import TrafficEnv
    monkeypatch.setenv('TSC_ENV_THREADS', threads)
    g = np.load(os.path.join(golden_dir, 'large_grid_ma2c.npz'))
    env = TrafficEnv(build_large_grid('ma2c'), seed=12)
    _replay(env, g, 'ep1_')
    _replay(env, g, 'ep2_')
    env.close()


def test_golden_ia2c(golden_dir):
    from deeprl_signal_control_amd.env import TrafficEnv
    g = np.load(os.path.join(golden_dir, 'large_grid_ia2c.npz'))
    env = TrafficEnv(build_large_grid('ia2c'), seed=12)
    _replay(env, g)
    env.close()


@pytest.mark.parametrize('tag,kw', [('queue', dict(objective='queue')), ('wait', dict(objective='wait')),
                                    ('norms', dict(norm_wave=3.0, norm_wait=40.0, clip_wave=1.5, clip_wait=1.0, coop_gamma=0.5,
                                                   coef_wait=0.5))])
def test_golden_reward_objectives_and_normalisation_constants(golden_dir, tag, kw):
    """envs/env.py:356-367 'queue' / 'wait' objectives and non-default norm / clip / cooperation constants through
    step_kernel's K5 / K6 branches -- fixtures from the reference LargeGridEnv with those [ENV_CONFIG] values."""
    from deeprl_signal_control_amd.env import TrafficEnv
    g = np.load(os.path.join(golden_dir, 'large_grid_ma2c_%s.npz' % tag))
    env = TrafficEnv(build_large_grid('ma2c', **kw), seed=12)
    _replay(env, g)
    env.close()
    assert np.abs(g['reward']).max() > 0


def test_golden_iql_agents_see_the_ia2c_env(golden_dir):
    """config_iqll_large.ini (agent = iqll; iqld alike): undiscounted neighbour waves, no fingerprints, global reward."""
    from deeprl_signal_control_amd.env import TrafficEnv
    g = np.load(os.path.join(golden_dir, 'large_grid_iqll.npz'))
    for agent in ('iqll', 'iqld'):
        env = TrafficEnv(build_large_grid(agent), seed=12)
        _replay(env, g)
        env.close()


def test_golden_test_mode_and_greedy(golden_dir):
    from deeprl_signal_control_amd.env import TrafficEnv
    g = np.load(os.path.join(golden_dir, 'large_grid_ma2c_test.npz'))
    env = TrafficEnv(build_large_grid('ma2c'), seed=12, test_seeds=(10000, 20000))
    env.train_mode = False
    _replay(env, g, test_ind=1)
    env.close()
    g = np.load(os.path.join(golden_dir, 'large_grid_greedy.npz'))
    scn = build_large_grid('greedy', norm_wave=1.0, norm_wait=1.0, clip_wave=1000.0, clip_wait=1000.0,
                           coop_gamma=0.75)
    env = TrafficEnv(scn, seed=42, test_seeds=(10000, 20000, 30000))
    env.train_mode = False
    _replay(env, g, test_ind=0)
    env.close()


@pytest.mark.parametrize('E,steps,p_random,threads', [(48, 150, 0.7, '256'), (48, 150, 0.7, '512'), (48, 150, 0.7, '1024'),
                                                      (8, 720, 1.0, '256'), (8, 720, 1.0, '')])
def test_batched_vs_oracle_state(E, steps, p_random, threads, monkeypatch):
    """E env instances with different seeds vs E independent oracle instances: obs, rewards and
    the full vehicle state (positions, speeds, waits, routes) must be identical -- for every workgroup size of the step
    kernel ('' = the library's own choice for this E)."""
    from deeprl_signal_control_amd.env import VecTrafficEnv
    from oracle.env_oracle import OracleEnv, greedy_large_grid
    if threads:
        monkeypatch.setenv('TSC_ENV_THREADS', threads)
    scn = build_large_grid('ma2c')
    env = VecTrafficEnv(scn, E, seed=100)
    orc = [OracleEnv(scn, seed=100 + e) for e in range(E)]
    obs = env.reset().cpu().numpy()
    oobs = [o.reset() for o in orc]
    rng = np.random.RandomState(E)
    for t in range(steps):
        pol = rng.dirichlet(np.ones(5), size=(E, 25)).astype(np.float32)
        act = np.zeros((E, 25), np.int32)
        for e in range(E):
            for a in range(25):
                act[e, a] = rng.randint(5) if rng.rand() < p_random else greedy_large_grid(oobs[e][a][:6])
        env.update_fingerprint(torch.from_numpy(pol).cuda())
        o, r, d, g = env.step(torch.from_numpy(act).cuda())
        o, r, d, g = o.cpu().numpy(), r.cpu().numpy(), d.cpu().numpy(), g.cpu().numpy()
        for e in range(E):
            orc[e].update_fingerprint(list(pol[e]))
            oo, orr, od, og = orc[e].step(list(act[e]))
            oobs[e] = oo
            for a in range(25):
                np.testing.assert_array_equal(o[e, a, :scn.n_s_ls[a]], oo[a].astype(np.float32),
                                              err_msg='t=%d e=%d a=%d' % (t, e, a))
            np.testing.assert_array_equal(r[e], orr, err_msg='t=%d e=%d' % (t, e))
            assert g[e] == og and bool(d[e]) == bool(od)
    for e in range(0, E, max(1, E // 6)):
        st, sn = env.get_state(e), orc[e].ms.snapshot()
        for k in ('n', 'x', 'v', 'sf', 'w', 'r'):
            np.testing.assert_array_equal(st[k], sn[k], err_msg='state %s e=%d' % (k, e))
    assert env.mean_live_vehicles() > 50
    env.close()


def test_zero_copy_fingerprint_and_reward_sum():
    """tsc_env_bind_fingerprint reads the caller's policy buffer in place: same obs as the copying path."""
    from deeprl_signal_control_amd.env import VecTrafficEnv
    scn = build_large_grid('ma2c')
    E = 6
    a, b = VecTrafficEnv(scn, E, seed=50), VecTrafficEnv(scn, E, seed=50)
    a.reset(); b.reset()
    rng = np.random.RandomState(0)
    tot = 0.0
    for t in range(30):
        pol = torch.from_numpy(rng.dirichlet(np.ones(5), size=(E, 25)).astype(np.float32)).cuda()
        act = torch.from_numpy(rng.randint(0, 5, (E, 25)).astype(np.int32)).cuda()
        a.update_fingerprint(pol)
        b.update_fingerprint(pol, zero_copy=True)
        oa, ra, _, ga = a.step(act)
        ob, rb, _, gb = b.step(act)
        assert torch.equal(oa, ob) and torch.equal(ra, rb)
        tot += float(ga.sum().item())
    assert abs(a.reward_sum() - tot) < 1e-6 * max(1.0, abs(tot))
    a.close(); b.close()


def test_helper_threads_equal_plain_walk(monkeypatch):
    """step_kernel<., true> (phase A1: car-following of queued vehicles on all threads, chunk-prefetched tail walk)
    against step_kernel<., false> (every vehicle evaluated by its lane thread) at a load where most lanes are
    queued: obs, rewards and the complete vehicle state stay bit-identical."""
    from deeprl_signal_control_amd.env import VecTrafficEnv
    scn = build_large_grid('ma2c')
    E = 96
    monkeypatch.setenv('TSC_ENV_HELP', '1')
    a = VecTrafficEnv(scn, E, seed=7)
    monkeypatch.setenv('TSC_ENV_HELP', '0')
    b = VecTrafficEnv(scn, E, seed=7)
    oa, ob = a.reset(), b.reset()
    assert torch.equal(oa, ob)
    g = torch.Generator(device='cuda'); g.manual_seed(3)
    for t in range(400):                                     # random phases: heavy congestion by t = 2000 s
        act = torch.randint(0, 5, (E, 25), generator=g, device='cuda', dtype=torch.int32)
        pol = torch.rand(E, 25, 5, generator=g, device='cuda')
        a.update_fingerprint(pol); b.update_fingerprint(pol)
        oa, ra, da, ga = a.step(act)
        ob, rb, db, gb = b.step(act)
        assert torch.equal(oa, ob) and torch.equal(ra, rb) and torch.equal(ga, gb) and torch.equal(da, db), t
    assert a.mean_live_vehicles() > 400
    for e in (0, 17, 95):
        sa, sb = a.get_state(e), b.get_state(e)
        for k in ('n', 'x', 'v', 'sf', 'w', 'r'):
            np.testing.assert_array_equal(sa[k], sb[k], err_msg='state %s e=%d' % (k, e))
    a.close(); b.close()


@pytest.mark.parametrize('kf,spec,threads', [('1', '1', '256'), ('2', '1', '256'), ('4', '1', '256'), ('1', '0', '256'), ('2', '0', '256'),
                                             ('1', '1', '512'), ('2', '1', '512'), ('1', '1', '1024'), ('2', '1', '1024')])
def test_flat_phase_super_rounds_at_saturation(kf, spec, threads, monkeypatch):
    """More queued vehicles than one super-round of the flat phase holds (kF x 256 per round): tripled demand and random
    phases fill most lanes to capacity.  The chain scan then crosses wavefronts and super-rounds (LDS carries, the saved
    old state of the previous round's last vehicle); obs, rewards and the full vehicle state stay bit-identical to the
    oracle's sequential walk, for every kF variant of the kernel, with the table dimensions as compile-time constants
    (the large_grid instantiation, spec 1) and as launch parameters (spec 0: what Monaco and small_grid run)."""
    from deeprl_signal_control_amd.env import VecTrafficEnv
    from oracle.env_oracle import OracleEnv
    monkeypatch.setenv('TSC_ENV_KF', kf)
    monkeypatch.setenv('TSC_ENV_SPEC', spec)
    monkeypatch.setenv('TSC_ENV_THREADS', threads)          # super-rounds of kF x threads vehicles; 256 = the benchmarked kernel
    scn = build_large_grid('ma2c', peak_flow1=3300, peak_flow2=2800)
    E = 6
    env = VecTrafficEnv(scn, E, seed=9)
    orc = [OracleEnv(scn, seed=9 + e) for e in range(E)]
    env.reset()
    for o in orc:
        o.reset()
    rng = np.random.RandomState(5)
    peak = 0.0
    for t in range(260):
        act = rng.randint(0, 5, (E, 25)).astype(np.int32)
        o, r, d, g = env.step(torch.from_numpy(act).cuda())
        o, r = o.cpu().numpy(), r.cpu().numpy()
        for e in range(E):
            oo, orr, _, _ = orc[e].step(list(act[e]))
            for a in range(25):
                np.testing.assert_array_equal(o[e, a, :scn.n_s_ls[a]], oo[a].astype(np.float32), err_msg='t=%d e=%d a=%d' % (t, e, a))
            np.testing.assert_array_equal(r[e], orr, err_msg='t=%d e=%d' % (t, e))
        peak = max(peak, env.mean_live_vehicles())
    assert peak > 600, peak                      # 256 threads: 3 super-rounds at kF = 1, 2 at kF = 2 (entry lanes cap the inflow)
    for e in range(E):
        st, sn = env.get_state(e), orc[e].ms.snapshot()
        for k in ('n', 'x', 'v', 'sf', 'w', 'r'):
            np.testing.assert_array_equal(st[k], sn[k], err_msg='state %s e=%d' % (k, e))
    env.close()


def test_resident_instances_hint_changes_the_launch_shape_not_the_results():
    """tsc_env_set_resident_instances: a handle of 64 instances on an otherwise empty device spreads every instance
    over 1024 threads; told that 2048 instances share the device it runs the 256-thread workgroups of a full device.  Same
    observations, rewards and vehicle state bit for bit (the parity tests above pin each size to the oracle)."""
    from deeprl_signal_control_amd.env import VecTrafficEnv
    scn = build_large_grid('ma2c')
    E = 64
    envs = [VecTrafficEnv(scn, E, seed=31), VecTrafficEnv(scn, E, seed=31, resident=2048)]
    rng = np.random.RandomState(4)
    for e_ in envs:
        e_.reset()
    for t in range(40):
        act = torch.from_numpy(rng.randint(0, 5, (E, 25)).astype(np.int32)).cuda()
        outs = [e_.step(act) for e_ in envs]
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    s0, s1 = envs[0].get_state(5), envs[1].get_state(5)
    for k in ('n', 'x', 'v', 'sf', 'w', 'r'):
        np.testing.assert_array_equal(s0[k], s1[k])
    for e_ in envs:
        e_.close()


@pytest.mark.parametrize('threads,spec,help_', [('256', '1', '1'), ('512', '1', '1'), ('1024', '1', '1'), ('256', '0', '1'), ('256', '1', '0')])
def test_lane_change_rule_vs_oracle(threads, spec, help_, monkeypatch):
    """MICROSIM_SPEC.md rule 10 (lane choice by the junction's connections + lane changes on the two-lane streets, lane_change=True):
    the step kernel against the CPU oracle over a whole greedy-driven episode of four instances -- obs, rewards, done, and
    the complete vehicle state at the demand peak and at the end -- with lane changers in the head platoons, wrong-lane heads
    lining up behind the sibling's queue, teleported changers, for every workgroup size, with and without the compile-time
    table dimensions and the flat phase."""
    from deeprl_signal_control_amd.env import VecTrafficEnv
    from oracle.env_oracle import OracleEnv, greedy_large_grid
    monkeypatch.setenv('TSC_ENV_THREADS', threads)
    monkeypatch.setenv('TSC_ENV_SPEC', spec)
    monkeypatch.setenv('TSC_ENV_HELP', help_)
    scn = build_large_grid('ma2c', lane_change=True)
    assert scn.lane_sib is not None and (scn.lane_sib >= 0).sum() == 120
    E = 4
    env = VecTrafficEnv(scn, E, seed=300)
    orc = [OracleEnv(scn, seed=300 + e) for e in range(E)]
    obs = env.reset().cpu().numpy()
    oobs = [o.reset() for o in orc]
    rng = np.random.RandomState(11)
    for t in range(720):
        pol = rng.dirichlet(np.ones(5), size=(E, 25)).astype(np.float32)
        act = np.zeros((E, 25), np.int32)
        for e in range(E):
            for a in range(25):
                act[e, a] = rng.randint(5) if rng.rand() < 0.1 * e else greedy_large_grid(oobs[e][a][:6])
        env.update_fingerprint(torch.from_numpy(pol).cuda())
        o, r, d, g = env.step(torch.from_numpy(act).cuda())
        o, r, d, g = o.cpu().numpy(), r.cpu().numpy(), d.cpu().numpy(), g.cpu().numpy()
        for e in range(E):
            orc[e].update_fingerprint(list(pol[e]))
            oo, orr, od, og = orc[e].step(list(act[e]))
            oobs[e] = oo
            for a in range(25):
                np.testing.assert_array_equal(o[e, a, :scn.n_s_ls[a]], oo[a].astype(np.float32), err_msg='t=%d e=%d a=%d' % (t, e, a))
            np.testing.assert_array_equal(r[e], orr, err_msg='t=%d e=%d' % (t, e))
            assert g[e] == og and bool(d[e]) == bool(od)
        if t in (299, 719):
            for e in range(E):
                st, sn = env.get_state(e), orc[e].ms.snapshot()
                for k in ('n', 'x', 'v', 'sf', 'w', 'r'):
                    np.testing.assert_array_equal(st[k], sn[k], err_msg='state %s t=%d e=%d' % (k, t, e))
    arrived, teleported = env.counters()
    tot = [o.ms.totals() for o in orc]
    assert [int(x) for x in arrived] == [x['arrived'] for x in tot] and [int(x) for x in teleported] == [x['teleported'] for x in tot]
    assert sum(o.ms.lanechange_counts()['changes'] for o in orc) > 2000 and sum(x['teleported'] for x in tot) > 0
    env.close()


def test_block_order_never_changes_a_result():
    """tsc_env_set_block_order (round 6 tuning hook): which workgroup simulates which instance is a launch-geometry choice; obs,
    rewards, dones and the per-instance vehicle counts of a run under a random permutation equal the identity run's bit for bit,
    and a non-permutation is refused."""
    import ctypes as C
    from deeprl_signal_control_amd import _lib
    from deeprl_signal_control_amd.env import VecTrafficEnv
    scn = build_large_grid('ma2c', episode_length_sec=600)
    E = 48
    envs = [VecTrafficEnv(scn, E, seed=5) for _ in range(2)]
    perm = np.random.RandomState(3).permutation(E).astype(np.int32)
    _lib.check(envs[1]._L.tsc_env_set_block_order(envs[1]._h, perm.ctypes.data_as(C.c_void_p)))
    bad = perm.copy(); bad[0] = bad[1]
    assert envs[1]._L.tsc_env_set_block_order(envs[1]._h, bad.ctypes.data_as(C.c_void_p)) != 0
    obs = [e.reset().clone() for e in envs]
    assert torch.equal(obs[0], obs[1])
    g = torch.Generator(device='cuda'); g.manual_seed(0)
    for t in range(120):
        act = torch.randint(0, 5, (E, 25), generator=g, device='cuda', dtype=torch.int32)
        outs = [e.step(act) for e in envs]
        for a, b in zip(outs[0][:3], outs[1][:3]):
            assert torch.equal(a, b), t
    cnt = [np.zeros(E, np.int32) for _ in envs]
    for e, c in zip(envs, cnt):
        _lib.check(e._L.tsc_env_vehicle_counts(e._h, c.ctypes.data_as(C.c_void_p)))
    np.testing.assert_array_equal(cnt[0], cnt[1])
    assert cnt[0].min() > 0 and abs(cnt[0].mean() - envs[0].mean_live_vehicles()) < 1e-9
    _lib.check(envs[1]._L.tsc_env_set_block_order(envs[1]._h, None))
    for e in envs:
        e.close()
