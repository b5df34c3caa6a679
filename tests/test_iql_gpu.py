"""GPU parity of the HIP Q-learner (csrc/tsc_iql.hip through the C-ABI, host mirror deeprl_signal_control_amd/iql.py)
against the float64 oracle of the reference's IQL agents (oracle/iql_oracle.py: agents/models.py:264-376,
agents/policies.py:285-389, agents/utils.py:231-263).

Tolerances (fp32 kernels vs float64 oracle): Q values |d| <= 2e-5; gradients |d| <= 2e-5 * max|g| per tensor (hidden
units within 1e-6 of a ReLU kink excepted); replay contents and minibatch indices exact; after an Adam step the
parameter change agrees to 2e-6 wherever the gradient is well above that tolerance (Adam's first steps move every
weight by ~lr * sign(g), so an entry whose gradient is rounding noise may legitimately move the other way)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _make(scenario, agent, model_type, E, seed=0, **cfg):
    from deeprl_signal_control_amd.iql import VecIQL
    from deeprl_signal_control_amd.scenario import build_scenario
    from oracle.iql_oracle import OracleIQL
    scn = build_scenario(scenario, agent)
    mc = dict(batch_size=20, buffer_size=1000, reward_norm=3000.0 if scenario == 'large_grid' else 1.0 if scenario == 'real_net' else 100.0)
    mc.update(cfg)
    m = VecIQL(scn.n_s_ls, scn.n_a_ls, scn.n_w_ls, E, scn.s_max, int(scn.green_tab.shape[1]), mc, total_step=10000,
               seed=seed, model_type=model_type)
    o = OracleIQL(m.get_agent_params(), m.n_wave_ls, m.n_w_ls, m.n_a_ls, E, batch_size=m.n_step,
                  buffer_size=int(m.cfg['buffer_size']), gamma=m.cfg['gamma'], reward_norm=m.cfg['reward_norm'],
                  reward_clip=m.cfg['reward_clip'], max_grad_norm=m.cfg['max_grad_norm'], replay_seed=m.replay_seed)
    return scn, m, o


def _rand_obs(scn, E, rng):
    obs = np.zeros((E, scn.n_agent, scn.s_max), np.float32)
    for a, n in enumerate(scn.n_s_ls):
        obs[:, a, :n] = rng.rand(E, n).astype(np.float32) * 2
    return obs


@pytest.mark.parametrize('scenario,agent,model_type,E', [('large_grid', 'iqld', 'dqn', 70), ('large_grid', 'iqll', 'lr', 33),
                                                         ('real_net', 'iqld', 'dqn', 40), ('real_net', 'iqll', 'lr', 5)])
def test_forward_and_epsilon_greedy(scenario, agent, model_type, E):
    from oracle.iql_oracle import act_epsilon_greedy
    from oracle.nets_oracle import sample_uniform
    scn, m, o = _make(scenario, agent, model_type, E, seed=3)
    assert scn.n_f_ls == [0] * scn.n_agent and (scenario == 'real_net') == (max(scn.n_w_ls) == 0)
    A = scn.n_agent
    rng = np.random.RandomState(E)
    for t in range(3):
        obs = _rand_obs(scn, E, rng)
        act, q = m.forward(torch.from_numpy(obs).cuda())                       # mode 'act': argmax
        act, q = act.cpu().numpy(), q.cpu().numpy()
        oq = o.forward(obs)
        for a in range(A):
            np.testing.assert_allclose(q[:, a, :scn.n_a_ls[a]], oq[a], atol=2e-5)
            assert np.all(q[:, a, scn.n_a_ls[a]:] == 0)
            np.testing.assert_array_equal(act[:, a], np.argmax(q[:, a, :scn.n_a_ls[a]], 1))
        eps_before = m.eps_scheduler.n
        act, q = m.forward(torch.from_numpy(obs).cuda(), mode='explore')
        eps = max(m.cfg['epsilon_min'], m.cfg['epsilon_init'] * (1 - (eps_before + 1) / (10000 * m.cfg['epsilon_ratio'])))
        act, q = act.cpu().numpy(), q.cpu().numpy()
        for e in range(E):
            for a in range(A):
                u0 = sample_uniform(m.sample_seed, m.act_step - 1, 2 * (e * A + a))
                u1 = sample_uniform(m.sample_seed, m.act_step - 1, 2 * (e * A + a) + 1)
                assert act[e, a] == act_epsilon_greedy(q[e, a, :scn.n_a_ls[a]], eps, u0, u1)
    m.close()


def _kinks(o, obs_rows, a, thr=1e-6):
    """(layer-1 columns, any layer-2 unit) of agent a whose pre-activation is within thr of zero on the minibatch."""
    from oracle.iql_oracle import DT
    p = o.qs[a].p
    if 'fcw_w' not in p:
        return None, False
    S = torch.as_tensor(np.asarray(obs_rows), dtype=DT)
    nw = o.nw[a]
    z = [S[:, :nw] @ p['fcw_w'] + p['fcw_b']]
    if o.nt[a]:
        z.append(S[:, nw:] @ p['fct_w'] + p['fct_b'])
    z1 = torch.cat(z, 1)
    z2 = torch.relu(z1) @ p['fc0_w'] + p['fc0_b']
    return (z1.abs() < thr).any(0).numpy(), bool((z2.abs() < thr).any())


def _rand_obs_off_the_kinks(scn, E, rng, o, thr=5e-6):
    """_rand_obs, with every row re-drawn until no hidden unit of its agent's net (the oracle's CURRENT parameters) has a
    pre-activation within thr of zero: on such rows the float64 gradient is the gradient, not one of two subgradients, and a
    20 480-row minibatch (which meets ~ 4 such units per agent otherwise) can be compared tensor by tensor without exceptions."""
    from oracle.iql_oracle import DT
    obs = _rand_obs(scn, E, rng)
    for a, n in enumerate(scn.n_s_ls):
        p, nw = o.qs[a].p, o.nw[a]
        if 'fcw_w' not in p:
            continue
        rows = np.arange(E)
        while rows.size:
            S = torch.as_tensor(obs[rows, a, :n].astype(np.float64), dtype=DT)
            z = [S[:, :nw] @ p['fcw_w'] + p['fcw_b']]
            if o.nt[a]:
                z.append(S[:, nw:] @ p['fct_w'] + p['fct_b'])
            z1 = torch.cat(z, 1)
            z2 = torch.relu(z1) @ p['fc0_w'] + p['fc0_b']
            bad = ((z1.abs() < thr).any(1) | (z2.abs() < thr).any(1)).numpy()
            rows = rows[bad]
            if rows.size:
                obs[rows, a, :n] = rng.rand(rows.size, n).astype(np.float32) * 2
    return obs


# fused: '1' = the one-kernel DeepQPolicy learner (csrc/tsc_iql_fused.h, what bench.py --config q1 times), '0' = the grouped-GEMM
# path (TSC_IQL_FUSED=0; IQL-LR always takes it).  E = 1024 x batch 20 = the 20 480 rows per agent of the benchmarked batch,
# distinct transitions in every instance; E = 70 leaves the last 64-row chunk ragged (1400 = 21 x 64 + 56),
# E = 3 is a single partial chunk.
@pytest.mark.parametrize('scenario,agent,model_type,E,cap,fused', [
    ('large_grid', 'iqld', 'dqn', 6, 30, '1'), ('large_grid', 'iqld', 'dqn', 6, 30, '0'), ('large_grid', 'iqll', 'lr', 9, 1000, '0'),
    ('real_net', 'iqld', 'dqn', 4, 64, '1'), ('real_net', 'iqld', 'dqn', 4, 64, '0'), ('large_grid', 'iqld', 'dqn', 3, 25, '1'),
    ('large_grid', 'iqld', 'dqn', 70, 22, '1'), ('large_grid', 'iqld', 'dqn', 1024, 22, '1'), ('large_grid', 'iqld', 'dqn', 1024, 22, '0'),
    ('real_net', 'iqld', 'dqn', 512, 22, '1'), ('large_grid', 'iqll', 'lr', 1024, 22, '0'),         # IQL-LR at the benchmarked batch too
    ('small_grid', 'iqld', 'dqn', 7, 40, '1')])              # s_max 12, 2 - 3 actions, wait inputs in the first 16-feature group
def test_replay_minibatch_gradient_and_adam(scenario, agent, model_type, E, cap, fused, monkeypatch):
    """Fill the rings past their capacity, then three minibatch steps: replay indices (Floyd on the documented
    uniform) exact, TD loss / gradient / clip norm / Adam-updated parameters against the oracle."""
    from deeprl_signal_control_amd import _lib
    monkeypatch.setenv('TSC_IQL_FUSED', fused)
    scn, m, o = _make(scenario, agent, model_type, E, seed=5, buffer_size=cap)
    assert m.fused == (fused == '1')
    A, B = scn.n_agent, m.n_step
    rng = np.random.RandomState(cap + E)
    n_add = cap + (7 if E < 100 else 3) if cap < 100 else 45
    # big batches: observations off the ReLU kinks of the initial nets, so that the first step is compared without exceptions
    draw = (lambda: _rand_obs_off_the_kinks(scn, E, rng, o)) if E >= 100 else (lambda: _rand_obs(scn, E, rng))
    obs = draw()
    assert m.backward() is None                                              # fewer than a batch: no update (models.py:321-322)
    for t in range(n_add):
        nobs = draw()
        act = np.stack([rng.randint(0, n, E) for n in scn.n_a_ls], 1).astype(np.int32)
        rew = -rng.rand(E, A) * 3.0 * m.cfg['reward_norm']
        done = (rng.rand(E) < 0.1).astype(np.uint8)
        m.add_transition(torch.from_numpy(obs).cuda(), torch.from_numpy(act).cuda(), torch.from_numpy(rew).cuda(),
                         torch.from_numpy(nobs).cuda(), torch.from_numpy(done).cuda())
        o.add_transition(obs, act, rew, nobs, done)
        obs = nobs
    assert m.replay_size() == (min(cap, n_add), n_add)
    lr = 1e-3
    kinked = set()          # agents with a hidden unit on a ReLU kink in some minibatch so far (enters Adam's moments)
    for step in range(3 if E < 100 else 2):
        before = m.get_flat().reshape(A, -1).copy()
        _lib.check(m._L.tsc_iql_compute_grads(m._h, m.replay_seed, m.update_step))
        m.update_step += 1
        idx = np.zeros((E, A, B), np.int32)
        _lib.check(m._L.tsc_iql_debug_batch(m._h, idx.ctypes.data_as(C.c_void_p)))
        g = m.layout.unpack(m.grad_tensor().cpu().numpy())
        flat_g = m.grad_tensor().cpu().numpy().reshape(A, -1).copy()
        stats = np.zeros((A, 2))
        _lib.check(m._L.tsc_iql_apply_grads(m._h, lr, 1.0, stats.ctypes.data_as(C.c_void_p)))
        obefore = m.layout.pack(o.agent_params()).reshape(A, -1)
        rows_before = [[o.rings[e][a].buffer for e in range(E)] for a in range(A)]
        params_before = [{k: v.clone() for k, v in q.p.items()} for q in o.qs]
        losses, norms, og = o.minibatch_step(lr)
        np.testing.assert_array_equal(idx, o.last_idx)
        assert all(len(set(idx[e, a])) == B and idx[e, a].max() < min(cap, n_add) for e in range(E) for a in range(A))
        tol = 2e-5
        for a in range(A):
            # ReLU kinks of this minibatch under the parameters the gradient was taken at
            saved = o.qs[a].p
            o.qs[a].p = params_before[a]
            rows = [rows_before[a][e][s][0] for e in range(E) for s in idx[e, a]]
            cols, deep = _kinks(o, rows, a)
            o.qs[a].p = saved
            if E >= 100 and step == 0:
                assert not deep and (cols is None or not cols.any()), 'agent %d: a row of the first minibatch sits on a ReLU kink' % a
            if deep or (cols is not None and cols.any()):
                kinked.add(a)
            for k, ref in og[a].items():
                if deep and k not in ('q_w', 'q_b'):
                    continue
                got, scale = g[a][k], max(np.abs(ref).max(), 1e-9)
                err = np.abs(got - ref)
                if cols is not None and k.startswith(('fcw', 'fct')) and cols.any():
                    sel = cols[:m.layout.n_fc0] if k.startswith('fcw') else cols[m.layout.n_fc0:]
                    err = err[..., ~sel] if err.ndim == 2 else err[~sel]
                assert err.size == 0 or err.max() <= tol * scale, 'step %d agent %d %s: %.2e' % (step, a, k, err.max() / scale)
        np.testing.assert_allclose(stats[:, 0], losses, rtol=1e-4, atol=1e-9)
        np.testing.assert_allclose(stats[:, 1], norms, rtol=1e-4)
        # Adam: this step's parameter change.  Adam's first steps move a weight by ~lr * sign-like(m / sqrt(v)): entries
        # whose gradient is well above the gradient tolerance must move alike; entries whose gradient is itself
        # rounding noise may move either way, by at most ~lr
        after = m.get_flat().reshape(A, -1)
        oflat = m.layout.pack(o.agent_params()).reshape(A, -1)
        d_hip, d_orc = after - before, oflat - obefore
        real = np.abs(flat_g) > 1e-2 * np.abs(flat_g).max(1, keepdims=True)
        real[sorted(kinked)] = False
        assert step > 0 or real.any()
        if real.any():
            assert np.abs(d_hip - d_orc)[real].max() <= 2e-6, np.abs(d_hip - d_orc)[real].max()
        assert np.abs(d_hip).max() <= 1.01 * lr and np.abs(d_hip - d_orc).max() <= 2.02 * lr
        if step == 0:
            assert np.array_equal(after == before, flat_g == 0)                  # structural zeros never move
        # every step is checked from the SAME state: hand the oracle the device's parameters and Adam moments (entries
        # whose gradient is rounding noise legitimately step the other way, and would otherwise compound)
        hm, hv, ht = m.get_opt_state()
        assert ht == step + 1 == o.qs[0].t
        for a, (pp, mm, vv) in enumerate(zip(m.get_agent_params(), m.layout.unpack(hm), m.layout.unpack(hv))):
            for k in o.qs[a].p:
                o.qs[a].p[k] = torch.as_tensor(pp[k].astype(np.float64))
                o.qs[a].m[k] = torch.as_tensor(mm[k].astype(np.float64))
                o.qs[a].v[k] = torch.as_tensor(vv[k].astype(np.float64))
        kinked.clear()
    m.close()


def test_iql_trainer_and_reference_loop_at_E1():
    """VecTrainer drives the IQL agents (utils.py:142-193 value-based branch, :284-295): epsilon decays per control
    step, nothing is learned until the rings hold a batch, then 10 Adam steps per rollout; the E = 1 adaptor driven by
    the reference's own loop lands on the same parameters bit for bit (same kernels, same order)."""
    from deeprl_signal_control_amd.env import TrafficEnv, VecTrafficEnv
    from deeprl_signal_control_amd.iql import IQL, VecIQL
    from deeprl_signal_control_amd.scenario import build_large_grid
    from deeprl_signal_control_amd.trainer import VecTrainer
    scn = build_large_grid('iqld', episode_length_sec=300)                  # 60 control steps = 3 rollouts of 20
    cfg = dict(batch_size=20, buffer_size=1000, reward_norm=3000.0)
    venv = VecTrafficEnv(scn, 1, seed=12, seed_stride=1)
    vmod = VecIQL(scn.n_s_ls, scn.n_a_ls, scn.n_w_ls, 1, scn.s_max, 5, cfg, total_step=1000, seed=0, model_type='dqn')
    p0 = vmod.get_flat().copy()
    tr = VecTrainer(venv, vmod)
    for _ in range(6):
        tr.run_iteration()
    torch.cuda.synchronize()
    assert vmod.eps_scheduler.n == 120 and vmod.update_step == 60 and vmod.get_opt_state()[2] == 60
    assert np.abs(vmod.get_flat() - p0).max() > 0
    # the reference's loop (utils.py:255-308, value-based branch) on the E = 1 adaptor
    env = TrafficEnv(scn, seed=12)
    model = IQL(scn.n_s_ls, scn.n_a_ls, scn.n_w_ls, 1000, cfg, seed=0, model_type='dqn')
    for _ in range(2):
        ob = env.reset()
        model.reset()
        while True:
            for _ in range(model.n_step):
                action, policy = model.forward(ob, mode='explore')
                next_ob, reward, done, global_reward = env.step(action)
                model.add_transition(ob, action, reward, next_ob, done)
                if done:
                    break
                ob = next_ob
            model.backward(None, 0)
            if done:
                env.terminate()
                break
    np.testing.assert_array_equal(model.vec.get_flat(), vmod.get_flat())
    # evaluation path: greedy and stochastic action selection
    venv.train_mode = False
    mean, std = tr.perform(0, 'default')
    assert mean.shape == (1,) and mean[0] < 0
    for x in (venv, env):
        x.close()
    vmod.close(); model.vec.close()


def test_every_instance_draws_its_own_minibatch():
    """Identical rings in every instance: the replay uniforms are keyed by (instance, agent), so the instances still
    draw different minibatches (independent samples, as E independent reference learners would)."""
    from deeprl_signal_control_amd import _lib
    E = 4
    scn, full, _ = _make('large_grid', 'iqld', 'dqn', E, seed=2, buffer_size=40)
    rng = np.random.RandomState(0)
    A = scn.n_agent
    obs = _rand_obs(scn, 1, rng)
    for t in range(30):
        nobs = _rand_obs(scn, 1, rng)
        act = rng.randint(0, 5, (1, A)).astype(np.int32)
        rew = -rng.rand(1, A) * 6000.0
        full.add_transition(torch.from_numpy(np.repeat(obs, E, 0)).cuda(), torch.from_numpy(np.repeat(act, E, 0)).cuda(),
                            torch.from_numpy(np.repeat(rew, E, 0)).cuda(), torch.from_numpy(np.repeat(nobs, E, 0)).cuda(),
                            torch.zeros(E, dtype=torch.uint8, device='cuda'))
        obs = nobs
    _lib.check(full._L.tsc_iql_compute_grads(full._h, 77, 0))
    idx = np.zeros((E, A, 20), np.int32)
    _lib.check(full._L.tsc_iql_debug_batch(full._h, idx.ctypes.data_as(C.c_void_p)))
    assert len({tuple(idx[e, 0]) for e in range(E)}) == E
    gf = full.grad_tensor().cpu().numpy()
    assert np.isfinite(gf).all() and np.abs(gf).max() > 0
    full.close()


def test_other_batch_sizes_take_the_generic_draw_and_the_same_kernels():
    """batch_size != 20 (the reference's value, which has its own register-resident Floyd draw): the generic sampler and the fused
    learner at 8 rows per instance -- indices exact, gradients against the oracle."""
    from deeprl_signal_control_amd import _lib
    scn, m, o = _make('large_grid', 'iqld', 'dqn', 9, seed=4, buffer_size=16, batch_size=8)
    assert m.fused and m.n_step == 8
    A, B, E = scn.n_agent, 8, 9
    rng = np.random.RandomState(1)
    obs = _rand_obs(scn, E, rng)
    for t in range(19):
        nobs = _rand_obs(scn, E, rng)
        act = np.stack([rng.randint(0, n, E) for n in scn.n_a_ls], 1).astype(np.int32)
        rew = -rng.rand(E, A) * 3.0 * m.cfg['reward_norm']
        done = (rng.rand(E) < 0.1).astype(np.uint8)
        m.add_transition(torch.from_numpy(obs).cuda(), torch.from_numpy(act).cuda(), torch.from_numpy(rew).cuda(),
                         torch.from_numpy(nobs).cuda(), torch.from_numpy(done).cuda())
        o.add_transition(obs, act, rew, nobs, done)
        obs = nobs
    _lib.check(m._L.tsc_iql_compute_grads(m._h, m.replay_seed, m.update_step))
    idx = np.zeros((E, A, B), np.int32)
    _lib.check(m._L.tsc_iql_debug_batch(m._h, idx.ctypes.data_as(C.c_void_p)))
    g = m.layout.unpack(m.grad_tensor().cpu().numpy())
    losses, norms, og = o.minibatch_step(1e-3)
    np.testing.assert_array_equal(idx, o.last_idx)
    assert all(len(set(idx[e, a])) == B and idx[e, a].max() < 16 for e in range(E) for a in range(A))
    for a in range(A):
        for k, ref in og[a].items():
            scale = max(np.abs(ref).max(), 1e-9)
            # 72 rows: a unit within 1e-6 of its ReLU kink is a ~ 1e-4 event per tensor; such a tensor would fail loudly, not silently
            assert np.abs(g[a][k] - ref).max() <= 5e-5 * scale, (a, k)
    m.close()
