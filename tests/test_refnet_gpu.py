"""Oracle-B on the GPU: the HIP learner (csrc/tsc_model.hip through the C-ABI, host mirror agents.VecA2C) replays the
training episodes recorded from the REFERENCE'S OWN learner code (agents/models.py, agents/policies.py,
agents/utils.py and Trainer.run of utils.py executed unmodified over oracle/fake_tf.py; tools/make_golden.py refnet ->
tests/golden/refnet_*.npz) -- MA2C / IA2C / FcACPolicy on large_grid, MA2C on Monaco (2..6 actions).

Every recorded quantity is compared: the initial weights (ortho_init under the recorded np.random seed), pi and v of
every forward call (including the 'v'-only bootstrap call that must not advance the LSTM state), the float32 returns
and advantages, the raw gradient of every variable of every agent, the per-agent global norm (one fixture clips), the
loss, and the variables and RMSProp accumulators after each of the consecutive updates.

Two ways in: the reference's API order (forward, add_transition, backward -> the update re-evaluates the forward graph
at training shape) and the zero-copy rollout slots with the fused forward's activation cache (what VecTrainer and the
benchmark run).  E > 1 feeds the recorded episode to every instance: the loss is a mean over the E * T samples of an
agent, so gradients, norms and updates must equal the reference's E = 1 values -- this is the parity test at the
benchmarked batch (E = 1024, T = 120: weight-stationary multi-tile forward, 5 row splits of 24 576 rows in the update).

Tolerances (float32 kernels against float64 recordings): forward 2e-5 absolute; returns 1e-6 (float32 v feeds a float64
recursion on both sides); gradients 1e-4 of the tensor's largest entry (1e-3 at E = 1024, measured 3.7e-4: E identical copies of every
addend are the worst case for a float32 running sum -- the rounding of equal addends does not average out; the same
shape on E DISTINCT instances is checked against the float64 restatement in tests/test_model_gpu.py); parameters
after an update 3e-5 absolute (lr 5e-4 times a unit-scale RMSProp step)."""
import numpy as np
import pytest
import torch

from oracle import refnet

pytestmark = pytest.mark.gpu


def _make(fx, E):
    from deeprl_signal_control_amd.agents import VecA2C
    cfg = refnet.fixture_model_cfg(fx)
    n_s, n_a, n_w, n_f = (fx[k].tolist() for k in ('n_s_ls', 'n_a_ls', 'n_w_ls', 'n_f_ls'))
    s_max = (max(n_s) + 3) // 4 * 4
    m = VecA2C(n_s, n_a, n_w, n_f, E, s_max, max(n_a), cfg, total_step=len(fx['actions']), device=0, seed=int(fx['seed_w']),
               name=str(fx['agent']), policy=str(fx['policy']))      # total_step as main.py passes it (the schedules' horizon)
    return m, s_max


def _obs(fx, i, E, s_max):
    o = np.zeros((E, fx['fw_obs'].shape[1], s_max), np.float32)
    o[:, :, :fx['fw_obs'].shape[2]] = fx['fw_obs'][i][None]
    return torch.from_numpy(o).cuda()


CASES = [('refnet_ma2c_large', 1, 'api'), ('refnet_ma2c_large', 96, 'slots'), ('refnet_ma2c_large', 1024, 'slots'),
         ('refnet_ia2c_large', 1, 'slots'), ('refnet_ia2c_large', 40, 'api'),
         ('refnet_fc_large', 1, 'api'), ('refnet_fc_large', 33, 'api'), ('refnet_fc_large', 256, 'api'),     # 256 x 120 = BASELINE configs[1]'s batch
         ('refnet_ma2c_real', 1, 'api'), ('refnet_ma2c_real', 64, 'slots'),
         ('refnet_ma2c_real', 512, 'slots')]                      # 512 x 40 = the batch bench.py --config c5 times (BASELINE configs[4] per GPU)


@pytest.mark.parametrize('name,E,path', CASES)
def test_hip_replays_reference_learner(name, E, path):
    fx = refnet.load_fixture(name)
    m, s_max = _make(fx, E)
    n_a = fx['n_a_ls'].tolist()
    A, n_step = len(n_a), int(fx['n_step'])
    # a17 through the product path: VecA2C(seed=s) holds the reference's weights under np.random.seed(s)
    refnet.check_digests(m.get_tower_params(), fx['w0/names'], fx['w0/rows'], 0.0, 'w0')
    dev = m.device
    sl = m.rollout_slots() if path == 'slots' else None
    m.reset()
    worst = dict(pi=0.0, v=0.0, g=0.0, w=0.0)
    # replicated data: every instance adds the same fp32 term, so the rounding of the batch sum is biased and grows with E
    # (distinct data at E=1024: 3e-6, test_model_gpu.py::test_update_benchmarked_batch_E1024_T120)
    gtol = 1e-4 if E <= 8 else 3e-4 if E <= 128 else 1e-3

    def backward(k, R):
        p = 'bw%d/' % k
        np.testing.assert_allclose(R.cpu().numpy(), np.broadcast_to(fx[p + 'R'], (E, A)), rtol=0, atol=2e-5)
        m.compute_grads(R)
        assert abs(m._cur_beta - float(fx[p + 'beta'])) < 1e-12 and abs(m._cur_lr - float(fx[p + 'lr'])) < 1e-12      # schedules (a16)
        Rs, Advs = np.zeros((n_step, E, A), np.float32), np.zeros((n_step, E, A), np.float32)
        from deeprl_signal_control_amd import _lib
        import ctypes as C
        _lib.check(m._L.tsc_model_get_returns(m._h, Rs.ctypes.data_as(C.c_void_p), Advs.ctypes.data_as(C.c_void_p)))
        np.testing.assert_allclose(Rs, np.broadcast_to(fx[p + 'Rs'][:, None, :], Rs.shape), rtol=0, atol=1e-6)
        np.testing.assert_allclose(Advs, np.broadcast_to(fx[p + 'Advs'][:, None, :], Advs.shape), rtol=0, atol=3e-5)
        g = m.unpack(m.grad_tensor().cpu().numpy())
        worst['g'] = max(worst['g'], refnet.check_digests(g, fx[p + 'g/names'], fx[p + 'g/rows'], gtol, p + 'g', sum_tol=2 * gtol))
        full = {k_[len(p) + 6:]: fx[k_] for k_ in fx if k_.startswith(p + 'gfull/')}
        for key, want in full.items():                                   # one agent's complete gradient (layout coverage)
            t, kk = key.split('/')
            np.testing.assert_allclose(g[int(t)][kk], want, rtol=0, atol=gtol * np.abs(want).max(), err_msg=key)
        stats = m.apply_grads(1.0, want_stats=True)
        np.testing.assert_allclose(stats[:, :3].sum(1), fx[p + 'loss'], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(stats[:, 3], fx[p + 'norm'], rtol=2e-4)
        want = refnet.unpack_digests(fx[p + 'w/names'], fx[p + 'w/rows'])
        got = refnet.tower_digest(m.get_tower_params())
        assert set(got) == set(want)
        for k_ in want:
            d = np.abs(got[k_][refnet.N_SUMS:] - want[k_][refnet.N_SUMS:]).max()
            worst['w'] = max(worst['w'], d)
            assert d <= 3e-5, (p, k_, d)
        ms = refnet.tower_digest(m.unpack(m.get_flat('ms')), sums_only=True)
        wms = refnet.unpack_digests(fx[p + 'ms/names'], fx[p + 'ms/rows'])
        for k_ in wms:
            np.testing.assert_allclose(ms[k_], wms[k_], rtol=2e-5, err_msg=p + 'ms ' + k_)

    t = bw = 0
    for i, typ in enumerate(fx['fw_type']):
        typ = str(typ)
        obs = _obs(fx, i, E, s_max)
        done = torch.full((E,), int(fx['fw_done'][i]), dtype=torch.uint8, device=dev)
        if typ == 'pv':
            ts = t % n_step
            if sl is not None:
                sl['obs'][ts].copy_(obs); sl['done'][ts].copy_(done)
                pi, v, _ = m.forward_sample(sl['obs'][ts], sl['done'][ts], v_out=sl['value'][ts], action_out=sl['action'][ts])
            else:
                pi, v = m.forward(obs, done, 'pv')
            pi_h, v_h = pi.cpu().numpy(), v.cpu().numpy()
            for a in range(A):
                worst['pi'] = max(worst['pi'], np.abs(pi_h[:, a, :n_a[a]] - fx['fw_pi'][i, a, :n_a[a]]).max())
                assert (pi_h[:, a, n_a[a]:] == 0).all()
            worst['v'] = max(worst['v'], np.abs(v_h - fx['fw_v'][i]).max())
            act = torch.from_numpy(np.broadcast_to(fx['actions'][t], (E, A)).astype(np.int32).copy()).to(dev)
            rew = torch.from_numpy(np.broadcast_to(fx['reward'][t], (E, A)).astype(np.float64).copy()).to(dev)
            dpost = torch.full((E,), int(fx['done'][t]), dtype=torch.uint8, device=dev)
            if sl is not None:
                sl['action'][ts].copy_(act); sl['reward'][ts].copy_(rew); sl['done'][ts + 1].copy_(dpost)
                m.commit_transition()
            else:
                m.add_transition(obs, done, act, rew, v, dpost)
            t += 1
            if t % n_step == 0 and fx['done'][t - 1]:
                backward(bw, torch.zeros(E, A, dtype=torch.float32, device=dev))
                bw += 1
        else:
            v = m.forward(obs, False, 'v')
            worst['v'] = max(worst['v'], np.abs(v.cpu().numpy() - fx['fw_v'][i]).max())
            backward(bw, v)
            bw += 1
    assert bw == int(fx['n_backward']) and t == len(fx['actions'])
    assert worst['pi'] <= 2e-5 and worst['v'] <= 2e-5, worst
    print('refnet %s E=%d %s: max |d pi| %.2e, |d v| %.2e, grad %.2e of max, weights %.2e' %
          (name, E, path, worst['pi'], worst['v'], worst['g'], worst['w']))
    m.close()


# ---- IQL-LR / IQL-DNN -------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('name,E', [('refnet_iqll_large', 1), ('refnet_iqld_large', 1), ('refnet_iqld_large', 8), ('refnet_iqld_large', 1024)])
def test_hip_replays_reference_iql(name, E):
    """csrc/tsc_iql.hip (VecIQL) against the reference IQL executed over oracle/fake_tf.py (agents/models.py:264-376,
    agents/policies.py:285-389, agents/utils.py:231-263 unmodified): weights under the seed, Q values of every forward,
    the epsilon schedule, ring contents through the reference's own minibatch draws (tsc_iql_compute_grads_at), loss and
    global norm of each of the 3 x 10 Adam steps, gradients of step 0 and 9, weights and Adam moments after every backward.
    E = 8 replicates the episode in every instance (the loss is a mean over E * batch rows)."""
    from deeprl_signal_control_amd.iql import VecIQL
    fx = refnet.load_fixture(name)
    n_s, n_w, n_a = (fx[k].tolist() for k in ('n_s_ls', 'n_w_ls', 'n_a_ls'))
    A, B, T = len(n_a), int(fx['n_step']), len(fx['actions'])
    s_max = (max(n_s) + 3) // 4 * 4
    kind = 'dqn' if str(fx['agent']) == 'iqld' else 'lr'
    m = VecIQL(n_s, n_a, n_w, E, s_max, max(n_a), dict(batch_size=B), total_step=T, device=0, seed=int(fx['seed_w']), model_type=kind)
    N = refnet.N_SUMS

    def agent_digest(agents, sums_only=False):
        return {'%d/%s' % (a, k): refnet.digest(v)[:N if sums_only else None] for a, p in enumerate(agents) for k, v in p.items()}

    def check(got, names, rows, tol, what, absolute=None):
        want = refnet.unpack_digests(names, rows)
        assert set(got) == set(want), what
        worst = 0.0
        for k in want:
            if absolute is None:
                np.testing.assert_allclose(got[k][:N], want[k][:N], rtol=2 * tol, atol=2 * tol * max(want[k][1], 1e-30), err_msg='%s %s' % (what, k))
            if len(want[k]) > N:
                d = np.abs(got[k][N:] - want[k][N:]).max()
                d = d if absolute is not None else d / max(want[k][3], 1e-30)
                worst = max(worst, d)
                assert d <= (absolute if absolute is not None else tol), (what, k, d)
        return worst
    check(agent_digest(m.get_agent_params()), fx['w0/names'], fx['w0/rows'], 0.0, 'w0')
    dev = m.device

    def dev_obs(x):
        o = np.zeros((E, A, s_max), np.float32)
        o[:, :, :x.shape[1]] = x[None]
        return torch.from_numpy(o).to(dev)
    worst = dict(q=0.0, g=0.0, w=0.0)
    bw = 0
    for t in range(T):
        obs, nxt = dev_obs(fx['fw_obs'][t]), dev_obs(fx['next_obs'][t])
        _, q = m.forward(obs, mode='explore')
        assert abs(m.last_eps - fx['fw_eps'][t]) < 1e-12
        qh = q.cpu().numpy()
        for a in range(A):
            worst['q'] = max(worst['q'], np.abs(qh[:, a, :n_a[a]] - fx['fw_q'][t, a, :n_a[a]]).max())
        act = torch.from_numpy(np.broadcast_to(fx['actions'][t], (E, A)).astype(np.int32).copy()).to(dev)
        rew = torch.from_numpy(np.broadcast_to(fx['reward'][t], (E, A)).astype(np.float64).copy()).to(dev)
        m.add_transition(obs, act, rew, nxt, torch.full((E,), int(fx['done'][t]), dtype=torch.uint8, device=dev))
        if (t + 1) % B == 0:
            p = 'bw%d/' % bw
            lr = m.lr_scheduler.get(m.n_step)
            assert abs(lr - float(fx[p + 'lr'])) < 1e-15
            for k in range(10):
                idx = torch.from_numpy(np.broadcast_to(fx[p + 'idx'][k], (E, A, B)).astype(np.int32).copy()).to(dev)
                if k in (0, 9):
                    from deeprl_signal_control_amd import _lib
                    import ctypes as C
                    _lib.check(m._L.tsc_iql_compute_grads_at(m._h, C.c_void_p(idx.data_ptr())))
                    g = m.layout.unpack(m.grad_tensor().cpu().numpy())
                    worst['g'] = max(worst['g'], check(agent_digest(g), fx[p + 'g%d/names' % k], fx[p + 'g%d/rows' % k], 1e-4, p + 'g%d' % k))
                stats = m.minibatch_step_at(idx, lr, want_stats=True)
                np.testing.assert_allclose(stats[:, 0], fx[p + 'loss'][k], rtol=2e-4, atol=1e-12)
                np.testing.assert_allclose(stats[:, 1], fx[p + 'norm'][k], rtol=2e-4, atol=1e-9)
            worst['w'] = max(worst['w'], check(agent_digest(m.get_agent_params()), fx[p + 'w/names'], fx[p + 'w/rows'], 0, p + 'w', absolute=3e-6))
            mm, vv, tt = m.get_opt_state()
            assert tt == 10 * (bw + 1)
            check(agent_digest(m.layout.unpack(mm), True), fx[p + 'm/names'], fx[p + 'm/rows'], 2e-4, p + 'm')
            check(agent_digest(m.layout.unpack(vv), True), fx[p + 'v/names'], fx[p + 'v/rows'], 2e-4, p + 'v')
            bw += 1
    assert bw == 3 and worst['q'] <= 2e-5, worst
    print('refnet %s E=%d: max |d q| %.2e, grad %.2e of max, weights %.2e' % (name, E, worst['q'], worst['g'], worst['w']))
    m.close()
