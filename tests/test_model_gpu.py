"""GPU parity of the HIP learner path (csrc/tsc_model.hip, csrc/tsc_gemm.h through the C-ABI)
against the CPU oracle (oracle/nets_oracle.py, float64 restatement of agents/policies.py +
agents/utils.py) and against known answers recorded from the reference's OnPolicyBuffer.

Tolerances (floating point, fp32 kernels vs float64 oracle): forward outputs |d| <= 2e-5;
gradients |d| <= 2e-5 * max|g| per tensor (1e-4 at n_step = 120; fp32 accumulation over T*E samples through a
T-step BPTT -- measured ~5e-7), except the weight-gradient columns of hidden units that sit on a ReLU kink in the
oracle (|pre-activation| < 1e-5 for some sample: float32 may evaluate the other side); returns/advantages
bit-exact (float64 recursion on both sides)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from deeprl_signal_control_amd.scenario import build_large_grid

pytestmark = pytest.mark.gpu


def _vp(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _gemm(form, epi, A, B, C_, M, N, K, lda, ldb, ldc, sA, sB, sC, groups, bias=None, aux=None, rr=None, colsum=None,
          ws=None, wsc=None):
    from deeprl_signal_control_amd import _lib
    from deeprl_signal_control_amd.agents import _setup_lib
    L = _lib.lib()
    _setup_lib(L)
    _lib.check(L.tsc_gemm_grouped_f32(form, epi, groups, M, N, K, _vp(A), sA, lda, _vp(B), sB, ldb, _vp(C_), sC, ldc,
                                      _vp(bias), _vp(aux), _vp(rr), _vp(colsum), _vp(ws), ws.numel() if ws is not None else 0,
                                      _vp(wsc), wsc.numel() if wsc is not None else 0,
                                      C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()


@pytest.mark.parametrize('M,N,K,G', [(300, 224, 52, 3), (128, 256, 224, 2), (1000, 8, 64, 2), (77, 100, 36, 1)])
def test_gemm_nn(M, N, K, G):
    rng = np.random.RandomState(M + N)
    A = rng.randn(G, M, K).astype(np.float32); B = rng.randn(G, K, N).astype(np.float32)
    bias = rng.randn(G, N).astype(np.float32)
    dA, dB, db = (torch.from_numpy(x).cuda() for x in (A, B, bias))
    ref = np.einsum('gmk,gkn->gmn', A.astype(np.float64), B.astype(np.float64))
    tol = 1e-5 * K
    for epi, want in ((0, ref), (1, ref + bias[:, None, :]), (2, np.maximum(ref + bias[:, None, :], 0))):
        out = torch.full((G, M, N), 7.0, device='cuda')
        _gemm(0, epi, dA, dB, out, M, N, K, K, N, N, M * K, K * N, M * N, G, bias=db)
        np.testing.assert_allclose(out.cpu().numpy(), want, atol=tol, rtol=1e-5)
    # relu-backward mask epilogue, in place over aux
    aux = rng.randn(G, M, N).astype(np.float32)
    d_aux = torch.from_numpy(aux).cuda()
    _gemm(0, 3, dA, dB, d_aux, M, N, K, K, N, N, M * K, K * N, M * N, G, aux=d_aux)
    np.testing.assert_allclose(d_aux.cpu().numpy(), ref * (aux > 0), atol=tol, rtol=1e-5)


@pytest.mark.parametrize('Kred,M,N,G,split', [(1000, 52, 224, 2, False), (512, 64, 256, 2, False), (333, 64, 8, 3, False),
                                               (4096, 224, 256, 1, False), (20000, 52, 224, 2, True), (9001, 64, 8, 3, True),
                                               (16384, 224, 256, 2, True)])
def test_gemm_tn_colsum_rowrange(Kred, M, N, G, split):
    rng = np.random.RandomState(Kred)
    A = rng.randn(G, Kred, M).astype(np.float32); B = rng.randn(G, Kred, N).astype(np.float32)
    dA, dB = torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda()
    ref = np.einsum('gkm,gkn->gmn', A.astype(np.float64), B.astype(np.float64))
    out = torch.zeros(G, M, N, device='cuda'); cs = torch.zeros(G, N, device='cuda')
    ws = torch.zeros(8 << 20, device='cuda') if split else None      # split-K workspace (deterministic chunks)
    wsc = torch.zeros(1 << 16, device='cuda') if split else None
    _gemm(1, 0, dA, dB, out, M, N, Kred, M, N, N, Kred * M, Kred * N, M * N, G, colsum=cs, ws=ws, wsc=wsc)
    tol = 2e-6 * Kred
    np.testing.assert_allclose(out.cpu().numpy(), ref, atol=tol, rtol=1e-4)
    np.testing.assert_allclose(cs.cpu().numpy(), B.astype(np.float64).sum(1), atol=tol, rtol=1e-4)
    rr = np.zeros((G, M, 2), np.int16)
    rr[:, :, 0] = rng.randint(0, N // 2, (G, M)); rr[:, :, 1] = rr[:, :, 0] + rng.randint(0, N // 2, (G, M))
    out2 = torch.zeros(G, M, N, device='cuda')
    _gemm(1, 4, dA, dB, out2, M, N, Kred, M, N, N, Kred * M, Kred * N, M * N, G, rr=torch.from_numpy(rr).cuda(), ws=ws, wsc=wsc)
    if split:                                              # deterministic: bit-identical on a second run
        out3 = torch.zeros(G, M, N, device='cuda')
        _gemm(1, 4, dA, dB, out3, M, N, Kred, M, N, N, Kred * M, Kred * N, M * N, G, rr=torch.from_numpy(rr).cuda(), ws=ws, wsc=wsc)
        assert torch.equal(out2, out3)
    n = np.arange(N)[None, None, :]
    mask = (n >= rr[:, :, :1]) & (n < rr[:, :, 1:])
    np.testing.assert_allclose(out2.cpu().numpy(), ref * mask, atol=tol, rtol=1e-4)


def _make(agent, E, n_step, seed=0, policy='lstm', scenario='large_grid', **cfg):
    from deeprl_signal_control_amd.agents import VecA2C
    from deeprl_signal_control_amd.scenario import build_scenario
    from oracle.nets_oracle import OracleA2C
    scn = build_scenario(scenario, agent)
    mc = dict(batch_size=n_step)
    if scenario == 'real_net':
        mc['reward_norm'] = 1.0                                # config/config_{ma2c,ia2c}_real.ini
    mc.update(cfg)
    m = VecA2C(scn.n_s_ls, scn.n_a_ls, scn.n_w_ls, scn.n_f_ls, E, scn.s_max, int(scn.green_tab.shape[1]), mc, device=0,
               seed=seed, name=agent, policy=policy)
    o = OracleA2C(m.get_tower_params(), m.n_wave_ls, m.n_w_ls, m.n_f_ls, m.n_a_ls, E,
                  gamma=m.cfg['gamma'], reward_norm=m.cfg['reward_norm'], reward_clip=m.cfg['reward_clip'],
                  value_coef=m.cfg['value_coef'], max_grad_norm=m.cfg['max_grad_norm'])
    return scn, m, o


def _grad_err(o, t, k, got, og):
    """max |got - og| / max|og| over the entries whose hidden unit is not on a ReLU kink in the oracle
    (OracleA2C.kink_cols: a float32 evaluation may sit on the other side of the kink there; those columns are only
    required to be finite and within 10 % of the tensor's scale)."""
    scale = max(np.abs(og).max(), 1e-7)
    err = np.abs(got - og)
    layer = k.split('_')[0]
    kink = getattr(o, 'kink_cols', None)
    if kink is not None and layer in kink[t]:
        cols = kink[t][layer]
        if cols.any():
            bad = err[..., cols] if err.ndim == 2 else err[cols]
            assert np.isfinite(bad).all() and bad.max() <= 0.1 * scale
            err = err[..., ~cols] if err.ndim == 2 else err[~cols]
            if err.size == 0:
                return 0.0
    return float(err.max() / scale)


def _rand_obs(scn, E, rng):
    obs = np.zeros((E, scn.n_agent, scn.s_max), np.float32)
    for a, n in enumerate(scn.n_s_ls):
        obs[:, a, :n] = rng.rand(E, n).astype(np.float32) * 2
    return obs


@pytest.mark.parametrize('agent,E,policy', [('ma2c', 5, 'lstm'), ('ia2c', 70, 'lstm'), ('ia2c', 33, 'fc')])
def test_forward_matches_oracle(agent, E, policy):
    scn, m, o = _make(agent, E, 4, policy=policy)
    rng = np.random.RandomState(1)
    m.reset(); o.reset()
    for t in range(5):
        obs = _rand_obs(scn, E, rng)
        done = (rng.rand(E) < (1.0 if t == 0 else 0.3)).astype(np.uint8)
        pi, v = m.forward(torch.from_numpy(obs).cuda(), torch.from_numpy(done).cuda(), 'pv')
        pi, v = pi.cpu().numpy(), v.cpu().numpy()
        opi, ov = o.forward(obs, done, 'pv')
        for a in range(scn.n_agent):
            np.testing.assert_allclose(pi[:, a, :scn.n_a_ls[a]], opi[a], atol=2e-5, err_msg='pi t=%d a=%d' % (t, a))
        np.testing.assert_allclose(v, ov, atol=2e-5)
        assert np.allclose(pi.sum(-1), 1.0, atol=1e-5)
    # bootstrap value: state must NOT advance (policies.py:127-135)
    obs = _rand_obs(scn, E, rng)
    vb = m.forward(torch.from_numpy(obs).cuda(), False, 'v').cpu().numpy()
    _, ovb = o.forward(obs, np.zeros(E), 'v')
    np.testing.assert_allclose(vb, ovb, atol=2e-5)
    vb2 = m.forward(torch.from_numpy(obs).cuda(), False, 'v').cpu().numpy()
    np.testing.assert_array_equal(vb, vb2)
    m.close()


@pytest.mark.parametrize('scenario,agent,E,mfma', [('large_grid', 'ia2c', 256, '1'), ('large_grid', 'ia2c', 45, '0'),
                                                   ('real_net', 'ia2c', 70, '1'), ('real_net', 'ia2c', 33, '0')])
def test_fc_policy_forward_kernels(scenario, agent, E, mfma, monkeypatch):
    """FcACPolicy rollout forward (agents/policies.py:214-240; IA2C only: the reference has no working fingerprint variant): the
    MFMA kernel (policy_fwd_fc_mfma_kernel, first layers of 160 / 128 columns = large_grid / Monaco, heterogeneous n_a on
    Monaco, ragged last tile) and the per-thread kernel (TSC_FC_MFMA=0) against the float64 oracle: pi, v, sampled action."""
    from oracle.nets_oracle import choice_from_uniform, sample_uniform
    monkeypatch.setenv('TSC_FC_MFMA', mfma)
    scn, m, o = _make(agent, E, 4, policy='fc', scenario=scenario, seed=3)
    rng = np.random.RandomState(5)
    m.reset(); o.reset()
    flips = 0
    for t in range(3):
        obs = _rand_obs(scn, E, rng)
        step = m.sample_step
        pi, v, act = m.forward_sample(torch.from_numpy(obs).cuda(), False, cache=False)
        pi, v, act = pi.cpu().numpy(), v.cpu().numpy(), act.cpu().numpy()
        opi, ov = o.forward(obs, np.zeros(E), 'pv')
        for a in range(scn.n_agent):
            na = scn.n_a_ls[a]
            np.testing.assert_allclose(pi[:, a, :na], opi[a], atol=2e-5, err_msg='pi t=%d a=%d' % (t, a))
            assert np.all(pi[:, a, na:] == 0)
        np.testing.assert_allclose(v, ov, atol=2e-5)
        for e in range(E):
            for a in range(scn.n_agent):
                u = sample_uniform(m.sample_seed, step, e * scn.n_agent + a)
                flips += int(act[e, a] != choice_from_uniform(pi[e, a, :scn.n_a_ls[a]], u))
    assert flips == 0                      # the action is np.random.choice on the kernel's own pi and the documented uniform
    m.close()


def test_sampling_is_numpy_choice_on_documented_uniform():
    from oracle.nets_oracle import choice_from_uniform, sample_uniform
    scn, m, o = _make('ma2c', 16, 4, seed=11)
    rng = np.random.RandomState(2)
    pi = rng.dirichlet(np.ones(5), size=(16, 25)).astype(np.float32)
    pi[3, 4] = [0, 0, 1, 0, 0]
    for step in range(3):
        act = m.sample(torch.from_numpy(pi).cuda()).cpu().numpy()
        for e in range(16):
            for a in range(25):
                u = sample_uniform(11, step, e * 25 + a)
                assert act[e, a] == choice_from_uniform(pi[e, a], u)
    assert act[3, 4] == 2
    counts = np.zeros(5)
    big = np.tile(np.array([0.1, 0.2, 0.3, 0.25, 0.15], np.float32), (16, 25, 1))
    for step in range(200):
        a = m.sample(torch.from_numpy(big).cuda()).cpu().numpy()
        counts += np.bincount(a.ravel(), minlength=5)
    np.testing.assert_allclose(counts / counts.sum(), big[0, 0], atol=0.01)
    m.close()


def test_fc_policy_forward_kernels_agree_with_each_other(monkeypatch):
    """the two FcACPolicy forward kernels sum the 64 hidden units of a head in different orders (the MFMA
    kernel: 16 lanes x 4 units, xor-shuffle tree; the per-thread kernel: sequentially), so their logits -- and with them pi
    -- agree to float32 rounding, not bit for bit.  Pinned here: |d pi| <= 4e-7 (a few ulp of a probability), v likewise, and
    the sampled action of the same (seed, step, index) uniform is the same wherever that uniform is not within 1e-6 of a
    boundary of the cumulative distribution (and nowhere else may it differ)."""
    from oracle.nets_oracle import sample_uniform
    E = 256
    out = {}
    for mfma in ('1', '0'):
        monkeypatch.setenv('TSC_FC_MFMA', mfma)
        scn, m, _ = _make('ia2c', E, 4, policy='fc', seed=3)
        rng = np.random.RandomState(5)
        m.reset()
        rows = []
        for t in range(3):
            obs = _rand_obs(scn, E, rng)
            step = m.sample_step
            pi, v, act = m.forward_sample(torch.from_numpy(obs).cuda(), False, cache=False)
            rows.append((pi.cpu().numpy().copy(), v.cpu().numpy().copy(), act.cpu().numpy().copy(), step, m.sample_seed))
        out[mfma] = rows
        m.close()
    differ = 0
    for (pa, va, aa, step, seed), (pb, vb, ab, _, _) in zip(out['1'], out['0']):
        assert np.abs(pa - pb).max() <= 4e-7 and np.abs(va - vb).max() <= 2e-6 * max(1.0, np.abs(va).max())
        for e, a in zip(*np.nonzero(aa != ab)):
            u = sample_uniform(seed, step, e * aa.shape[1] + a)
            cdf = np.cumsum(pa[e, a].astype(np.float64))
            assert np.abs(cdf - u).min() < 1e-6, (e, a, u, cdf)
            differ += 1
    assert differ <= 2                      # 3 x 256 x 25 draws: a boundary within 1e-6 of the uniform is a ~1e-5 event per draw
    print('FC kernels: %d of %d sampled actions differ (all on a cdf boundary)' % (differ, 3 * E * 25))


def _fill(scn, m, o, E, T, rng, p_done=0.1, terminal=False, use_cache=False):
    obs = _rand_obs(scn, E, rng)
    done = np.ones(E, np.uint8)
    for t in range(T):
        if use_cache:      # the trainer's path: fused forward + sampling, activations cached for the update
            pi, v, _ = m.forward_sample(torch.from_numpy(obs).cuda(), torch.from_numpy(done).cuda())
        else:
            pi, v = m.forward(torch.from_numpy(obs).cuda(), torch.from_numpy(done).cuda(), 'pv')
        v = v.cpu().numpy()
        o.forward(obs, done, 'pv')
        act = np.stack([rng.randint(0, n, E) for n in scn.n_a_ls], 1).astype(np.int32)
        rew = -rng.rand(E, scn.n_agent) * 3.0 * m.cfg['reward_norm']
        dpost = (rng.rand(E) < p_done).astype(np.uint8)
        if terminal and t == T - 1:
            dpost[:] = 1
        m.add_transition(torch.from_numpy(obs).cuda(), torch.from_numpy(done).cuda(), torch.from_numpy(act).cuda(),
                         torch.from_numpy(rew).cuda(), torch.from_numpy(v).cuda(), torch.from_numpy(dpost).cuda())
        o.add_transition(obs, done, act, rew, v, dpost)
        obs, done = _rand_obs(scn, E, rng), dpost
    return obs, done


@pytest.mark.parametrize('agent,E,T,terminal,policy,use_cache', [
    ('ma2c', 3, 6, False, 'lstm', False), ('ma2c', 66, 8, True, 'lstm', False), ('ia2c', 4, 40, False, 'lstm', False),
    ('ia2c', 37, 12, True, 'fc', False), ('ma2c', 70, 7, True, 'lstm', True), ('ia2c', 5, 30, False, 'lstm', True)])
def test_backward_matches_oracle(agent, E, T, terminal, policy, use_cache):
    scn, m, o = _make(agent, E, T, seed=5, policy=policy)
    rng = np.random.RandomState(E * T)
    m.reset(); o.reset()
    from deeprl_signal_control_amd import _lib
    for it in range(2):                                       # second round exercises states_bw / carried done
        obs, done = _fill(scn, m, o, E, T, rng, terminal=terminal and it == 0, use_cache=use_cache)
        Rb = m.forward(torch.from_numpy(obs).cuda(), False, 'v').clone()
        _, oRb = o.forward(obs, np.zeros(E), 'v')
        Rb_np = Rb.cpu().numpy()
        beta = 0.01
        _lib.check(m._L.tsc_model_compute_grads(m._h, C.c_void_p(Rb.data_ptr()), beta))
        ograds, ostats = o.compute_grads(Rb_np, beta)
        Rs = np.zeros((T, E, scn.n_agent), np.float32); Advs = np.zeros_like(Rs)
        _lib.check(m._L.tsc_model_get_returns(m._h, Rs.ctypes.data_as(C.c_void_p), Advs.ctypes.data_as(C.c_void_p)))
        np.testing.assert_array_equal(Rs, o.Rs)               # float64 recursion on both sides
        np.testing.assert_array_equal(Advs, o.Advs)
        g = m.unpack(m.grad_tensor().cpu().numpy())
        for t in range(m.G):
            for k, og in ograds[t].items():
                err = _grad_err(o, t, k, g[t][k], og.numpy())
                # second round: the parameters already differ by the first update's fp32 rounding (<= 3e-5)
                assert err <= (2e-5 if it == 0 else 2e-4), 'it=%d tower=%d %s: |dg| / max|g| = %.2e' % (it, t, k, err)
        # structural zeros of the block-diagonal FC must have exactly zero gradient
        flat = m.grad_tensor().cpu().numpy().reshape(m.G, m.stride)
        packed = m.pack(g).reshape(m.G, m.stride)
        np.testing.assert_array_equal(flat, packed)
        stats = np.zeros((scn.n_agent, 4))
        _lib.check(m._L.tsc_model_apply_grads(m._h, 5e-4, 1.0, stats.ctypes.data_as(C.c_void_p)))
        m.cur_t = 0
        onorm = o.apply_grads(ograds, 5e-4)
        np.testing.assert_allclose(stats[:, :3], ostats, rtol=2e-3, atol=1e-6)
        np.testing.assert_allclose(stats[:, 3], onorm, rtol=2e-3)
        p, op = m.get_tower_params(), o.tower_params()
        for t in range(m.G):
            for k in op[t]:
                np.testing.assert_allclose(p[t][k], op[t][k], atol=3e-5, err_msg='param tower=%d %s' % (t, k))
    m.close()


def test_returns_match_reference_buffer_on_gpu(golden_dir):
    """Feed the reference OnPolicyBuffer known answers through add_transition + returns kernel."""
    from deeprl_signal_control_amd import _lib
    g = np.load(os.path.join(golden_dir, 'learner_known_answers.npz'))
    for c in ('c0', 'c1', 'c2', 'c3'):
        r, v, dpost = g[c + '_r'], g[c + '_v'], g[c + '_done_post'].astype(np.uint8)
        T = len(r)
        scn, m, _ = _make('ma2c', 1, T, reward_norm=0.0, reward_clip=0.0)
        obs = torch.zeros(1, 25, scn.s_max, device='cuda')
        dpre = np.concatenate([[int(g[c + '_done0'])], dpost[:-1]]).astype(np.uint8)
        for t in range(T):
            m.add_transition(obs, torch.tensor([dpre[t]], dtype=torch.uint8, device='cuda'),
                             torch.zeros(1, 25, dtype=torch.int32, device='cuda'),
                             torch.full((1, 25), float(r[t]), dtype=torch.float64, device='cuda'),
                             torch.full((1, 25), float(v[t]), dtype=torch.float32, device='cuda'),
                             torch.tensor([dpost[t]], dtype=torch.uint8, device='cuda'))
        Rb = torch.full((1, 25), float(g[c + '_R']), dtype=torch.float32, device='cuda')
        _lib.check(m._L.tsc_model_compute_grads(m._h, C.c_void_p(Rb.data_ptr()), 0.01))
        Rs = np.zeros((T, 1, 25), np.float32); Advs = np.zeros_like(Rs)
        _lib.check(m._L.tsc_model_get_returns(m._h, Rs.ctypes.data_as(C.c_void_p), Advs.ctypes.data_as(C.c_void_p)))
        if abs(float(np.float32(g[c + '_R'])) - float(g[c + '_R'])) == 0 or dpost[-1]:
            np.testing.assert_array_equal(Rs[:, 0, 7], g[c + '_Rs'])
            np.testing.assert_array_equal(Advs[:, 0, 7], g[c + '_Advs'])
        else:                                                  # bootstrap passed as float32 (TF output)
            np.testing.assert_allclose(Rs[:, 0, 7], g[c + '_Rs'], rtol=1e-6)
            np.testing.assert_allclose(Advs[:, 0, 7], g[c + '_Advs'], rtol=1e-5, atol=1e-6)
        m.close()


def test_forward_sample_equals_forward_then_sample():
    scn, m, o = _make('ma2c', 40, 4, seed=3)
    rng = np.random.RandomState(9)
    obs = torch.from_numpy(_rand_obs(scn, 40, rng)).cuda()
    m.reset()
    pi, v, act = m.forward_sample(obs, True)
    pi, v, act = pi.clone(), v.clone(), act.clone()
    m.reset(); m.sample_step -= 1
    pi2, v2 = m.forward(obs, True, 'pv')
    act2 = m.sample(pi2)
    assert torch.equal(pi, pi2) and torch.equal(v, v2) and torch.equal(act, act2)
    m.close()


@pytest.mark.parametrize('agent,knob,policy', [('ma2c', 'TSC_UNFUSED_DW', 'lstm'), ('ia2c', 'TSC_UNFUSED_DW', 'lstm'),
                                               ('ma2c', 'TSC_UNFUSED_DX', 'lstm'), ('ia2c', 'TSC_UNFUSED_DX', 'lstm'),
                                               ('ia2c', 'TSC_UNFUSED_DX', 'fc')])
def test_fused_update_kernels_equal_grouped_gemms(agent, knob, policy, monkeypatch):
    """(policy 'fc': fc_bwd_kernel -- dWfc | dbfc | dW1 | db1 of the FcACPolicy in one pass -- against its three grouped GEMMs.)
    dwxh_kernel (dWx | dWh | dbl in one pass, whole tower output in accumulators) and dx1w1_kernel2 (dX1 kept in
    registers, dW1 | db1 from the same pass, only the structurally non-zero feature tiles of W1) against the grouped GEMMs
    they replace: same gradient up to fp32 summation order, structural zeros of W1 exactly zero."""
    E, T = 40, 9
    rng = np.random.RandomState(11)
    grads = []
    for unfused in ('0', '1'):
        monkeypatch.setenv(knob, unfused)
        scn, m, o = _make(agent, E, T, seed=5, policy=policy)
        m.reset(); o.reset()
        r2 = np.random.RandomState(123)
        obs, done = _fill(scn, m, o, E, T, r2, terminal=False, use_cache=True)
        Rb = m.forward(torch.from_numpy(obs).cuda(), False, 'v').clone()
        from deeprl_signal_control_amd import _lib
        _lib.check(m._L.tsc_model_compute_grads(m._h, C.c_void_p(Rb.data_ptr()), 0.01))
        grads.append(m.grad_tensor().cpu().numpy().copy())
        m.close()
    scale = np.abs(grads[1]).max()
    assert scale > 0
    np.testing.assert_allclose(grads[0], grads[1], atol=2e-5 * scale, rtol=0)
    np.testing.assert_array_equal(grads[0] == 0, grads[1] == 0)


def test_multibatch_trainer_keeps_replicas_identical():
    """MultiBatchTrainer: two half-batches on two streams; after an iteration every handle holds the same
    parameters, and they equal one VecA2C fed the summed gradient (the N-rank update rule)."""
    from deeprl_signal_control_amd.agents import VecA2C
    from deeprl_signal_control_amd.env import VecTrafficEnv
    from deeprl_signal_control_amd.scenario import build_large_grid
    from deeprl_signal_control_amd.trainer import MultiBatchTrainer
    scn = build_large_grid('ma2c')
    E, T = 8, 6
    cfg = {'batch_size': T, 'reward_norm': 2000.0}
    envs = [VecTrafficEnv(scn, E, seed=30 + 100 * b) for b in range(2)]
    models = [VecA2C(scn.n_s_ls, scn.n_a_ls, scn.n_w_ls, scn.n_f_ls, E, scn.s_max, 5, cfg, seed=3, name='ma2c')
              for b in range(2)]
    models[1].sample_seed = 51                                # same initial parameters, different action draws
    p0 = models[0].get_flat().copy()
    np.testing.assert_array_equal(p0, models[1].get_flat())
    tr = MultiBatchTrainer(envs, models)
    for _ in range(2):
        tr.run_iteration()
    torch.cuda.synchronize()
    pa, pb = models[0].get_flat(), models[1].get_flat()
    np.testing.assert_array_equal(pa, pb)
    assert np.abs(pa - p0).max() > 0
    for e in envs:
        e.close()
    for m in models:
        m.close()


def test_evaluation_path_perform_and_evaluate(tmp_path):
    """VecTrainer.perform / evaluate (utils.py:195-234, 257-275): deterministic evaluation is reproducible, each
    instance sees its own test seed, rewards are the un-shaped global reward, train mode is restored."""
    from deeprl_signal_control_amd.agents import VecA2C
    from deeprl_signal_control_amd.env import VecTrafficEnv
    from deeprl_signal_control_amd.scenario import build_large_grid
    from deeprl_signal_control_amd.trainer import VecTrainer
    scn = build_large_grid('ma2c')
    E = 4
    env = VecTrafficEnv(scn, E, seed=12, test_seeds=(10000, 20000))
    model = VecA2C(scn.n_s_ls, scn.n_a_ls, scn.n_w_ls, scn.n_f_ls, E, scn.s_max, 5, {'batch_size': 120}, seed=1, name='ma2c')
    tr = VecTrainer(env, model)
    env.train_mode = False
    m1, s1 = tr.perform([0, 1, 0, 1], 'deterministic')
    m2, s2 = tr.perform([0, 1, 0, 1], 'deterministic')
    np.testing.assert_array_equal(m1, m2)
    np.testing.assert_array_equal(s1, s2)
    assert m1[0] == m1[2] and m1[1] == m1[3] and m1[0] != m1[1]      # same seed -> same episode; seeds differ
    assert np.all(m1 < 0) and np.all(s1 > 0)
    env.train_mode = True
    rows = tr.evaluate('deterministic', output_path=str(tmp_path), step=7)
    assert env.train_mode is True
    assert [r['test_id'] for r in rows] == [0, 1]
    assert rows[0]['avg_reward'] == pytest.approx(m1[0]) and rows[1]['avg_reward'] == pytest.approx(m1[1])
    assert (tmp_path / 'train_reward.csv').read_text().splitlines()[0] == 'agent,avg_reward,std_reward,step,test_id'
    ms, _ = tr.perform(0, 'stochastic')
    assert ms.shape == (E,) and len(set(ms.tolist())) > 1            # sampled actions differ per instance
    tr.run_iteration()                                               # training resumes from a fresh episode
    env.close(); model.close()


# ---- parity on the BENCHMARKED shapes ("what's weak" 1-3) -------------------------------------------
def _read_cache(m, what, g, row0, nrows, width):
    from deeprl_signal_control_amd import _lib
    out = np.zeros((nrows, width), np.float32)
    _lib.check(m._L.tsc_model_debug_read(m._h, what, g, row0, nrows, out.ctypes.data_as(C.c_void_p)))
    return out


@pytest.mark.parametrize('scenario,agent,E,ws', [
    ('large_grid', 'ma2c', 1024, '1'),      # bench.py's shape: 5 workgroups per tower x 13 / 12 half tiles (6 tiles + a 16-instance one), 8-slot logits buffer
    ('large_grid', 'ma2c', 200, '1'),       # ragged: 13 half tiles over 5 workgroups (2 / 3 / 2 / 3 / 3), the last one holds 8 instances
    ('large_grid', 'ma2c', 1000, '1'),      # 63 half tiles: 12 / 13 / 12 / 13 / 13, the last one 8 instances
    ('large_grid', 'ma2c', 48, '1'),        # one workgroup per tower: a full tile + a half tile
    ('large_grid', 'ma2c', 240, '1'),       # 15 half tiles over 5 workgroups, 3 each: EVERY split ends in a half tile
    ('large_grid', 'ma2c', 16, '1'),        # nothing but a half tile
    ('large_grid', 'ia2c', 1024, '1'),      # H = 160 instantiation
    ('large_grid', 'ma2c', 200, '0'),       # TSC_FWD_WS=0: policy_fwd_fused_kernel, ragged 64-instance tile
    ('real_net', 'ma2c', 200, '1'),         # Monaco: H = 192 -> policy_fwd_ws_kernel<128>, n_a 2..6, no wait state
    ('real_net', 'ia2c', 96, '1'),
])
def test_forward_sample_multi_tile_vs_oracle(scenario, agent, E, ws, monkeypatch):
    """tsc_model_forward_sample for 3 consecutive slots on multi-tile batches against the float64 oracle: pi, v,
    the sampled action (documented uniform -> numpy choice) and the activation rows (X1, gates, c, h, masked
    h_prev) the kernel caches for the update, at every slot and for rows of every tile."""
    from oracle.nets_oracle import choice_from_uniform, sample_uniform, tower_activations, t64
    monkeypatch.setenv('TSC_FWD_WS', ws)
    T = 3
    scn, m, o = _make(agent, E, T, seed=7, scenario=scenario)
    A = scn.n_agent
    rng = np.random.RandomState(E + len(agent))
    m.reset(); o.reset()
    done = np.ones(E, np.uint8)
    rows = sorted(r for r in set(list(range(0, E, 37)) + [15, 16, 31, 32, 63, 64, E - 33, E - 17, E - 2, E - 1]) if 0 <= r < E)   # rows of every tile
    for t in range(T):
        obs = _rand_obs(scn, E, rng)
        s_before = [s.clone() for s in o.s_fw]
        pi, v, act = m.forward_sample(torch.from_numpy(obs).cuda(), torch.from_numpy(done).cuda())
        pi, v, act = pi.cpu().numpy(), v.cpu().numpy(), act.cpu().numpy()
        opi, ov = o.forward(obs, done, 'pv')
        for a in range(A):
            na = scn.n_a_ls[a]
            np.testing.assert_allclose(pi[:, a, :na], opi[a], atol=2e-5, err_msg='pi t=%d a=%d' % (t, a))
            assert np.all(pi[:, a, na:] == 0)
        np.testing.assert_allclose(v, ov, atol=2e-5)
        # the action is numpy's choice on the kernel's own pi (float32) and the documented uniform
        bad = 0
        for e in rows:
            for a in range(A):
                u = sample_uniform(m.sample_seed, m.sample_step - 1, e * A + a)
                bad += int(act[e, a] != choice_from_uniform(pi[e, a, :scn.n_a_ls[a]], u))
        assert bad == 0
        assert act.min() >= 0 and np.all(act < np.asarray(scn.n_a_ls)[None, :])
        # cached activations of this slot (row = t * E + e) for a few towers
        for g in (0, 1, 2 * (A // 2), 2 * A - 1):
            a = g // 2
            ref = tower_activations(o.p[g], o._ob(obs, a), t64(done.astype(np.float64)), s_before[g], o.nw[a], o.nt[a], o.nf[a])
            for what, key, width in ((0, 'X1', m.H), (1, 'gates', 256), (2, 'h', 64), (3, 'c', 64), (4, 'hprev', 64)):
                got = _read_cache(m, what, g, t * E, E, width)
                np.testing.assert_allclose(got, ref[key].numpy(), atol=3e-5, err_msg='cache %s t=%d g=%d' % (key, t, g))
        m.add_transition(torch.from_numpy(obs).cuda(), torch.from_numpy(done).cuda(), torch.from_numpy(act).cuda(),
                         torch.zeros(E, A, dtype=torch.float64, device='cuda'), torch.from_numpy(v).cuda(),
                         torch.zeros(E, dtype=torch.uint8, device='cuda'))
        done = (rng.rand(E) < 0.2).astype(np.uint8)
    m.close()


def _update_vs_oracle(scn, m, o, E, T, rng, iters, use_cache, terminal_first=False, gtol=2e-5):
    from deeprl_signal_control_amd import _lib
    worst = {}
    for it in range(iters):
        obs, done = _fill(scn, m, o, E, T, rng, terminal=terminal_first and it == 0, use_cache=use_cache)
        Rb = m.forward(torch.from_numpy(obs).cuda(), False, 'v').clone()
        _lib.check(m._L.tsc_model_compute_grads(m._h, C.c_void_p(Rb.data_ptr()), 0.01))
        ograds, ostats = o.compute_grads(Rb.cpu().numpy(), 0.01)
        Rs = np.zeros((T, E, scn.n_agent), np.float32); Advs = np.zeros_like(Rs)
        _lib.check(m._L.tsc_model_get_returns(m._h, Rs.ctypes.data_as(C.c_void_p), Advs.ctypes.data_as(C.c_void_p)))
        np.testing.assert_array_equal(Rs, o.Rs)
        np.testing.assert_array_equal(Advs, o.Advs)
        g = m.unpack(m.grad_tensor().cpu().numpy())
        for t in range(m.G):
            for k, og in ograds[t].items():
                err = _grad_err(o, t, k, g[t][k], og.numpy())
                worst[k] = max(worst.get(k, 0.0), err)
                assert err <= (gtol if it == 0 else 10 * gtol), 'it=%d tower=%d %s: |dg| / max|g| = %.2e' % (it, t, k, err)
        stats = np.zeros((scn.n_agent, 4))
        _lib.check(m._L.tsc_model_apply_grads(m._h, 5e-4, 1.0, stats.ctypes.data_as(C.c_void_p)))
        m.cur_t = 0
        onorm = o.apply_grads(ograds, 5e-4)
        np.testing.assert_allclose(stats[:, :3], ostats, rtol=2e-3, atol=1e-6)
        np.testing.assert_allclose(stats[:, 3], onorm, rtol=2e-3)
        p, op = m.get_tower_params(), o.tower_params()
        for t in range(m.G):
            for k in op[t]:
                np.testing.assert_allclose(p[t][k], op[t][k], atol=3e-5, err_msg='param tower=%d %s' % (t, k))
    return worst


def test_update_bench_shape_T120():
    """One update at the benchmark's n_step = 120 with E = 160 (five 32-instance forward tiles, three 64-instance
    BPTT tiles, 3840 rows per split of dwxh / dx1w1 = 120 chunks of 32 rows, split boundaries inside a time step) through the
    cached-activation path, against the float64 oracle.  Tolerance 1e-4 of the tensor's largest gradient entry
    (hidden units on a ReLU kink excepted, see _grad_err); a 32-row chunk lost at a split boundary would be 2e-3."""
    E, T = 160, 120
    scn, m, o = _make('ma2c', E, T, seed=5)
    m.reset(); o.reset()
    worst = _update_vs_oracle(scn, m, o, E, T, np.random.RandomState(3), 1, True, gtol=1e-4)
    print('T=120 E=160 worst |dg| / max|g|:', {k: '%.1e' % v for k, v in worst.items()})
    m.close()


@pytest.mark.parametrize('scenario,agent,policy,E,T,pick', [
    ('large_grid', 'ma2c', 'lstm', 1024, 120, None),      # BASELINE configs[2]: first, a four-neighbour and the last agent
    ('large_grid', 'ma2c', 'lstm', 1024, 120, (7, 12, 20)),      # three other agents (interior four-neighbour ones and an edge)
    ('large_grid', 'ia2c', 'lstm', 1024, 120, (1, 9, 18)),
    ('large_grid', 'ia2c', 'fc', 256, 120, (4, 13, 22)),         # BASELINE configs[1]
    ('real_net', 'ma2c', 'lstm', 512, 40, (6, 18, 25)),           # BASELINE configs[4] per GPU: Monaco, H = 192, n_step 40; agents with n_s 6 / n_a 2, n_a 6, n_s 50
])
def test_update_benchmarked_batch_E1024_T120(scenario, agent, policy, E, T, pick):
    """The update AT the benchmarked batch -- E = 1024 distinct instances, T = 120: 122 880 rows per agent-tower, the five
    row splits of dwxh / dx1w1 cover 24 576 rows each, lstm_bwd runs its full grid -- through the rollout path the
    benchmark uses (fused forward, activation cache), for MA2C (H = 224) and IA2C (H = 160: the dwxh<7> / dx1w1<5>
    instantiations).  The float64 oracle evaluates three agents (six towers: all the CPU can do in seconds) -- the first,
    a four-neighbour and the last one, or the three of `pick`, so that the parametrisations cover different agent
    indices (every index at once: test_update_at_the_benchmarked_batch_equals_the_replicated_small_batch_for_every_agent); their gradient
    slices, returns, losses, norms and updated parameters are compared.
    ('ia2c', 'fc', 256): BASELINE configs[1] -- FcACPolicy (agents/policies.py:214-256), 256 instances x 120 steps = 30 720
    rows per agent-tower through the training-shape forward and the one-pass first / second layer backward (fc_bwd_kernel).
    ('real_net', 'ma2c', 512, T = 40): what bench.py --config c5 times (config/config_ma2c_real.ini:1-46): 28 agents,
    policy_fwd_ws_kernel<128> with 16 tiles per tower over 4 workgroups, dwxh_kernel<8> / dx1w1_kernel2<12> at 20 480
    rows per tower, heterogeneous action counts (masked logits), reward_norm 1."""
    from deeprl_signal_control_amd import _lib
    from oracle.nets_oracle import OracleA2C
    scn, m, _ = _make(agent, E, T, seed=5, policy=policy, scenario=scenario)
    A = scn.n_agent
    sel = [0, 3, A - 1] if pick is None else list(pick)
    tw = m.get_tower_params()
    o = OracleA2C([tw[2 * a + k] for a in sel for k in (0, 1)], [m.n_wave_ls[a] for a in sel], [m.n_w_ls[a] for a in sel],
                  [m.n_f_ls[a] for a in sel], [m.n_a_ls[a] for a in sel], E, gamma=m.cfg['gamma'],
                  reward_norm=m.cfg['reward_norm'], reward_clip=m.cfg['reward_clip'], value_coef=m.cfg['value_coef'],
                  max_grad_norm=m.cfg['max_grad_norm'])
    rng = np.random.RandomState(17)
    m.reset(); o.reset()
    obs, done = _rand_obs(scn, E, rng), np.ones(E, np.uint8)
    for t in range(T):
        d_obs, d_done = torch.from_numpy(obs).cuda(), torch.from_numpy(done).cuda()
        pi, v, _ = m.forward_sample(d_obs, d_done)
        v = v.cpu().numpy()
        _, ov = o.forward(obs[:, sel], done, 'pv')
        np.testing.assert_allclose(v[:, sel], ov, atol=3e-5)
        act = np.stack([rng.randint(0, n, E) for n in scn.n_a_ls], 1).astype(np.int32)
        rew = -rng.rand(E, A) * 3.0 * m.cfg['reward_norm']
        dpost = (rng.rand(E) < 0.05).astype(np.uint8)
        m.add_transition(d_obs, d_done, torch.from_numpy(act).cuda(), torch.from_numpy(rew).cuda(), torch.from_numpy(v).cuda(),
                         torch.from_numpy(dpost).cuda())
        o.add_transition(obs[:, sel], done, act[:, sel], rew[:, sel], v[:, sel], dpost)
        obs, done = _rand_obs(scn, E, rng), dpost
    Rb = m.forward(torch.from_numpy(obs).cuda(), False, 'v').clone()
    _lib.check(m._L.tsc_model_compute_grads(m._h, C.c_void_p(Rb.data_ptr()), 0.01))
    ograds, ostats = o.compute_grads(Rb.cpu().numpy()[:, sel], 0.01)
    Rs = np.zeros((T, E, A), np.float32); Advs = np.zeros_like(Rs)
    _lib.check(m._L.tsc_model_get_returns(m._h, Rs.ctypes.data_as(C.c_void_p), Advs.ctypes.data_as(C.c_void_p)))
    np.testing.assert_array_equal(Rs[:, :, sel], o.Rs)
    np.testing.assert_array_equal(Advs[:, :, sel], o.Advs)
    g = m.unpack(m.grad_tensor().cpu().numpy())
    worst = {}
    for i, a in enumerate(sel):
        for k2 in (0, 1):
            for k, og in ograds[2 * i + k2].items():
                err = _grad_err(o, 2 * i + k2, k, g[2 * a + k2][k], og.numpy())
                worst[k] = max(worst.get(k, 0.0), err)
                assert err <= 1e-4, 'agent %d tower %d %s: |dg| / max|g| = %.2e' % (a, k2, k, err)
    stats = np.zeros((A, 4))
    _lib.check(m._L.tsc_model_apply_grads(m._h, 5e-4, 1.0, stats.ctypes.data_as(C.c_void_p)))
    m.cur_t = 0
    onorm = o.apply_grads(ograds, 5e-4)
    np.testing.assert_allclose(stats[sel, :3], ostats, rtol=2e-3, atol=1e-6)
    np.testing.assert_allclose(stats[sel, 3], onorm, rtol=2e-3)
    p, op = m.get_tower_params(), o.tower_params()
    for i, a in enumerate(sel):
        for k2 in (0, 1):
            for k in op[2 * i + k2]:
                np.testing.assert_allclose(p[2 * a + k2][k], op[2 * i + k2][k], atol=3e-5, err_msg='param agent=%d %s' % (a, k))
    print('%s E=%d T=%d %s %s agents %s worst |dg| / max|g|:' % (scenario, E, T, agent, policy, sel), {k: '%.1e' % v for k, v in worst.items()})
    m.close()


@pytest.mark.parametrize('scenario,E,T', [('large_grid', 1024, 120), ('real_net', 512, 40)])
def test_update_at_the_benchmarked_batch_equals_the_replicated_small_batch_for_every_agent(scenario, E, T):
    """Every agent index at the benchmarked batch (the float64 oracle above affords three agents of
    25 / 28).  The loss is a mean over the batch (agents/policies.py:41-61), so a batch that holds eight shuffled copies of
    a 128-instance (Monaco: 64-instance) rollout has the gradient of that rollout: the small rollout is inside what the
    all-agent oracle tests cover (E <= 160), the large one runs the benchmark's grid -- 5 (4) workgroups per tower in the
    forward, the five row splits of dwxh / dx1w1, lstm_bwd's full grid.  An agent-indexed layout slip that only shows at
    the large batch moves that agent's slice; the copies sit at shuffled instance indices, so an instance-indexed one
    moves them all.  Tolerance 2e-5 of each tensor's largest entry (float32 summation order is all that differs)."""
    from deeprl_signal_control_amd import _lib
    Es = E // 8
    scn, ms, _ = _make('ma2c', Es, T, seed=5, scenario=scenario)
    _, mb, _ = _make('ma2c', E, T, seed=5, scenario=scenario)
    mb.set_tower_params(ms.get_tower_params())
    A = scn.n_agent
    rng = np.random.RandomState(E + T)
    src = rng.permutation(E) % Es                              # instance e of the large batch is a copy of src[e]
    ms.reset(); mb.reset()
    obs, done = _rand_obs(scn, Es, rng), np.ones(Es, np.uint8)
    up = lambda x: torch.from_numpy(np.ascontiguousarray(x[src])).cuda()
    dn = lambda x: torch.from_numpy(x).cuda()
    for t in range(T):
        _, vs, _ = ms.forward_sample(dn(obs), dn(done))
        _, vb, _ = mb.forward_sample(up(obs), up(done))
        vs = vs.cpu().numpy()
        np.testing.assert_allclose(vb.cpu().numpy(), vs[src], atol=2e-6)
        act = np.stack([rng.randint(0, n, Es) for n in scn.n_a_ls], 1).astype(np.int32)
        rew = -rng.rand(Es, A) * 3.0 * ms.cfg['reward_norm']
        dpost = (rng.rand(Es) < 0.05).astype(np.uint8)
        ms.add_transition(dn(obs), dn(done), dn(act), dn(rew), dn(vs), dn(dpost))
        mb.add_transition(up(obs), up(done), up(act), up(rew), up(vs), up(dpost))
        obs, done = _rand_obs(scn, Es, rng), dpost
    Rs_ = ms.forward(dn(obs), False, 'v').clone()
    Rb_ = mb.forward(up(obs), False, 'v').clone()
    np.testing.assert_allclose(Rb_.cpu().numpy(), Rs_.cpu().numpy()[src], atol=2e-6)
    Rb_.copy_(Rs_[torch.from_numpy(src).cuda()])               # the same bootstrap values on both sides
    _lib.check(ms._L.tsc_model_compute_grads(ms._h, C.c_void_p(Rs_.data_ptr()), 0.01))
    _lib.check(mb._L.tsc_model_compute_grads(mb._h, C.c_void_p(Rb_.data_ptr()), 0.01))
    gs, gb = ms.unpack(ms.grad_tensor().cpu().numpy()), mb.unpack(mb.grad_tensor().cpu().numpy())
    worst = 0.0
    for g in range(ms.G):
        for k in gs[g]:
            scale = max(float(np.abs(gs[g][k]).max()), 1e-7)
            err = float(np.abs(gb[g][k] - gs[g][k]).max()) / scale
            worst = max(worst, err)
            assert err <= 2e-5, 'agent %d tower %d %s: |dg| / max|g| = %.2e' % (g // 2, g % 2, k, err)
    print('%s E=%d vs 8 x E=%d, T=%d, all %d agents: worst |dg| / max|g| = %.1e' % (scenario, E, Es, T, A, worst))
    ms.close(); mb.close()


@pytest.mark.parametrize('agent,E,T,use_cache', [('ma2c', 70, 40, True), ('ia2c', 40, 40, True), ('ma2c', 33, 10, False)])
def test_monaco_learner_vs_oracle(agent, E, T, use_cache):
    """Monaco (real_net) learner shapes -- H = 192 (MA2C: fw 128 + fp 64, no wait FC) / 128 (IA2C), heterogeneous
    n_a 2..6 with masked logits, n_step 40, reward_norm 1: forward (policy_fwd_fused_kernel), cached activations,
    grouped-GEMM weight gradients, clip, RMSProp over two consecutive updates vs the float64 oracle."""
    scn, m, o = _make(agent, E, T, seed=9, scenario='real_net')
    assert m.H == (192 if agent == 'ma2c' else 128) and max(scn.n_a_ls) == 6 and min(scn.n_a_ls) == 2
    m.reset(); o.reset()
    worst = _update_vs_oracle(scn, m, o, E, T, np.random.RandomState(E), 2, use_cache, terminal_first=True)
    print('monaco %s worst |dg| / max|g|:' % agent, {k: '%.1e' % v for k, v in worst.items()})
    m.close()


def test_set_params_keeps_optimizer_state_and_checkpoint_roundtrip(tmp_path):
    """set_params must not wipe RMSProp's accumulator; save/load restores parameters, accumulator,
    schedules and the action-RNG stream, and refuses a checkpoint of another layout."""
    scn, m, o = _make('ma2c', 8, 4, seed=2, lr_decay='linear', lr_min=1e-5)
    m.total_step = 1000; m._init_scheduler()
    rng = np.random.RandomState(0)
    m.reset(); o.reset()
    obs, done = _fill(scn, m, o, 8, 4, rng, use_cache=True)
    m.backward(m.forward(torch.from_numpy(obs).cuda(), False, 'v').clone())
    ms = m.get_flat('ms')
    assert np.abs(ms - 1.0).max() > 0
    m.set_tower_params(m.get_tower_params())
    np.testing.assert_array_equal(m.get_flat('ms'), ms)                  # untouched
    m.save(str(tmp_path), 120)
    p0, step0, n0 = m.get_flat(), m.sample_step, m.lr_scheduler.n
    scn2, m2, _ = _make('ma2c', 8, 4, seed=99, lr_decay='linear', lr_min=1e-5)
    m2.total_step = 1000; m2._init_scheduler()
    assert m2.load(str(tmp_path))
    np.testing.assert_array_equal(m2.get_flat(), p0)
    np.testing.assert_array_equal(m2.get_flat('ms'), ms)
    assert (m2.sample_step, m2.sample_seed, m2.lr_scheduler.n) == (step0, m.sample_seed, n0)
    assert m2.load(str(tmp_path), checkpoint=7) is False
    _, m3, _ = _make('ia2c', 8, 4, seed=1)
    with pytest.raises(ValueError, match='does not fit'):
        m3.load(str(tmp_path))
    for x in (m, m2, m3):
        x.close()
