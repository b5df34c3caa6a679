"""Oracle-B on the CPU: oracle/nets_oracle.py (the float64 restatement the GPU parity tests check the HIP learner
against) replays the fixtures recorded from the REFERENCE'S OWN learner code executed over oracle/fake_tf.py
(tools/make_golden.py refnet -> tests/golden/refnet_*.npz; agents/models.py:174-229, agents/policies.py:41-61,99-155,
191-256, agents/utils.py:88-116,182-228 and Trainer.run of utils.py:255-308, all unmodified).

Pinned here: the reference's weights under np.random.seed (ortho_init in variable-creation order, agents/utils.py:11-24),
every forward (pi, v; 'pv' advancing the LSTM state, 'v' not), the float32 returns / advantages, the raw tf.gradients
of every variable, the per-agent global norm, the loss, and the variables + RMSProp `rms` slots after each update.
Tolerances: float64 against float64 -- 1e-9 relative (summation order only)."""
import numpy as np
import pytest

from oracle import refnet
from oracle.nets_oracle import OracleA2C

FIXTURES = refnet.A2C_FIXTURES
load, model_cfg, dims, initial_towers, check_digests = (refnet.load_fixture, refnet.fixture_model_cfg, refnet.fixture_dims,
                                                        refnet.initial_towers, refnet.check_digests)


@pytest.mark.parametrize('name', FIXTURES)
def test_reference_weights_under_seed(name):
    """a17: init_tower_params(RandomState(s)) IS the reference's ortho_init under np.random.seed(s) -- every variable of
    every agent, float32-exact."""
    fx = load(name)
    check_digests(initial_towers(fx), fx['w0/names'], fx['w0/rows'], 0.0, 'w0')


def replay(fx, model_factory, on_forward, on_backward):
    """Drive `model` (OracleA2C API) through the recorded episode."""
    m = model_factory()
    n_step, A = int(fx['n_step']), len(fx['n_a_ls'])
    m.reset()
    t = bw = 0
    for i, typ in enumerate(fx['fw_type']):
        obs = fx['fw_obs'][i][None]
        pis, v = m.forward(obs, bool(fx['fw_done'][i]), str(typ))
        on_forward(i, str(typ), pis, v)
        if str(typ) == 'pv':
            m.add_transition(obs, np.array([bool(fx['fw_done'][i])]), fx['actions'][t][None], fx['reward'][t][None], v,
                             np.array([fx['done'][t]]))
            t += 1
            if t % n_step == 0 and fx['done'][t - 1]:
                on_backward(m, bw, np.zeros((1, A)))             # utils.py:186-188: R = [0] * n_agent
                bw += 1
        else:
            on_backward(m, bw, np.asarray(v, np.float32).astype(np.float64).reshape(1, A))   # fetched values are float32
            bw += 1
    assert bw == int(fx['n_backward']) and t == len(fx['actions'])
    return m


@pytest.mark.parametrize('name', FIXTURES)
def test_oracle_replays_reference_learner(name):
    fx = load(name)
    n_wave, n_w, n_f, n_a, n_fc = dims(fx)
    cfg = model_cfg(fx)
    A = len(n_a)

    def factory():
        return OracleA2C(initial_towers(fx), n_wave, n_w, n_f, n_a, 1, gamma=cfg['gamma'], reward_norm=cfg['reward_norm'],
                         reward_clip=cfg['reward_clip'], value_coef=cfg['value_coef'], max_grad_norm=cfg['max_grad_norm'],
                         alpha=cfg['rmsp_alpha'], eps=cfg['rmsp_epsilon'], state_f32=True)

    def on_forward(i, typ, pis, v):
        if 'p' in typ:
            for a in range(A):
                np.testing.assert_allclose(pis[a][0], fx['fw_pi'][i, a, :n_a[a]], rtol=0, atol=1e-12)
        if 'v' in typ:
            np.testing.assert_allclose(np.asarray(v)[0], fx['fw_v'][i], rtol=1e-10, atol=1e-12)

    def on_backward(m, k, R):
        p = 'bw%d/' % k
        np.testing.assert_allclose(R[0], fx[p + 'R'], rtol=0, atol=1e-6)         # the reference holds float32 values
        grads, stats = m.compute_grads(R, float(np.float32(fx[p + 'beta'])))      # float32 placeholders (policies.py:45,58)
        np.testing.assert_array_equal(m.Rs[:, 0, :], fx[p + 'Rs'])              # float32, bit-exact
        np.testing.assert_array_equal(m.Advs[:, 0, :], fx[p + 'Advs'])
        np.testing.assert_allclose(stats.sum(1), fx[p + 'loss'], rtol=1e-10)
        check_digests([{k_: g.numpy() for k_, g in t.items()} for t in grads], fx[p + 'g/names'], fx[p + 'g/rows'], 1e-9, p + 'g')
        norms = m.apply_grads(grads, float(np.float32(fx[p + 'lr'])))
        np.testing.assert_allclose(norms, fx[p + 'norm'], rtol=1e-10)
        check_digests(m.tower_params_f64(), fx[p + 'w/names'], fx[p + 'w/rows'], 1e-10, p + 'w')
        check_digests([{k_: v.numpy() for k_, v in t.items()} for t in m.ms], fx[p + 'ms/names'], fx[p + 'ms/rows'], 1e-10,
                      p + 'ms', sums_only=True)
        if fx.get(p + 'states_bw') is not None and p + 'states_bw' in fx:        # policies.py:153
            sb = np.stack([np.stack([m.s_bw[2 * a][0].numpy(), m.s_bw[2 * a + 1][0].numpy()]) for a in range(A)])
            np.testing.assert_allclose(sb, fx[p + 'states_bw'], rtol=0, atol=1e-6)

    replay(fx, factory, on_forward, on_backward)


def test_clip_bites_in_the_ia2c_fixture():
    """The IA2C fixture was recorded with a max_grad_norm inside the range of its gradient norms, so that
    tf.clip_by_global_norm is the identity for some agents and not for others."""
    fx = load('refnet_ia2c_large')
    clip = refnet.fixture_model_cfg(fx)['max_grad_norm']
    assert (fx['bw0/norm'] > clip).sum() >= 3 and (fx['bw0/norm'] < clip).sum() >= 3


# ---- IQL-LR / IQL-DNN (agents/models.py:264-376, agents/policies.py:285-389, agents/utils.py:231-263) ---------------------
IQL_FIXTURES = ['refnet_iqll_large', 'refnet_iqld_large']


def iql_layout(fx):
    from deeprl_signal_control_amd.iql import QParamLayout
    n_s, n_w, n_a = (fx[k].tolist() for k in ('n_s_ls', 'n_w_ls', 'n_a_ls'))
    kind = 'dqn' if str(fx['agent']) == 'iqld' else 'lr'
    n_wave = [s - w for s, w in zip(n_s, n_w)]
    return QParamLayout(n_wave, n_w, n_a, (max(n_s) + 3) // 4 * 4, kind, 128, 64), n_wave, n_w, n_a


def iql_initial(fx):
    from deeprl_signal_control_amd.iql import init_agent_params
    return init_agent_params(iql_layout(fx)[0], np.random.RandomState(int(fx['seed_w'])))


def agent_digest(agents, sums_only=False):
    return {'%d/%s' % (a, k): refnet.digest(v)[:refnet.N_SUMS if sums_only else None] for a, p in enumerate(agents) for k, v in p.items()}


def check_agent_digests(got, names, rows, tol, what):
    want = refnet.unpack_digests(names, rows)
    assert set(got) == set(want), what
    for k in want:
        n = refnet.N_SUMS
        np.testing.assert_allclose(got[k][:n], want[k][:n], rtol=tol, atol=tol * max(want[k][1], 1e-30), err_msg='%s %s sums' % (what, k))
        if len(want[k]) > n:
            assert np.abs(got[k][n:] - want[k][n:]).max() <= tol * max(want[k][3], 1e-30), (what, k)


@pytest.mark.parametrize('name', IQL_FIXTURES)
def test_oracle_replays_reference_iql(name):
    """The float64 restatement (oracle/iql_oracle.py) against the reference IQL executed over oracle/fake_tf.py: initial
    weights under the seed, Q values of every forward, the epsilon schedule, and for each of the 3 x 10 minibatch steps per
    agent (the reference's own random.sample draws) loss and global norm; raw gradients of step 0 and 9; weights and both
    Adam moments after every backward."""
    from oracle.iql_oracle import OracleIQL
    fx = load(name)
    lay, n_wave, n_w, n_a = iql_layout(fx)
    A = len(n_a)
    w0 = iql_initial(fx)
    check_agent_digests(agent_digest(w0), fx['w0/names'], fx['w0/rows'], 0.0, 'w0')
    o = OracleIQL(w0, n_wave, n_w, n_a, 1, batch_size=int(fx['n_step']), buffer_size=1000, reward_norm=3000.0, reward_clip=2.0,
                  max_grad_norm=40.0)
    from deeprl_signal_control_amd.agents import Scheduler
    T = len(fx['actions'])
    eps = Scheduler(1.0, 0.01, T * 0.5, decay='linear')                          # config_iql*_large.ini, total_step = T
    bw = 0
    S = lay.s_max
    for t in range(T):
        obs = np.zeros((1, A, S)); obs[0, :, :fx['fw_obs'].shape[2]] = fx['fw_obs'][t]
        nxt = np.zeros((1, A, S)); nxt[0, :, :fx['next_obs'].shape[2]] = fx['next_obs'][t]
        qs = o.forward(obs)
        for a in range(A):
            np.testing.assert_allclose(qs[a][0], fx['fw_q'][t, a, :n_a[a]], rtol=0, atol=1e-12)
        assert abs(eps.get(1) - fx['fw_eps'][t]) < 1e-12
        o.add_transition(obs, fx['actions'][t][None], fx['reward'][t][None], nxt, np.array([fx['done'][t]]))
        if (t + 1) % int(fx['n_step']) == 0:
            p = 'bw%d/' % bw
            for k in range(10):
                loss, norm, grads = o.minibatch_step(float(np.float32(fx[p + 'lr'])), idx_given=fx[p + 'idx'][k][None])
                np.testing.assert_allclose(loss, fx[p + 'loss'][k], rtol=1e-9, atol=1e-18)
                np.testing.assert_allclose(norm, fx[p + 'norm'][k], rtol=1e-9, atol=1e-18)
                if k in (0, 9):
                    check_agent_digests(agent_digest(grads), fx[p + 'g%d/names' % k], fx[p + 'g%d/rows' % k], 1e-9, p + 'g%d' % k)
            check_agent_digests(agent_digest([{k_: v.numpy() for k_, v in q.p.items()} for q in o.qs]), fx[p + 'w/names'], fx[p + 'w/rows'], 1e-10, p + 'w')
            check_agent_digests(agent_digest([{k_: v.numpy() for k_, v in q.m.items()} for q in o.qs], True), fx[p + 'm/names'], fx[p + 'm/rows'], 1e-9, p + 'm')
            check_agent_digests(agent_digest([{k_: v.numpy() for k_, v in q.v.items()} for q in o.qs], True), fx[p + 'v/names'], fx[p + 'v/rows'], 1e-9, p + 'v')
            bw += 1
    assert bw == int(fx['n_backward']) == 3
