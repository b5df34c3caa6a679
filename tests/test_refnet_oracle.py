"""Oracle-B on the CPU: oracle/nets_oracle.py (the float64 restatement the GPU parity tests check the HIP learner
against) replays the fixtures recorded from the REFERENCE'S OWN learner code executed over oracle/fake_tf.py
(tools/make_golden.py refnet -> tests/golden/refnet_*.npz; agents/models.py:174-229, agents/policies.py:41-61,99-155,
191-256, agents/utils.py:88-116,182-228 and Trainer.run of utils.py:255-308, all unmodified).

Pinned here: the reference's weights under np.random.seed (ortho_init in variable-creation order, agents/utils.py:11-24),
every forward (pi, v; 'pv' advancing the LSTM state, 'v' not), the float32 returns / advantages, the raw tf.gradients
of every variable, the per-agent global norm, the loss, and the variables + RMSProp `rms` slots after each update.
Tolerances: float64 against float64 -- 1e-9 relative (summation order only)."""
import os

import numpy as np
import pytest

from deeprl_signal_control_amd.agents import A2C_DEFAULTS, init_tower_params
from oracle import refnet
from oracle.nets_oracle import OracleA2C

GOLD = os.path.join(os.path.dirname(__file__), 'golden')
FIXTURES = ['refnet_ma2c_large', 'refnet_ia2c_large', 'refnet_fc_large', 'refnet_ma2c_real']


def load(name):
    z = np.load(os.path.join(GOLD, name + '.npz'))
    return {k: z[k] for k in z.files}


def model_cfg(fx):
    """[MODEL_CONFIG] of the reference INI the fixture was recorded with (config/config_{ma2c,ia2c}_{large,real}.ini)."""
    cfg = dict(A2C_DEFAULTS)
    cfg['batch_size'] = int(fx['n_step'])
    if str(fx['scenario']) == 'real_net':
        cfg['reward_norm'] = 1.0
    elif str(fx['agent']) == 'ia2c':
        cfg['reward_norm'] = 3000.0
    if str(fx['agent']) == 'ia2c' and str(fx['policy']) == 'lstm':
        cfg['max_grad_norm'] = 1.8                              # tools/make_golden.py refnet_ia2c_large
    return cfg


def dims(fx):
    n_s, n_w, n_f, n_a = (fx[k].tolist() for k in ('n_s_ls', 'n_w_ls', 'n_f_ls', 'n_a_ls'))
    n_wave = [s - w - f for s, w, f in zip(n_s, n_w, n_f)]
    ma2c = str(fx['agent']) == 'ma2c'
    n_fc = (128, 64 if ma2c else 0, 32 if max(n_w) > 0 else 0)
    return n_wave, n_w, n_f, n_a, n_fc


def initial_towers(fx):
    n_wave, n_w, n_f, n_a, n_fc = dims(fx)
    return init_tower_params(n_wave, n_w, n_f, n_a, n_fc, 64, str(fx['policy']), np.random.RandomState(int(fx['seed_w'])))


def check_digests(got_towers, names, rows, rtol, what, sums_only=False):
    want = refnet.unpack_digests(names, rows)
    got = refnet.tower_digest(got_towers, sums_only=sums_only)
    assert set(got) == set(want), what
    for k in want:
        scale = max(np.abs(want[k][3:]).max() if len(want[k]) > 3 else abs(want[k][2]), 1e-30)
        np.testing.assert_allclose(got[k][:3], want[k][:3], rtol=rtol, atol=rtol * max(want[k][1], 1e-30), err_msg='%s %s sums' % (what, k))
        if len(want[k]) > 3:
            np.testing.assert_allclose(got[k][3:], want[k][3:], rtol=0, atol=rtol * scale, err_msg='%s %s' % (what, k))


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
    """The IA2C fixture was recorded with max_grad_norm = 1.8 so that tf.clip_by_global_norm is not the identity."""
    fx = load('refnet_ia2c_large')
    assert (fx['bw0/norm'] > 1.8).sum() >= 3 and (fx['bw0/norm'] < 1.8).sum() >= 3
