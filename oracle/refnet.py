"""Oracle-B driver -- TEST INFRASTRUCTURE ONLY (oracle/__init__.py).

Runs the reference's learner UNMODIFIED (agents/models.py, agents/policies.py, agents/utils.py, the loop in utils.py)
over oracle/fake_tf.py, on the reference's env classes over oracle/fake_traci.py, and records what its own code
computes: every `model.forward` (pi, v, LSTM states, float64 next to the float32 the caller sees), `env.step`,
`add_transition`, and for every `model.backward` the float64 returns / advantages, losses, raw `tf.gradients`, the
global norms and the variables + RMSProp / Adam slots after the update.  Only usable where /root/reference exists
(tools/make_golden.py `refnet`); the fixtures it writes travel to the GPU box.

Also the small pure helpers the replay tests share: the reference's TF variable names -> this repo's tower-dict keys,
and the digest (sums + sampled entries) that stands for a multi-megabyte tensor in a fixture.
"""
import os
import re
import sys
import tempfile

import numpy as np

REFERENCE_ROOT = '/root/reference'
_LAYER = {'fcw': 'fcw', 'fcf': 'fcf', 'fct': 'fct', 'fc': 'fc'}


def ref_var_key(name):
    """'fplstm_3a/pi_lstm/wx' -> (agent 3, tower 0, 'lstm_wx');  '<policy>_<i>a/<pi|v>[_<layer>]/<w|b|wx|wh>'
    (scopes of agents/policies.py:89-96 and agents/utils.py:66-74,96-102).  IQL: 'dqn_2a_q/q_fcw/w' -> (2, 0, 'fcw_w'),
    'lr_0a_q/q/w' -> (0, 0, 'q_w'), 'q_fc_0' -> 'fc0'."""
    m = re.match(r'(?:fplstm|lstm|fc|fpfc)_(\d+)a/(pi|v)(?:_(\w+))?/(\w+)$', name)
    if m:
        a, tower, layer, leaf = int(m.group(1)), 0 if m.group(2) == 'pi' else 1, m.group(3), m.group(4)
        if layer is None:
            return a, tower, 'out_' + leaf
        if layer == 'lstm':
            return a, tower, 'lstm_' + leaf
        return a, tower, '%s_%s' % (_LAYER[layer], leaf)
    m = re.match(r'(?:dqn|lr)_(\d+)a_q/q(?:_(\w+))?/(\w+)$', name)
    if m:
        a, layer, leaf = int(m.group(1)), m.group(2), m.group(3)
        key = {None: 'q', 'fcw': 'fcw', 'fct': 'fct', 'fc_0': 'fc0'}[layer]
        return a, 0, '%s_%s' % (key, leaf)
    raise ValueError('unknown reference variable %r' % name)


N_SAMPLE = 24
N_SUMS = 4


def digest_index(size):
    """Fixed pseudo-random positions of a flat tensor of `size` elements (shared by generator and tests)."""
    if size <= N_SAMPLE:
        return np.arange(size)
    return np.sort(np.random.RandomState(size).choice(size, N_SAMPLE, replace=False))


def digest(arr):
    """[sum, sum |x|, sqrt(sum x^2), max |x|, sampled entries...] in float64."""
    a = np.asarray(arr, np.float64).ravel()
    return np.concatenate([[a.sum(), np.abs(a).sum(), np.sqrt((a * a).sum()), np.abs(a).max()], a[digest_index(a.size)]])


def tower_digest(towers, sums_only=False):
    """{'<g>/<key>': digest} for a list of per-tower dicts (order agent0 pi, agent0 v, agent1 pi, ...)."""
    return {'%d/%s' % (g, k): digest(v)[:N_SUMS if sums_only else None] for g, p in enumerate(towers) for k, v in p.items()}


def pack_digests(d):
    """{name: digest} -> (names [n] str, rows [n, N_SUMS + N_SAMPLE] float64, NaN-padded): one array instead of thousands."""
    names = sorted(d)
    width = max(len(d[k]) for k in names)
    rows = np.full((len(names), width), np.nan)
    for i, k in enumerate(names):
        rows[i, :len(d[k])] = d[k]
    return np.array(names), rows


def unpack_digests(names, rows):
    return {str(k): r[~np.isnan(r)] for k, r in zip(names, rows)}


GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
A2C_FIXTURES = ['refnet_ma2c_large', 'refnet_ia2c_large', 'refnet_fc_large', 'refnet_ma2c_real']


def load_fixture(name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    return {k: z[k] for k in z.files}


def fixture_model_cfg(fx):
    """[MODEL_CONFIG] of the reference INI a fixture was recorded with (config/config_{ma2c,ia2c}_{large,real}.ini) plus
    the overrides of tools/make_golden.py refnet_fixtures."""
    from deeprl_signal_control_amd.agents import A2C_DEFAULTS
    cfg = dict(A2C_DEFAULTS)
    cfg['batch_size'] = int(fx['n_step'])
    if str(fx['scenario']) == 'real_net':
        cfg['reward_norm'] = 1.0
    elif str(fx['agent']) == 'ia2c':
        cfg['reward_norm'] = 3000.0
    if str(fx['agent']) == 'ia2c' and str(fx['policy']) == 'lstm':
        cfg.update(max_grad_norm=2.1, lr_decay='linear', lr_min=1e-4, entropy_decay='linear', entropy_coef_min=0.002, entropy_ratio=1.0)
    return cfg


def fixture_dims(fx):
    """-> n_wave_ls, n_w_ls, n_f_ls, n_a_ls, (num_fw, num_fp, num_ft)."""
    n_s, n_w, n_f, n_a = (fx[k].tolist() for k in ('n_s_ls', 'n_w_ls', 'n_f_ls', 'n_a_ls'))
    n_wave = [s - w - f for s, w, f in zip(n_s, n_w, n_f)]
    n_fc = (128, 64 if str(fx['agent']) == 'ma2c' else 0, 32 if max(n_w) > 0 else 0)
    return n_wave, n_w, n_f, n_a, n_fc


def initial_towers(fx):
    """The reference's initial weights, regenerated from the recorded np.random seed (pinned by the fixture's digests)."""
    from deeprl_signal_control_amd.agents import init_tower_params
    n_wave, n_w, n_f, n_a, n_fc = fixture_dims(fx)
    return init_tower_params(n_wave, n_w, n_f, n_a, n_fc, 64, str(fx['policy']), np.random.RandomState(int(fx['seed_w'])))


def check_digests(got_towers, names, rows, tol, what, sums_only=False, sum_tol=None):
    """Sampled entries within tol * max|tensor|, the sums within sum_tol (default tol) relative."""
    want = unpack_digests(names, rows)
    got = tower_digest(got_towers, sums_only=sums_only)
    assert set(got) == set(want), what
    sum_tol = tol if sum_tol is None else sum_tol
    worst = 0.0
    for k in want:
        np.testing.assert_allclose(got[k][:N_SUMS], want[k][:N_SUMS], rtol=sum_tol, atol=sum_tol * max(want[k][1], 1e-30),
                                   err_msg='%s %s sums' % (what, k))
        if len(want[k]) > N_SUMS:
            scale = max(want[k][3], 1e-30)
            err = np.abs(got[k][N_SUMS:] - want[k][N_SUMS:]).max() / scale
            worst = max(worst, err)
            assert err <= tol, '%s %s: %.3g > %.3g' % (what, k, err, tol)
    return worst


# ---- everything below needs /root/reference ---------------------------------------------------------------------
def _install():
    from oracle import fake_tf, fake_traci
    fake_tf.install()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    for mod in [m for m in sys.modules if m == 'agents' or m.startswith('agents.') or m == 'utils']:
        if getattr(sys.modules[mod], '__file__', '') and REFERENCE_ROOT in (sys.modules[mod].__file__ or ''):
            del sys.modules[mod]                # agents/utils.py binds act=tf.nn.relu at import: re-import over fake_tf
    if not hasattr(np, 'bool'):
        np.bool = bool                          # agents/utils.py:226
    return fake_tf, fake_traci


def graph_towers(fake_tf, n_agent, slot=None, optimizers=None):
    """The graph's variables (or one optimizer slot of them) as this repo's per-tower dicts, float64."""
    towers = [dict() for _ in range(2 * n_agent)]
    for name, var in fake_tf.get_default_graph().variables.items():
        a, t, key = ref_var_key(name)
        if slot is None:
            val = var.value
        else:
            val = optimizers[a].slots[var][slot]
        towers[2 * a + t][key] = val.numpy().copy()
    return towers


def graph_agents(fake_tf, n_agent, slot=None, optimizers=None):
    """IQL: one dict per agent."""
    agents = [dict() for _ in range(n_agent)]
    for name, var in fake_tf.get_default_graph().variables.items():
        a, _, key = ref_var_key(name)
        agents[a][key] = (var.value if slot is None else optimizers[a].slots[var][slot]).numpy().copy()
    return agents


class _Tap:
    """Wrap a bound method: record(args, kwargs, result) after calling the original."""

    def __init__(self, obj, name, before=None, after=None):
        self.orig = getattr(obj, name)
        self.before, self.after = before, after
        setattr(obj, name, self)

    def __call__(self, *a, **k):
        if self.before:
            self.before(*a, **k)
        out = self.orig(*a, **k)
        if self.after:
            new = self.after(out, *a, **k)
            if new is not None:
                out = new
        return out


def run_reference_a2c(scenario, agent, seed_w, episode_sec, policy='lstm', model_over=None, full_agents=()):
    """main.train (main.py:82-127) for one shortened episode: the reference env, the reference IA2C / MA2C and the
    reference Trainer.run, instrumented.  policy='fc' swaps in the reference's FcACPolicy (agents/policies.py:214-256,
    which agents/models.py never instantiates) through IA2C._init_policy.
    -> dict of arrays (the fixture)."""
    fake_tf, fake_traci = _install()
    from deeprl_signal_control_amd.scenario import build_scenario
    cfg = fake_traci.ref_config(scenario, agent)
    cfg['ENV_CONFIG']['episode_length_sec'] = str(episode_sec)
    for k, v in (model_over or {}).items():
        cfg['MODEL_CONFIG'][k] = str(v)
    scn = build_scenario(scenario, agent, episode_length_sec=episode_sec)
    env = fake_traci.ref_env(scenario, agent, scn=scn, config=cfg)
    import agents.models as ref_models
    from utils import Counter, Trainer
    n_step = cfg['MODEL_CONFIG'].getint('batch_size')
    T = episode_sec // cfg['ENV_CONFIG'].getint('control_interval_sec')
    total_step = T
    seed = cfg.getint('ENV_CONFIG', 'seed')
    np.random.seed(seed_w)                      # ortho_init draws from the global stream (agents/utils.py:19)
    if agent == 'ia2c':
        cls = ref_models.IA2C
        if policy == 'fc':
            class IA2CFc(ref_models.IA2C):
                def _init_policy(self, n_s, n_a, n_w, n_f, model_config, agent_name=None):
                    return ref_models.FcACPolicy(n_s, n_a, n_w, self.n_step, n_fc_wave=model_config.getint('num_fw'),
                                                 n_fc_wait=model_config.getint('num_ft'),
                                                 n_lstm=model_config.getint('num_lstm'), name=agent_name)

                def reset(self):                # FcACPolicy is stateless and has no _reset (IA2C.reset would raise)
                    pass
            cls = IA2CFc
        model = cls(env.n_s_ls, env.n_a_ls, env.n_w_ls, total_step, cfg['MODEL_CONFIG'], seed=seed)
    else:
        model = ref_models.MA2C(env.n_s_ls, env.n_a_ls, env.n_w_ls, env.n_f_ls, total_step, cfg['MODEL_CONFIG'], seed=seed)
    A = model.n_agent
    rec = dict(fw_obs=[], fw_done=[], fw_type=[], fw_pi=[], fw_v=[], fw_pi32=[], fw_v32=[], actions=[], reward=[],
               global_reward=[], done=[], fp=[], bw=[])
    amax = max(env.n_a_ls)
    smax = max(env.n_s_ls)
    w0 = graph_towers(fake_tf, A)
    sess = model.sess

    def fw_before(obs, done, out_type='pv'):
        sess.trace = []

    def fw_after(out, obs, done, out_type='pv'):
        o = np.zeros((A, smax), np.float32)
        for a, ob in enumerate(obs):
            o[a, :len(ob)] = np.asarray(ob, np.float32)
        pi = np.zeros((A, amax)); v = np.zeros(A)
        for a, vals in enumerate(sess.trace):
            if 'p' in out_type:
                pi[a, :env.n_a_ls[a]] = vals[0]
            if 'v' in out_type:
                v[a] = vals[1] if 'p' in out_type else vals[0]
        rec['fw_obs'].append(o); rec['fw_done'].append(bool(done)); rec['fw_type'].append(out_type)
        rec['fw_pi'].append(pi); rec['fw_v'].append(v)
        sess.trace = None
        if out_type == 'v':
            # NumPy 1.x emulation (the reference is Python 3.5 / NumPy 1.x): the bootstrap values come back as
            # np.float32 scalars and `self.gamma * R` (agents/utils.py:207) is a float64 product there; NumPy 2's weak
            # Python scalars would make it float32.  Same values, float64 type.
            return [np.float64(x) for x in out]
    _Tap(model, 'forward', fw_before, fw_after)

    def step_after(out, action):
        ob, r, done, g = out
        rec['actions'].append([int(x) for x in action]); rec['reward'].append(np.asarray(r, np.float64).copy())
        rec['global_reward'].append(float(g)); rec['done'].append(bool(done))
    _Tap(env, 'step', None, step_after)
    if agent == 'ma2c':
        def fp_after(out, pol):
            p = np.zeros((A, amax), np.float32)
            for a, x in enumerate(pol):
                p[a, :len(x)] = x
            rec['fp'].append(p)
        _Tap(env, 'update_fingerprint', None, fp_after)

    bundles = {tuple(id(x) for x in b.xs): b for b in fake_tf.get_default_graph().grad_bundles}
    cur = {}
    for a, pol in enumerate(model.policy_ls):
        def bw_after(out, sess_, obs, acts, dones, Rs, Advs, cur_lr, cur_beta, summary_writer=None, global_step=None, a=a, pol=pol):
            memo = sess.last_memo
            wts = fake_tf.trainable_variables(scope=pol.name)
            raw = memo[id(bundles[tuple(id(x) for x in wts)].node)]
            g = {}
            for var, gr in zip(wts, raw):
                _, t, key = ref_var_key(var.var_name)
                g['%d/%s' % (2 * a + t, key)] = gr.numpy().copy()
            cur.setdefault('grads', {}).update(g)
            cur.setdefault('Rs', []).append(np.asarray(Rs)); cur.setdefault('Advs', []).append(np.asarray(Advs))
            cur.setdefault('dones', []).append(np.asarray(dones, np.float64)); cur.setdefault('acts', []).append(np.asarray(acts))
            cur.setdefault('norm', []).append(float(memo[id(pol.grad_norm)].detach()))
            cur.setdefault('loss', []).append(float(memo[id(pol.loss)].detach()))
            cur['lr'], cur['beta'] = float(cur_lr), float(cur_beta)
        _Tap(pol, 'backward', None, bw_after)

    def model_bw_before(R_ls, *a, **k):
        cur.clear()
        cur['R'] = np.asarray(R_ls, np.float64).copy()

    def model_bw_after(out, R_ls, *a, **k):
        opts = [p.optimizer for p in model.policy_ls]
        b = dict(R=cur['R'], Rs=np.stack(cur['Rs'], 1), Advs=np.stack(cur['Advs'], 1), dones_pre=cur['dones'][0],
                 acts=np.stack(cur['acts'], 1), norm=np.array(cur['norm']), loss=np.array(cur['loss']), lr=cur['lr'],
                 beta=cur['beta'], grads=dict(cur['grads']), w=graph_towers(fake_tf, A),
                 ms=graph_towers(fake_tf, A, 'rms', opts),
                 states_bw=np.stack([p.states_bw for p in model.policy_ls]) if hasattr(model.policy_ls[0], 'states_bw') else None)
        rec['bw'].append(b)
    _Tap(model, 'backward', model_bw_before, model_bw_after)

    out_dir = tempfile.mkdtemp(prefix='tsc_refnet_') + '/'
    trainer = Trainer(env, model, Counter(total_step, 10 ** 9, 10 ** 9), fake_tf.summary.FileWriter(out_dir), False,
                      output_path=out_dir)
    trainer.run()

    fx = dict(scenario=scenario, agent=agent, policy=policy, seed_w=seed_w, episode_sec=episode_sec, n_step=n_step,
              env_seed=seed, n_s_ls=np.array(env.n_s_ls), n_a_ls=np.array(env.n_a_ls), n_w_ls=np.array(env.n_w_ls),
              n_f_ls=np.array(env.n_f_ls if agent == 'ma2c' else [0] * A),
              fw_obs=np.array(rec['fw_obs']), fw_done=np.array(rec['fw_done']), fw_type=np.array(rec['fw_type']),
              fw_pi=np.array(rec['fw_pi']), fw_v=np.array(rec['fw_v']), actions=np.array(rec['actions'], np.int32),
              reward=np.array(rec['reward']), global_reward=np.array(rec['global_reward']), done=np.array(rec['done']),
              train_reward_csv=open(out_dir + 'train_reward.csv').read())
    if rec['fp']:
        fx['fp'] = np.array(rec['fp'])
    fx['w0/names'], fx['w0/rows'] = pack_digests(tower_digest(w0))
    for i, b in enumerate(rec['bw']):
        p = 'bw%d/' % i
        for k in ('R', 'Rs', 'Advs', 'dones_pre', 'acts', 'norm', 'loss', 'lr', 'beta'):
            fx[p + k] = np.asarray(b[k])
        if b['states_bw'] is not None:
            fx[p + 'states_bw'] = b['states_bw']
        fx[p + 'g/names'], fx[p + 'g/rows'] = pack_digests({k: digest(v) for k, v in b['grads'].items()})
        for k, v in b['grads'].items():
            if int(k.split('/')[0]) // 2 in full_agents and i == 0:
                fx[p + 'gfull/' + k] = np.asarray(v, np.float32)
        fx[p + 'w/names'], fx[p + 'w/rows'] = pack_digests(tower_digest(b['w']))
        fx[p + 'ms/names'], fx[p + 'ms/rows'] = pack_digests(tower_digest(b['ms'], sums_only=True))
    fx['n_backward'] = len(rec['bw'])
    return fx


def run_reference_iql(agent, seed_w, episode_sec, scenario='large_grid'):
    """main.train for agent = iqll / iqld (main.py:117-122): the reference env, the reference IQL (LRQPolicy / DeepQPolicy,
    ReplayBuffer, AdamOptimizer) and Trainer.run, instrumented.  `random.sample` draws transitions, not indices
    (agents/utils.py:252-258); the indices are recovered by replaying the same draw on range(len(buffer)) from the saved
    generator state (checked to leave the generator where the reference left it)."""
    import random
    fake_tf, fake_traci = _install()
    from deeprl_signal_control_amd.scenario import build_scenario
    cfg = fake_traci.ref_config(scenario, agent, 'config_%s_large.ini' % agent)
    cfg['ENV_CONFIG']['episode_length_sec'] = str(episode_sec)
    scn = build_scenario(scenario, agent, episode_length_sec=episode_sec)
    env = fake_traci.ref_env(scenario, agent, scn=scn, config=cfg)
    import agents.models as ref_models
    from utils import Counter, Trainer
    n_step = cfg['MODEL_CONFIG'].getint('batch_size')
    T = episode_sec // cfg['ENV_CONFIG'].getint('control_interval_sec')
    np.random.seed(seed_w)
    random.seed(seed_w)
    model = ref_models.IQL(env.n_s_ls, env.n_a_ls, env.n_w_ls, T, cfg['MODEL_CONFIG'], seed=0,
                           model_type='dqn' if agent == 'iqld' else 'lr')
    A = model.n_agent
    amax, smax = max(env.n_a_ls), max(env.n_s_ls)
    w0 = graph_agents(fake_tf, A)
    sess = model.sess
    rec = dict(fw_obs=[], fw_q=[], fw_eps=[], actions=[], reward=[], next_obs=[], done=[], global_reward=[], bw=[])

    def pad(obs):
        o = np.zeros((A, smax), np.float32)
        for a, ob in enumerate(obs):
            o[a, :len(ob)] = np.asarray(ob, np.float32)
        return o

    def fw_before(obs, mode='act', stochastic=False):
        sess.trace = []

    def fw_after(out, obs, mode='act', stochastic=False):
        q = np.zeros((A, amax))
        for a, vals in enumerate(sess.trace):
            q[a, :env.n_a_ls[a]] = vals[0].reshape(-1)
        rec['fw_obs'].append(pad(obs)); rec['fw_q'].append(q)
        sess.trace = None
    _Tap(model, 'forward', fw_before, fw_after)
    _Tap(model.eps_scheduler, 'get', None, lambda out, n: rec['fw_eps'].append(float(out)))      # agents/models.py:334

    def step_after(out, action):
        ob, r, done, g = out
        rec['actions'].append([int(x) for x in action]); rec['reward'].append(np.asarray(r, np.float64).copy())
        rec['next_obs'].append(pad(ob)); rec['done'].append(bool(done)); rec['global_reward'].append(float(g))
    _Tap(env, 'step', None, step_after)

    cur = {}
    for a, buf in enumerate(model.trans_buffer_ls):
        def st_before(a=a, buf=buf):
            cur['state'] = random.getstate()

        def st_after(out, a=a, buf=buf):
            after = random.getstate()
            random.setstate(cur['state'])
            idx = random.sample(range(len(buf.buffer)), buf.batch_size)
            assert random.getstate() == after
            obs = out[0]
            for j, i in enumerate(idx):                   # the recovered indices name the sampled transitions
                assert np.array_equal(obs[j], buf.buffer[i][0])
            cur.setdefault('idx', {}).setdefault(a, []).append(idx)
        _Tap(buf, 'sample_transition', st_before, st_after)
    bundles = {tuple(id(x) for x in b.xs): b for b in fake_tf.get_default_graph().grad_bundles}
    for a, pol in enumerate(model.policy_ls):
        def bw_after(out, sess_, obs, acts, next_obs, dones, rs, cur_lr, summary_writer=None, global_step=None, a=a, pol=pol):
            memo = sess.last_memo
            wts = fake_tf.trainable_variables(scope=pol.name)
            raw = memo[id(bundles[tuple(id(x) for x in wts)].node)]
            k = len(cur.setdefault('loss', {}).setdefault(a, []))
            cur['loss'][a].append(float(memo[id(pol.loss)].detach()))
            cur.setdefault('norm', {}).setdefault(a, []).append(float(memo[id(pol.grad_norm)].detach()))
            if k in (0, 9):
                for var, gr in zip(wts, raw):
                    _, _, key = ref_var_key(var.var_name)
                    cur.setdefault('g%d' % k, {})['%d/%s' % (a, key)] = gr.numpy().copy()
            cur['lr'] = float(cur_lr)
        _Tap(pol, 'backward', None, bw_after)

    def model_bw_before(*a, **k):
        cur.clear()

    def model_bw_after(out, *a, **k):
        if 'idx' not in cur:
            return
        opts = [p.optimizer for p in model.policy_ls]
        rec['bw'].append(dict(idx=np.array([cur['idx'][a_] for a_ in range(A)]).transpose(1, 0, 2),      # [10, A, B]
                              loss=np.array([cur['loss'][a_] for a_ in range(A)]).T, norm=np.array([cur['norm'][a_] for a_ in range(A)]).T,
                              lr=cur['lr'], g0=dict(cur['g0']), g9=dict(cur['g9']), w=graph_agents(fake_tf, A),
                              m=graph_agents(fake_tf, A, 'm', opts), v=graph_agents(fake_tf, A, 'v', opts)))
    _Tap(model, 'backward', model_bw_before, model_bw_after)

    out_dir = tempfile.mkdtemp(prefix='tsc_refnet_') + '/'
    Trainer(env, model, Counter(T, 10 ** 9, 10 ** 9), fake_tf.summary.FileWriter(out_dir), False, output_path=out_dir).run()

    def agent_digest(agents, sums_only=False):
        return {'%d/%s' % (a, k): digest(v)[:N_SUMS if sums_only else None] for a, p in enumerate(agents) for k, v in p.items()}
    fx = dict(scenario=scenario, agent=agent, seed_w=seed_w, episode_sec=episode_sec, n_step=n_step,
              n_s_ls=np.array(env.n_s_ls), n_a_ls=np.array(env.n_a_ls), n_w_ls=np.array(env.n_w_ls),
              fw_obs=np.array(rec['fw_obs']), fw_q=np.array(rec['fw_q']), fw_eps=np.array(rec['fw_eps']),
              actions=np.array(rec['actions'], np.int32), reward=np.array(rec['reward']), next_obs=np.array(rec['next_obs']),
              done=np.array(rec['done']), global_reward=np.array(rec['global_reward']), n_backward=len(rec['bw']))
    fx['w0/names'], fx['w0/rows'] = pack_digests(agent_digest(w0))
    for i, b in enumerate(rec['bw']):
        p = 'bw%d/' % i
        for k in ('idx', 'loss', 'norm', 'lr'):
            fx[p + k] = np.asarray(b[k])
        for k in ('g0', 'g9'):
            fx[p + k + '/names'], fx[p + k + '/rows'] = pack_digests({kk: digest(v) for kk, v in b[k].items()})
        fx[p + 'w/names'], fx[p + 'w/rows'] = pack_digests(agent_digest(b['w']))
        fx[p + 'm/names'], fx[p + 'm/rows'] = pack_digests(agent_digest(b['m'], True))
        fx[p + 'v/names'], fx[p + 'v/rows'] = pack_digests(agent_digest(b['v'], True))
    return fx
