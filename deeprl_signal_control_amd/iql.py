"""Host-side mirror of the reference's IQL agents (agents/models.py:264-376: IQL-LR ``model_type='lr'`` and IQL-DNN
``'dqn'``) over the HIP Q-learner (include/tsc.h tsc_iql_*, csrc/tsc_iql.hip).

* ``VecIQL`` -- all agents x E env instances, torch tensors in/out (device memory only).  Every instance keeps its own
  replay ring per agent; a minibatch step draws ``batch_size`` transitions from every ring, the loss averages over the
  E * batch_size rows of an agent (E = 1 is the reference).
* ``IQL`` -- the reference's duck-type for E = 1: ``forward(obs, mode, stochastic) -> (actions, qs)``,
  ``add_transition(obs, actions, rewards, next_obs, done)``, ``backward(summary_writer, global_step)``, ``reset``,
  ``save / load``, attrs ``n_step n_agent``.

Weights: every layer of the reference's Q nets is ``fc`` with its default ``ortho_init(sqrt(2))``
(agents/policies.py:299-303, agents/utils.py:66-74); restated by ``agents.ortho_init``.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from .agents import CKPT_FORMAT, resume_sample_seed, Scheduler, allreduce_grads_, coerce_config, ortho_init, replica_sample_seed

IQL_DEFAULTS = dict(max_grad_norm=40.0, gamma=0.99, lr_init=1e-4, lr_decay='constant', lr_min=0.0, epsilon_init=1.0,
                    epsilon_min=0.01, epsilon_decay='linear', epsilon_ratio=0.5, num_fc=128, num_h=64, batch_size=20,
                    buffer_size=1000.0, reward_norm=3000.0, reward_clip=2.0)     # config/config_iqld_large.ini
N_UPDATE = 10                                                                    # agents/models.py:324


class TscIqlCfg(C.Structure):
    _fields_ = [('n_agent', C.c_int32), ('s_max', C.c_int32), ('a_max', C.c_int32),
                ('n_wave', C.POINTER(C.c_int32)), ('n_wait', C.POINTER(C.c_int32)), ('n_act', C.POINTER(C.c_int32)),
                ('kind', C.c_int32), ('n_fc0', C.c_int32), ('n_h', C.c_int32), ('batch_size', C.c_int32),
                ('buffer_size', C.c_int32), ('gamma', C.c_double), ('reward_norm', C.c_double),
                ('reward_clip', C.c_double), ('max_grad_norm', C.c_double)]


def _setup_lib(L):
    if getattr(L, '_iql_ready', False):
        return
    vp = C.c_void_p
    L.tsc_iql_create.argtypes = [C.POINTER(TscIqlCfg), C.c_int32, C.c_int32, C.POINTER(vp)]
    L.tsc_iql_destroy.argtypes = [vp]
    L.tsc_iql_set_stream.argtypes = [vp, vp]
    L.tsc_iql_layout.argtypes = [vp, C.POINTER(C.c_int64)]
    L.tsc_iql_set_params.argtypes = [vp, vp]
    L.tsc_iql_get_params.argtypes = [vp, vp]
    L.tsc_iql_get_opt_state.argtypes = [vp, vp, vp, C.POINTER(C.c_int64)]
    L.tsc_iql_set_opt_state.argtypes = [vp, vp, vp, C.c_int64]
    L.tsc_iql_forward.argtypes = [vp, vp, vp, vp, C.c_int32, C.c_double, C.c_uint64, C.c_uint64]
    L.tsc_iql_add_transition.argtypes = [vp, vp, vp, vp, vp, vp]
    L.tsc_iql_replay_size.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.tsc_iql_compute_grads.argtypes = [vp, C.c_uint64, C.c_uint64]
    L.tsc_iql_compute_grads_at.argtypes = [vp, vp]
    L.tsc_iql_grad_buffer.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_int64)]
    L.tsc_iql_apply_grads.argtypes = [vp, C.c_double, C.c_double, vp]
    L.tsc_iql_debug_batch.argtypes = [vp, vp]
    L.tsc_iql_path.argtypes = [vp, C.POINTER(C.c_int32)]
    L.tsc_iql_debug_clock.argtypes = [vp, C.c_int32, vp, C.c_int32]
    L._iql_ready = True


class QParamLayout:
    """Flat per-agent layout of csrc/tsc_iql.hip (include/tsc.h tsc_iql_layout) <-> TF-variable-style dicts
    (q_fcw / q_fct / q_fc_0 / q of agents/policies.py:299-303,355-362)."""

    def __init__(self, n_wave_ls, n_w_ls, n_a_ls, s_max, kind, n_fc0, n_h, out_pad=8):
        self.n_wave_ls, self.n_w_ls, self.n_a_ls = list(n_wave_ls), list(n_w_ls), list(n_a_ls)
        self.s_max, self.kind, self.n_fc0, self.out_pad = int(s_max), kind, int(n_fc0), out_pad
        self.A = len(self.n_a_ls)
        self.ft = self.n_fc0 // 4 if (kind == 'dqn' and max(self.n_w_ls) > 0) else 0
        self.H1 = self.n_fc0 + self.ft if kind == 'dqn' else 0
        self.H2 = int(n_h) if kind == 'dqn' else 0
        if kind == 'dqn':
            self.oW1, self.ob1 = 0, self.s_max * self.H1
            self.oW2 = self.ob1 + self.H1
            self.ob2 = self.oW2 + self.H1 * self.H2
            self.oWq = self.ob2 + self.H2
            self.obq = self.oWq + self.H2 * out_pad
        else:
            self.oW1 = self.ob1 = self.oW2 = self.ob2 = 0
            self.oWq, self.obq = 0, self.s_max * out_pad
        self.stride = self.obq + out_pad
        self.n_param = self.A * self.stride

    def as_tuple(self):
        return (self.A, self.stride, self.H1, self.H2, self.oW1, self.ob1, self.oW2, self.ob2, self.oWq, self.obq,
                self.out_pad, 1 if self.kind == 'dqn' else 0)

    def shapes(self, a):
        nw, nt, na = self.n_wave_ls[a], self.n_w_ls[a], self.n_a_ls[a]
        if self.kind == 'lr':
            return {'q_w': (nw + nt, na), 'q_b': (na,)}
        sh = {'fcw_w': (nw, self.n_fc0), 'fcw_b': (self.n_fc0,)}
        if self.ft:
            sh.update({'fct_w': (nt, self.ft), 'fct_b': (self.ft,)})
        sh.update({'fc0_w': (self.H1, self.H2), 'fc0_b': (self.H2,), 'q_w': (self.H2, na), 'q_b': (na,)})
        return sh

    def pack(self, agents):
        flat = np.zeros((self.A, self.stride), np.float32)
        for a, p in enumerate(agents):
            nw, nt, na = self.n_wave_ls[a], self.n_w_ls[a], self.n_a_ls[a]
            f = flat[a]
            rows = self.H2 if self.kind == 'dqn' else self.s_max
            Wq = np.zeros((rows, self.out_pad), np.float32)
            Wq[:p['q_w'].shape[0], :na] = p['q_w']
            f[self.oWq:self.obq] = Wq.ravel()
            f[self.obq:self.obq + na] = p['q_b']
            if self.kind == 'dqn':
                W1 = np.zeros((self.s_max, self.H1), np.float32)
                b1 = np.zeros(self.H1, np.float32)
                W1[:nw, :self.n_fc0] = p['fcw_w']; b1[:self.n_fc0] = p['fcw_b']
                if self.ft:
                    W1[nw:nw + nt, self.n_fc0:] = p['fct_w']; b1[self.n_fc0:] = p['fct_b']
                f[self.oW1:self.ob1] = W1.ravel(); f[self.ob1:self.oW2] = b1
                f[self.oW2:self.ob2] = np.asarray(p['fc0_w'], np.float32).ravel(); f[self.ob2:self.oWq] = p['fc0_b']
        return flat.ravel()

    def unpack(self, flat):
        flat = np.asarray(flat, np.float32).reshape(self.A, self.stride)
        out = []
        for a in range(self.A):
            nw, nt, na = self.n_wave_ls[a], self.n_w_ls[a], self.n_a_ls[a]
            f = flat[a]
            rows = self.H2 if self.kind == 'dqn' else self.s_max
            Wq = f[self.oWq:self.obq].reshape(rows, self.out_pad)
            p = {'q_w': Wq[:(rows if self.kind == 'dqn' else nw + nt), :na].copy(), 'q_b': f[self.obq:self.obq + na].copy()}
            if self.kind == 'dqn':
                W1 = f[self.oW1:self.ob1].reshape(self.s_max, self.H1); b1 = f[self.ob1:self.oW2]
                p.update({'fcw_w': W1[:nw, :self.n_fc0].copy(), 'fcw_b': b1[:self.n_fc0].copy()})
                if self.ft:
                    p.update({'fct_w': W1[nw:nw + nt, self.n_fc0:].copy(), 'fct_b': b1[self.n_fc0:].copy()})
                p.update({'fc0_w': f[self.oW2:self.ob2].reshape(self.H1, self.H2).copy(), 'fc0_b': f[self.ob2:self.oWq].copy()})
            out.append(p)
        return out


def init_agent_params(layout, rng):
    """Initial Q-net weights of all agents in the reference's variable-creation order (q_fcw, q_fct, q_fc_0, q per agent,
    agents/policies.py:299-303,355-362; ortho_init is called when the variable is created): with
    rng = np.random.RandomState(s) the reference's weights under np.random.seed(s) (tests/test_refnet_oracle.py)."""
    return [{k: ortho_init(sh, rng) if len(sh) == 2 else np.zeros(sh, np.float32) for k, sh in layout.shapes(a).items()}
            for a in range(layout.A)]


class VecIQL:
    """IQL-LR / IQL-DNN for A agents x E env instances on one GPU."""

    def __init__(self, n_s_ls, n_a_ls, n_w_ls, n_env, s_max, a_max, model_config=None, total_step=0, device=0, seed=None,
                 model_type='dqn', process_group=None, replica=0):
        if not torch.cuda.is_available():
            raise RuntimeError('VecIQL needs a GPU (MI355X); there is no CPU fallback')
        if model_type not in ('dqn', 'lr'):
            raise ValueError("model_type must be 'dqn' or 'lr' (agents/models.py:298-307)")
        cfg = coerce_config(model_config, IQL_DEFAULTS)
        self.cfg, self.name, self.model_type = cfg, 'iql', model_type
        self.n_agent, self.E = len(n_s_ls), int(n_env)
        self.n_s_ls, self.n_a_ls, self.n_w_ls = list(n_s_ls), list(n_a_ls), list(n_w_ls)
        self.n_wave_ls = [s - w for s, w in zip(self.n_s_ls, self.n_w_ls)]
        self.n_step = int(cfg['batch_size'])
        self.s_max, self.a_max = int(s_max), int(a_max)
        self.device = torch.device('cuda', device) if not isinstance(device, torch.device) else device
        self.pg, self.total_step = process_group, total_step
        self._init_scheduler()
        L = _lib.lib()
        _setup_lib(L)
        self._L = L
        ip = C.POINTER(C.c_int32)
        self._arrs = [np.ascontiguousarray(x, np.int32) for x in (self.n_wave_ls, self.n_w_ls, self.n_a_ls)]
        mc = TscIqlCfg(self.n_agent, self.s_max, self.a_max, *[a.ctypes.data_as(ip) for a in self._arrs],
                       1 if model_type == 'dqn' else 0, int(cfg['num_fc']), int(cfg['num_h']), self.n_step,
                       int(float(cfg['buffer_size'])), float(cfg['gamma']), float(cfg['reward_norm']),
                       float(cfg['reward_clip']), float(cfg['max_grad_norm']))
        h = C.c_void_p()
        _lib.check(L.tsc_iql_create(C.byref(mc), self.E, self.device.index or 0, C.byref(h)))
        self._h = h
        lay = (C.c_int64 * 12)()
        _lib.check(L.tsc_iql_layout(h, lay))
        self.layout = QParamLayout(self.n_wave_ls, self.n_w_ls, self.n_a_ls, self.s_max, model_type, cfg['num_fc'], cfg['num_h'])
        assert self.layout.as_tuple() == tuple(int(x) for x in lay), 'host / device parameter layouts disagree'
        self.n_param = self.layout.n_param
        f = C.c_int32()
        _lib.check(L.tsc_iql_path(h, C.byref(f)))
        self.fused = bool(f.value)          # the one-kernel DeepQPolicy learner (csrc/tsc_iql_fused.h) or the grouped-GEMM path
        with torch.cuda.device(self.device):
            self.stream = torch.cuda.current_stream(self.device)
            _lib.check(L.tsc_iql_set_stream(h, C.c_void_p(self.stream.cuda_stream)))
            self.q = torch.zeros(self.E, self.n_agent, self.a_max, dtype=torch.float32, device=self.device)
            self.action = torch.zeros(self.E, self.n_agent, dtype=torch.int32, device=self.device)
        gp, cnt = C.c_void_p(), C.c_int64()
        _lib.check(L.tsc_iql_grad_buffer(h, C.byref(gp), C.byref(cnt)))
        assert int(cnt.value) == self.n_param
        self._grad_ptr, self._grad_t = gp.value, None
        dist = torch.distributed
        self.rank = dist.get_rank(process_group) if dist.is_available() and dist.is_initialized() else 0
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.base_seed, self.replica = (0 if seed is None else int(seed)), int(replica)
        self.sample_seed = replica_sample_seed(self.base_seed, self.rank, replica)
        self.replay_seed = self.sample_seed ^ 0x5DEECE66D
        self.act_step = 0           # forward(mode='explore' / stochastic) calls so far: the action-RNG counter
        self.update_step = 0        # minibatch steps so far: the replay-sampling counter
        self.init_params(seed)
        if self.world > 1:
            t = torch.from_numpy(self.get_flat()).to(self.device)
            dist.broadcast(t, src=0, group=self.pg)
            self.set_flat(t.cpu().numpy())

    def _init_scheduler(self):
        """agents/models.py:297-317."""
        c = self.cfg
        if c['lr_decay'] == 'constant':
            self.lr_scheduler = Scheduler(c['lr_init'], decay='constant')
        else:
            self.lr_scheduler = Scheduler(c['lr_init'], c['lr_min'], self.total_step, decay=c['lr_decay'])
        if c['epsilon_decay'] == 'constant':
            self.eps_scheduler = Scheduler(c['epsilon_init'], decay='constant')
        else:
            self.eps_scheduler = Scheduler(c['epsilon_init'], c['epsilon_min'], self.total_step * c['epsilon_ratio'],
                                           decay=c['epsilon_decay'])

    # ---- parameters -------------------------------------------------------------------------------------------
    def init_params(self, seed=None):
        rng = np.random.RandomState(seed) if seed is not None else np.random
        self.set_agent_params(init_agent_params(self.layout, rng))
        z = np.zeros(self.n_param, np.float32)
        _lib.check(self._L.tsc_iql_set_opt_state(self._h, z.ctypes.data_as(C.c_void_p), z.ctypes.data_as(C.c_void_p), 0))

    def set_flat(self, flat):
        flat = np.ascontiguousarray(flat, np.float32)
        assert flat.size == self.n_param
        _lib.check(self._L.tsc_iql_set_params(self._h, flat.ctypes.data_as(C.c_void_p)))

    def get_flat(self):
        out = np.zeros(self.n_param, np.float32)
        _lib.check(self._L.tsc_iql_get_params(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    def set_agent_params(self, agents):
        self.set_flat(self.layout.pack(agents))

    def get_agent_params(self):
        return self.layout.unpack(self.get_flat())

    def get_opt_state(self):
        m, v, t = np.zeros(self.n_param, np.float32), np.zeros(self.n_param, np.float32), C.c_int64()
        _lib.check(self._L.tsc_iql_get_opt_state(self._h, m.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p), C.byref(t)))
        return m, v, int(t.value)

    def grad_tensor(self):
        if self._grad_t is None:
            class _Holder:
                pass
            hold = _Holder()
            hold.__cuda_array_interface__ = {'shape': (self.n_param,), 'typestr': '<f4', 'data': (self._grad_ptr, False),
                                             'version': 3, 'strides': None}
            self._grad_t = torch.as_tensor(hold, device=self.device)
        return self._grad_t

    def use_stream(self, stream):
        self.stream = stream
        _lib.check(self._L.tsc_iql_set_stream(self._h, C.c_void_p(stream.cuda_stream)))

    # ---- reference API (batched) ------------------------------------------------------------------------------
    def reset(self):
        """agents/models.py:350-352: nothing to reset."""

    def forward(self, obs, mode='act', stochastic=False):
        """agents/models.py:332-348 -> (action int32 [E,A], qs f32 [E,A,AMAX]); both are the model's own buffers."""
        m, eps = 0, 0.0
        if mode == 'explore':
            m, eps = 1, float(self.eps_scheduler.get(1))
            self.last_eps = eps
        elif stochastic:
            m = 2
        _lib.check(self._L.tsc_iql_forward(self._h, C.c_void_p(obs.data_ptr()), C.c_void_p(self.q.data_ptr()),
                                           C.c_void_p(self.action.data_ptr()), m, eps, self.sample_seed, self.act_step))
        if m:
            self.act_step += 1
        return self.action, self.q

    def add_transition(self, obs, actions, rewards, next_obs, done):
        """agents/models.py:354-361; rewards f64 [E,A], done u8 [E] (device)."""
        if not torch.is_tensor(done):
            done = torch.full((self.E,), int(bool(done)), dtype=torch.uint8, device=self.device)
        _lib.check(self._L.tsc_iql_add_transition(self._h, C.c_void_p(obs.data_ptr()), C.c_void_p(actions.data_ptr()),
                                                  C.c_void_p(rewards.data_ptr()), C.c_void_p(next_obs.data_ptr()),
                                                  C.c_void_p(done.data_ptr())))

    def replay_size(self):
        s, c = C.c_int64(), C.c_int64()
        _lib.check(self._L.tsc_iql_replay_size(self._h, C.byref(s), C.byref(c)))
        return int(s.value), int(c.value)

    def minibatch_step(self, lr, want_stats=False):
        """One of the 10 minibatch updates of IQL.backward: sample, gradient, (all-reduce,) clip, Adam."""
        _lib.check(self._L.tsc_iql_compute_grads(self._h, self.replay_seed, self.update_step))
        self.update_step += 1
        scale = 1.0
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            with torch.cuda.stream(self.stream):
                scale = allreduce_grads_(self.grad_tensor(), self.pg)
        stats = np.zeros((self.n_agent, 2), np.float64) if want_stats else None
        _lib.check(self._L.tsc_iql_apply_grads(self._h, float(lr), float(scale),
                                               stats.ctypes.data_as(C.c_void_p) if want_stats else None))
        return stats

    def minibatch_step_at(self, idx, lr, want_stats=False, validate=True):
        """minibatch_step with the caller's draw: idx int32 [E, A, batch_size] (device), ring slots of every (instance,
        agent) -- ReplayBuffer.sample_transition's pick (include/tsc.h tsc_iql_compute_grads_at).  validate: check the
        draw against the filled part of the rings on the host (one device reduction + one blocking read; the library clamps
        the indices anyway) -- switch it off on a throughput path."""
        assert idx.dtype == torch.int32 and idx.is_contiguous() and tuple(idx.shape) == (self.E, self.n_agent, self.n_step)
        if validate:
            lo, hi = torch.stack(torch.aminmax(idx)).tolist()      # one reduction, one blocking read
            size = self.replay_size()[0]                           # host-side bookkeeping of the library, no device access
            if lo < 0 or hi >= size:    # an index nobody filled is a caller bug (random.sample cannot produce one)
                raise ValueError('minibatch_step_at: ring slots [%d, %d] outside the filled part [0, %d)' % (lo, hi, size))
        _lib.check(self._L.tsc_iql_compute_grads_at(self._h, C.c_void_p(idx.data_ptr())))
        self.update_step += 1
        stats = np.zeros((self.n_agent, 2), np.float64) if want_stats else None
        _lib.check(self._L.tsc_iql_apply_grads(self._h, float(lr), 1.0, stats.ctypes.data_as(C.c_void_p) if want_stats else None))
        return stats

    def backward(self, summary_writer=None, global_step=None, want_stats=False):
        """agents/models.py:319-330: nothing until the rings hold one batch, then 10 minibatch steps."""
        cur_lr = self.lr_scheduler.get(self.n_step)
        if self.replay_size()[0] < self.n_step:
            return None
        stats = None
        for _ in range(N_UPDATE):
            stats = self.minibatch_step(cur_lr, want_stats)
        return stats

    # ---- checkpoints (agents/models.py:83-108) ----------------------------------------------------------------
    def save(self, model_dir, global_step):
        os.makedirs(model_dir, exist_ok=True)
        m, v, t = self.get_opt_state()
        np.savez(os.path.join(model_dir, 'checkpoint-%d.npz' % int(global_step)), params=self.get_flat(), adam_m=m, adam_v=v,
                 layout=np.array(self.layout.as_tuple() + (self.s_max,), np.int64),
                 counters=np.array([t, self.act_step, self.update_step, self.base_seed, self.lr_scheduler.n, self.eps_scheduler.n], np.int64),
                 format=np.int64(CKPT_FORMAT))

    def load(self, model_dir, checkpoint=None):
        save_file, save_step = None, 0
        if os.path.exists(model_dir):
            if checkpoint is None:
                for f in os.listdir(model_dir):
                    if f.startswith('checkpoint'):
                        tokens = f.split('.')[0].split('-')
                        if len(tokens) == 2 and int(tokens[1]) > save_step:
                            save_file, save_step = f, int(tokens[1])
            else:
                save_file = 'checkpoint-%d.npz' % int(checkpoint)
        if save_file is None or not os.path.exists(os.path.join(model_dir, save_file)):
            return False
        z = np.load(os.path.join(model_dir, save_file))
        want = self.layout.as_tuple() + (self.s_max,)
        if 'adam_m' not in z.files or tuple(int(x) for x in z['layout']) != want or z['params'].size != self.n_param:
            raise ValueError('checkpoint %s does not fit this model' % save_file)
        self.set_flat(z['params'])
        m, v = np.ascontiguousarray(z['adam_m'], np.float32), np.ascontiguousarray(z['adam_v'], np.float32)
        c = [int(x) for x in z['counters']]
        _lib.check(self._L.tsc_iql_set_opt_state(self._h, m.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p), c[0]))
        self.act_step, self.update_step = c[1], c[2]
        base, self.sample_seed = resume_sample_seed(c[3], z['format'] if 'format' in z.files else None, self.rank, self.replica)
        self.base_seed = self.base_seed if base is None else base
        self.replay_seed = self.sample_seed ^ 0x5DEECE66D
        self.lr_scheduler.n, self.eps_scheduler.n = c[4], c[5]
        return True

    def close(self):
        if getattr(self, '_h', None):
            self._L.tsc_iql_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class IQL:
    """E = 1 duck-type of agents/models.py:264-376 (lists of per-agent arrays in and out)."""

    def __init__(self, n_s_ls, n_a_ls, n_w_ls, total_step, model_config, seed=0, model_type='dqn', device=0):
        s_max = (max(n_s_ls) + 3) // 4 * 4
        self.vec = VecIQL(n_s_ls, n_a_ls, n_w_ls, 1, s_max, max(n_a_ls), model_config, total_step, device=device, seed=seed,
                          model_type=model_type)
        self.name, self.model_type = 'iql', model_type
        self.n_agent, self.n_step = len(n_s_ls), self.vec.n_step
        self.n_s_ls, self.n_a_ls, self.n_w_ls = list(n_s_ls), list(n_a_ls), list(n_w_ls)
        self.sess = None
        d = self.vec.device
        self._obs = [torch.zeros(1, self.n_agent, s_max, dtype=torch.float32, device=d) for _ in range(2)]

    def _put(self, obs, which=0):
        o = np.zeros((1, self.n_agent, self.vec.s_max), np.float32)
        for a, ob in enumerate(obs):
            o[0, a, :len(ob)] = np.asarray(ob, np.float32)
        self._obs[which].copy_(torch.from_numpy(o))
        return self._obs[which]

    def forward(self, obs, mode='act', stochastic=False):
        act, q = self.vec.forward(self._put(obs), mode, stochastic)
        act, q = act[0].cpu().numpy(), q[0].cpu().numpy()
        return [int(x) for x in act], [q[a, :n].copy() for a, n in enumerate(self.n_a_ls)]

    def add_transition(self, obs, actions, rewards, next_obs, done):
        d = self.vec.device
        r = np.broadcast_to(np.asarray(rewards, np.float64), (self.n_agent,)).reshape(1, -1)
        self.vec.add_transition(self._put(obs, 0), torch.tensor([list(map(int, actions))], dtype=torch.int32, device=d),
                                torch.tensor(np.ascontiguousarray(r), device=d), self._put(next_obs, 1),
                                torch.tensor([int(bool(done))], dtype=torch.uint8, device=d))

    def backward(self, summary_writer=None, global_step=None):
        return self.vec.backward(summary_writer, global_step)

    def reset(self):
        self.vec.reset()

    def save(self, model_dir, global_step):
        self.vec.save(model_dir, global_step)

    def load(self, model_dir, checkpoint=None):
        return self.vec.load(model_dir, checkpoint)
