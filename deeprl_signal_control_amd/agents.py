"""Host-side model mirror of the reference's IA2C / MA2C (agents/models.py:132-262) over the
HIP nets (include/tsc.h tsc_model_*, csrc/tsc_model.hip).

* ``VecA2C`` -- all agents x E env instances, torch tensors in/out (device memory only).
* ``IA2C`` / ``MA2C`` -- the reference's duck-type for E = 1
  (``forward/backward/add_transition/reset/save/load``, attrs ``n_step n_agent``), so a
  ``utils.Trainer``-style loop can drive them unchanged.

Weight init restates ``ortho_init`` (agents/utils.py:11-24, scale sqrt(2), biases 0); like the
reference it draws from ``np.random`` -- pass ``seed`` to make it reproducible (the reference's
is not).  Parity tests inject weights through ``set_agent_params``.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib

A2C_DEFAULTS = dict(rmsp_alpha=0.99, rmsp_epsilon=1e-5, max_grad_norm=40.0, gamma=0.99, lr_init=5e-4,
                    lr_decay='constant', entropy_coef_init=0.01, entropy_coef_min=0.01,
                    entropy_decay='constant', entropy_ratio=0.5, value_coef=0.5, num_fw=128, num_ft=32,
                    num_lstm=64, num_fp=64, batch_size=120, reward_norm=2000.0, reward_clip=2.0,
                    lr_min=0.0)     # config/config_ma2c_large.ini [MODEL_CONFIG]


class TscModelCfg(C.Structure):
    _fields_ = [('n_agent', C.c_int32), ('s_max', C.c_int32), ('a_max', C.c_int32),
                ('n_wave', C.POINTER(C.c_int32)), ('n_wait', C.POINTER(C.c_int32)),
                ('n_fp', C.POINTER(C.c_int32)), ('n_act', C.POINTER(C.c_int32)),
                ('n_fc_wave', C.c_int32), ('n_fc_wait', C.c_int32), ('n_fc_fp', C.c_int32),
                ('n_lstm', C.c_int32), ('n_step', C.c_int32),
                ('gamma', C.c_double), ('reward_norm', C.c_double), ('reward_clip', C.c_double),
                ('value_coef', C.c_double), ('max_grad_norm', C.c_double), ('rmsp_alpha', C.c_double),
                ('rmsp_epsilon', C.c_double), ('policy_kind', C.c_int32)]


class Scheduler:
    """agents/utils.py:268-281."""

    def __init__(self, val_init, val_min=0, total_step=0, decay='linear'):
        self.val, self.N, self.val_min, self.decay, self.n = val_init, float(total_step), val_min, decay, 0

    def get(self, n_step):
        self.n += n_step
        if self.decay == 'linear':
            return max(self.val_min, self.val * (1 - self.n / self.N))
        return self.val


def ortho_init(shape, rng, scale=np.sqrt(2)):
    """agents/utils.py:11-24 for 2-D shapes."""
    a = rng.standard_normal(shape)
    u, _, v = np.linalg.svd(a, full_matrices=False)
    q = u if u.shape == tuple(shape) else v
    return (scale * q.reshape(shape)).astype(np.float32)


def tower_param_shapes(n_wave, n_wait, n_fp, n_fc, n_lstm, policy='lstm'):
    """TF-variable-style shapes of one tower (agents/policies.py:99-118,191-211,227-235), in the order the reference
    creates the variables: fcw, [fcf], [fct], then lstm wx / wh / b (or fc); the head comes last."""
    fw, fp, ft = n_fc
    sh = {'fcw_w': (n_wave, fw), 'fcw_b': (fw,)}
    if fp:
        sh.update({'fcf_w': (n_fp, fp), 'fcf_b': (fp,)})
    if ft:
        sh.update({'fct_w': (n_wait, ft), 'fct_b': (ft,)})
    H = fw + fp + ft
    if policy == 'lstm':
        sh.update({'lstm_wx': (H, 4 * n_lstm), 'lstm_wh': (n_lstm, 4 * n_lstm), 'lstm_b': (4 * n_lstm,)})
    else:
        sh.update({'fc_w': (H, n_lstm), 'fc_b': (n_lstm,)})
    return sh


def init_tower_params(n_wave_ls, n_w_ls, n_f_ls, n_a_ls, n_fc, n_lstm, policy, rng):
    """Initial weights of all towers (agent-major, pi then v) drawn in the reference's variable-creation order:
    `get_variable` calls ortho_init when the variable is created (agents/utils.py:66-74,96-102), policies are built
    agent by agent, pi tower before v tower (agents/models.py:150-153, agents/policies.py:91-93).  With
    rng = np.random.RandomState(s) this reproduces the reference's weights under np.random.seed(s)
    (tests/test_refnet_oracle.py)."""
    towers = []
    for a in range(len(n_a_ls)):
        for tower in ('pi', 'v'):
            p = {}
            for k, sh in tower_param_shapes(n_wave_ls[a], n_w_ls[a], n_f_ls[a], n_fc, n_lstm, policy).items():
                p[k] = ortho_init(sh, rng) if len(sh) == 2 else np.zeros(sh, np.float32)
            n_out = n_a_ls[a] if tower == 'pi' else 1
            p['out_w'] = ortho_init((n_lstm, n_out), rng)
            p['out_b'] = np.zeros(n_out, np.float32)
            towers.append(p)
    return towers


class ParamLayout:
    """Flat parameter layout of csrc/tsc_model.hip (see include/tsc.h tsc_model_layout): group
    g = 2*agent + tower owns `stride` floats  W1[s_max][H] | b1[H] | Wx[H][4L] | Wh[L][4L] | bl[4L] |
    Wo[L][8] | bo[8]."""

    def __init__(self, n_wave_ls, n_w_ls, n_f_ls, n_a_ls, s_max, n_fc, n_lstm=64, out_pad=8, policy='lstm'):
        self.policy = policy
        self.n_wave_ls, self.n_w_ls, self.n_f_ls, self.n_a_ls = map(list, (n_wave_ls, n_w_ls, n_f_ls, n_a_ls))
        self.s_max, self.n_fc, self.Lh, self.out_pad = int(s_max), tuple(n_fc), int(n_lstm), int(out_pad)
        self.G = 2 * len(self.n_a_ls)
        self.H = sum(self.n_fc)
        L4 = 4 * self.Lh if policy == 'lstm' else self.Lh      # second-layer width (gates | fc units)
        self.NZ = L4
        self.oW1 = 0
        self.ob1 = self.s_max * self.H
        self.oWx = self.ob1 + self.H
        self.oWh = self.oWx + self.H * L4
        self.obl = self.oWh + (self.Lh * L4 if policy == 'lstm' else 0)
        self.oWo = self.obl + L4
        self.obo = self.oWo + self.Lh * self.out_pad
        self.stride = self.obo + self.out_pad
        self.n_param = self.G * self.stride

    def as_tuple(self):
        return (self.G, self.stride, self.H, self.Lh, self.oW1, self.ob1, self.oWx, self.oWh, self.obl, self.oWo,
                self.obo, self.out_pad)

    def pack(self, towers):
        """List of per-tower dicts (order: agent0 pi, agent0 v, agent1 pi, ...) -> flat [G*stride]."""
        flat = np.zeros((self.G, self.stride), np.float32)
        fw, fp, ft = self.n_fc
        for g, p in enumerate(towers):
            a = g // 2
            nw, nt, nf = self.n_wave_ls[a], self.n_w_ls[a], self.n_f_ls[a]
            W1 = np.zeros((self.s_max, self.H), np.float32)
            b1 = np.zeros(self.H, np.float32)
            W1[:nw, :fw] = p['fcw_w']; b1[:fw] = p['fcw_b']
            if fp:
                W1[nw + nt:nw + nt + nf, fw:fw + fp] = p['fcf_w']; b1[fw:fw + fp] = p['fcf_b']
            if ft:
                W1[nw:nw + nt, fw + fp:] = p['fct_w']; b1[fw + fp:] = p['fct_b']
            Wo = np.zeros((self.Lh, self.out_pad), np.float32)
            bo = np.zeros(self.out_pad, np.float32)
            Wo[:, :p['out_w'].shape[1]] = p['out_w']; bo[:len(p['out_b'])] = p['out_b']
            f = flat[g]
            f[self.oW1:self.ob1] = W1.ravel(); f[self.ob1:self.oWx] = b1
            if self.policy == 'lstm':
                f[self.oWx:self.oWh] = np.asarray(p['lstm_wx'], np.float32).ravel()
                f[self.oWh:self.obl] = np.asarray(p['lstm_wh'], np.float32).ravel()
                f[self.obl:self.oWo] = p['lstm_b']
            else:
                f[self.oWx:self.oWh] = np.asarray(p['fc_w'], np.float32).ravel()
                f[self.obl:self.oWo] = p['fc_b']
            f[self.oWo:self.obo] = Wo.ravel(); f[self.obo:] = bo
        return flat.ravel()

    def unpack(self, flat):
        flat = np.asarray(flat, np.float32).reshape(self.G, self.stride)
        fw, fp, ft = self.n_fc
        towers = []
        for g in range(self.G):
            a = g // 2
            nw, nt, nf = self.n_wave_ls[a], self.n_w_ls[a], self.n_f_ls[a]
            f = flat[g]
            W1 = f[self.oW1:self.ob1].reshape(self.s_max, self.H); b1 = f[self.ob1:self.oWx]
            p = {'fcw_w': W1[:nw, :fw].copy(), 'fcw_b': b1[:fw].copy()}
            if fp:
                p['fcf_w'] = W1[nw + nt:nw + nt + nf, fw:fw + fp].copy(); p['fcf_b'] = b1[fw:fw + fp].copy()
            if ft:
                p['fct_w'] = W1[nw:nw + nt, fw + fp:].copy(); p['fct_b'] = b1[fw + fp:].copy()
            if self.policy == 'lstm':
                p['lstm_wx'] = f[self.oWx:self.oWh].reshape(self.H, 4 * self.Lh).copy()
                p['lstm_wh'] = f[self.oWh:self.obl].reshape(self.Lh, 4 * self.Lh).copy()
                p['lstm_b'] = f[self.obl:self.oWo].copy()
            else:
                p['fc_w'] = f[self.oWx:self.oWh].reshape(self.H, self.Lh).copy()
                p['fc_b'] = f[self.obl:self.oWo].copy()
            n_out = self.n_a_ls[a] if g % 2 == 0 else 1
            p['out_w'] = f[self.oWo:self.obo].reshape(self.Lh, self.out_pad)[:, :n_out].copy()
            p['out_b'] = f[self.obo:self.obo + n_out].copy()
            towers.append(p)
        return towers


def coerce_config(model_config, defaults):
    """A `configparser` section (what main.py hands over: config['MODEL_CONFIG'], agents/models.py:53-69 reads it with
    getfloat / getint / get) or a plain dict -> dict typed like `defaults`.  Keys are case-insensitive like the
    reference's getters (LR_MIN / ENTROPY_COEF_MIN); unknown keys are kept as given."""
    cfg = dict(defaults)
    for k, v in dict(model_config or {}).items():
        k = str(k).lower()
        if k in defaults and isinstance(v, str):
            d = defaults[k]
            v = v.strip()
            v = int(float(v)) if isinstance(d, int) and not isinstance(d, bool) else float(v) if isinstance(d, float) else v
        cfg[k] = v
    return cfg


CKPT_FORMAT = 2      # checkpoint npz `format` entry: 2 = counters hold the BASE seed (absent: rounds 1-3, the same meaning in practice -- resume_sample_seed)


def replica_sample_seed(seed, rank=0, replica=0):
    """Action-sampling stream of one replica (torch.distributed rank x half-batch index).  Replicas must share the
    weight-init seed, so the sampling seed is derived from (seed, rank, replica): identical parameters, independent
    exploration (the uniform of instance idx at step s is U(sample_seed, s, idx), include/tsc.h tsc_model_sample)."""
    x = (int(seed) * 0x9E3779B97F4A7C15 + (int(rank) + 1) * 0xBF58476D1CE4E5B9 + (int(replica) + 1) * 0x94D049BB133111EB) & ((1 << 64) - 1)
    x ^= x >> 31
    return x & ((1 << 63) - 1) if (rank or replica) else int(seed)


def resume_sample_seed(stored_seed, fmt, rank=0, replica=0):
    """(base_seed, sample_seed) from the seed slot of a checkpoint's `counters`.  Format 2 stores the BASE seed and every
    rank / replica re-derives its own stream.  Files without a `format` entry are read the same way: round 3
    already stored the base seed, without the marker, and the files of rounds 1 - 2 were written by rank 0 / replica 0, whose
    stream seed IS the base seed (replica_sample_seed returns it unchanged there) -- taking the stored value as an
    already-derived stream would put every rank and replica that resumes from such a file on rank 0's action stream."""
    del fmt                                                        # every format written so far holds the base seed in this slot
    return int(stored_seed), replica_sample_seed(int(stored_seed), rank, replica)


def allreduce_grads_(flat_grad, group=None):
    """The one collective of the path: sum the flat gradient buffer over ranks (RCCL
    on GPUs, gloo in the CPU test) and return the 1/world factor apply_grads folds in BEFORE the
    per-agent clip, so every replica applies the identical update."""
    dist = torch.distributed
    if not (dist.is_available() and dist.is_initialized()):
        return 1.0
    world = dist.get_world_size(group)
    dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)     # also with one rank: same code path everywhere
    return 1.0 / world


def shard_seeds(seed0, n_env, rank):
    """Global env index -> seed: rank r owns instances [r*E, (r+1)*E)."""
    return [seed0 + rank * n_env + e for e in range(n_env)]


def _setup_lib(L):
    if getattr(L, '_model_ready', False):
        return
    vp = C.c_void_p
    L.tsc_model_create.argtypes = [C.POINTER(TscModelCfg), C.c_int32, C.c_int32, C.POINTER(vp)]
    L.tsc_model_destroy.argtypes = [vp]
    L.tsc_model_set_stream.argtypes = [vp, vp]
    L.tsc_model_layout.argtypes = [vp, C.POINTER(C.c_int64)]
    for f in ('tsc_model_set_params', 'tsc_model_get_params', 'tsc_model_get_opt_state', 'tsc_model_set_opt_state'):
        getattr(L, f).argtypes = [vp, vp]
    L.tsc_model_reset_opt_state.argtypes = [vp]
    L.tsc_model_debug_read.argtypes = [vp, C.c_int32, C.c_int32, C.c_int64, C.c_int64, vp]
    L.tsc_model_reset.argtypes = [vp]
    L.tsc_model_forward.argtypes = [vp, vp, vp, vp, vp, C.c_int32]
    L.tsc_model_sample.argtypes = [vp, vp, vp, C.c_uint64, C.c_uint64]
    L.tsc_model_forward_sample.argtypes = [vp, vp, vp, vp, vp, vp, C.c_uint64, C.c_uint64, C.c_int32]
    L.tsc_model_add_transition.argtypes = [vp, C.c_int32, vp, vp, vp, vp, vp, vp]
    L.tsc_model_rollout_slot.argtypes = [vp, C.c_int32, C.POINTER(vp)]
    L.tsc_model_compute_grads.argtypes = [vp, vp, C.c_double]
    L.tsc_model_grad_buffer.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_int64)]
    L.tsc_model_apply_grads.argtypes = [vp, C.c_double, C.c_double, vp]
    L.tsc_model_get_returns.argtypes = [vp, vp, vp]
    L.tsc_model_debug_clock.argtypes = [vp, C.c_int32, vp, C.c_int32]
    L.tsc_gemm_grouped_f32.argtypes = [C.c_int32] * 6 + [vp, C.c_int64, C.c_int32, vp, C.c_int64, C.c_int32,
                                                        vp, C.c_int64, C.c_int32, vp, vp, vp, vp, vp, C.c_int64, vp, C.c_int64, vp]
    L._model_ready = True


class VecA2C:
    """IA2C / MA2C for A agents x E env instances on one GPU."""

    def __init__(self, n_s_ls, n_a_ls, n_w_ls, n_f_ls, n_env, s_max, a_max, model_config=None,
                 total_step=0, device=0, seed=None, name='ma2c', process_group=None, policy='lstm', replica=0):
        if not torch.cuda.is_available():
            raise RuntimeError('VecA2C needs a GPU (MI355X); there is no CPU fallback')
        cfg = coerce_config(model_config, A2C_DEFAULTS)
        self.cfg, self.name = cfg, name
        self.policy = policy                    # 'lstm' (what the reference instantiates) or 'fc' (FcACPolicy)
        if policy == 'fc' and name == 'ma2c':
            raise ValueError('FcACPolicy has no working fingerprint variant in the reference (policies.py:259-282)')
        self.n_agent, self.E = len(n_s_ls), int(n_env)
        self.n_s_ls, self.n_a_ls, self.n_w_ls, self.n_f_ls = map(list, (n_s_ls, n_a_ls, n_w_ls, n_f_ls))
        if name != 'ma2c':
            self.n_f_ls = [0] * self.n_agent
        self.n_wave_ls = [s - w - f for s, w, f in zip(self.n_s_ls, self.n_w_ls, self.n_f_ls)]
        self.n_step = int(cfg['batch_size'])
        self.s_max, self.a_max = int(s_max), int(a_max)
        self.device = torch.device('cuda', device) if not isinstance(device, torch.device) else device
        self.pg = process_group
        self.total_step = total_step
        self._init_scheduler()
        L = _lib.lib()
        _setup_lib(L)
        self._L = L
        ip = C.POINTER(C.c_int32)
        self._arrs = [np.ascontiguousarray(x, np.int32) for x in (self.n_wave_ls, self.n_w_ls, self.n_f_ls, self.n_a_ls)]
        n_fp = int(cfg['num_fp']) if name == 'ma2c' else 0
        n_ft = int(cfg['num_ft']) if max(self.n_w_ls) > 0 else 0
        mc = TscModelCfg(self.n_agent, self.s_max, self.a_max, *[a.ctypes.data_as(ip) for a in self._arrs],
                         int(cfg['num_fw']), n_ft, n_fp, int(cfg['num_lstm']), self.n_step, float(cfg['gamma']),
                         float(cfg['reward_norm']), float(cfg['reward_clip']), float(cfg['value_coef']),
                         float(cfg['max_grad_norm']), float(cfg['rmsp_alpha']), float(cfg['rmsp_epsilon']),
                         1 if policy == 'fc' else 0)
        self.n_fc = (int(cfg['num_fw']), n_fp, n_ft)
        h = C.c_void_p()
        _lib.check(L.tsc_model_create(C.byref(mc), self.E, self.device.index or 0, C.byref(h)))
        self._h = h
        lay = (C.c_int64 * 12)()
        _lib.check(L.tsc_model_layout(h, lay))
        (self.G, self.stride, self.H, self.Lh, self.oW1, self.ob1, self.oWx, self.oWh, self.obl, self.oWo,
         self.obo, self.out_pad) = [int(x) for x in lay]
        self.layout = ParamLayout(self.n_wave_ls, self.n_w_ls, self.n_f_ls, self.n_a_ls, self.s_max, self.n_fc,
                                  self.Lh, self.out_pad, policy)
        assert self.layout.as_tuple() == tuple(int(x) for x in lay), 'host / device parameter layouts disagree'
        with torch.cuda.device(self.device):
            self.stream = torch.cuda.current_stream(self.device)
            _lib.check(L.tsc_model_set_stream(h, C.c_void_p(self.stream.cuda_stream)))
            d = self.device
            self.pi = torch.zeros(self.E, self.n_agent, self.a_max, dtype=torch.float32, device=d)
            self.v = torch.zeros(self.E, self.n_agent, dtype=torch.float32, device=d)
            self.v_boot = torch.zeros(self.E, self.n_agent, dtype=torch.float32, device=d)
            self.action = torch.zeros(self.E, self.n_agent, dtype=torch.int32, device=d)
            self._false = torch.zeros(self.E, dtype=torch.uint8, device=d)
        gp, cnt = C.c_void_p(), C.c_int64()
        _lib.check(L.tsc_model_grad_buffer(h, C.byref(gp), C.byref(cnt)))
        self._grad_ptr, self.n_param = gp.value, int(cnt.value)
        self.cur_t = 0
        self.sample_step = 0
        dist = torch.distributed
        self.rank = dist.get_rank(process_group) if dist.is_available() and dist.is_initialized() else 0
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.base_seed, self.replica = (0 if seed is None else int(seed)), int(replica)
        self.sample_seed = replica_sample_seed(self.base_seed, self.rank, replica)
        self.init_params(seed)
        self.sync_replicas()

    # ---- parameters -----------------------------------------------------------------------
    def _init_scheduler(self):
        """agents/models.py:53-69."""
        c = self.cfg
        if c['lr_decay'] == 'constant':
            self.lr_scheduler = Scheduler(c['lr_init'], decay='constant')
        else:
            self.lr_scheduler = Scheduler(c['lr_init'], c['lr_min'], self.total_step, decay=c['lr_decay'])
        if c['entropy_decay'] == 'constant':
            self.beta_scheduler = Scheduler(c['entropy_coef_init'], decay='constant')
        else:
            self.beta_scheduler = Scheduler(c['entropy_coef_init'], c['entropy_coef_min'],
                                            self.total_step * c['entropy_ratio'], decay=c['entropy_decay'])

    def agent_param_shapes(self, a):
        """TF-variable-style shapes of one tower of agent a (agents/policies.py:99-118,191-211)."""
        return tower_param_shapes(self.n_wave_ls[a], self.n_w_ls[a], self.n_f_ls[a], self.n_fc, self.Lh, self.policy)

    def init_params(self, seed=None):
        rng = np.random.RandomState(seed) if seed is not None else np.random
        self.set_tower_params(init_tower_params(self.n_wave_ls, self.n_w_ls, self.n_f_ls, self.n_a_ls, self.n_fc, self.Lh,
                                                self.policy, rng))
        _lib.check(self._L.tsc_model_reset_opt_state(self._h))      # TF1 RMSProp slot init: ms = 1

    def sync_replicas(self, src=0):
        """Data-parallel replicas must start identical (only gradients are exchanged afterwards): broadcast the flat
        parameters and the RMSProp accumulator from rank `src`.  No-op without an initialised process group."""
        if self.world <= 1:
            return
        # The communicator's FIRST collective (lazy RCCL init: bootstrap, xGMI topology, channel set-up) happens here, fenced
        # by device synchronisations, on the very buffer and stream every update uses afterwards (the gradient buffer is
        # still all zeros: the sum leaves it unchanged) -- so the first all-reduce inside backward() is an ordinary one.
        torch.cuda.synchronize(self.device)
        with torch.cuda.stream(self.stream):
            torch.distributed.all_reduce(self.grad_tensor(), op=torch.distributed.ReduceOp.SUM, group=self.pg)
        torch.cuda.synchronize(self.device)
        for what, setter in (('params', self._L.tsc_model_set_params), ('ms', self._L.tsc_model_set_opt_state)):
            t = torch.from_numpy(self.get_flat(what)).to(self.device)
            torch.distributed.broadcast(t, src=src, group=self.pg)
            flat = np.ascontiguousarray(t.cpu().numpy())
            _lib.check(setter(self._h, flat.ctypes.data_as(C.c_void_p)))

    def copy_from(self, other):
        """Make this handle a replica of `other` (same layout): parameters + optimizer state."""
        assert self.layout.as_tuple() == other.layout.as_tuple()
        for what, setter in (('params', self._L.tsc_model_set_params), ('ms', self._L.tsc_model_set_opt_state)):
            flat = np.ascontiguousarray(other.get_flat(what))
            _lib.check(setter(self._h, flat.ctypes.data_as(C.c_void_p)))

    def pack(self, towers):
        return self.layout.pack(towers)

    def unpack(self, flat):
        return self.layout.unpack(flat)

    def set_tower_params(self, towers):
        flat = np.ascontiguousarray(self.pack(towers))
        _lib.check(self._L.tsc_model_set_params(self._h, flat.ctypes.data_as(C.c_void_p)))

    def get_flat(self, what='params'):
        out = np.zeros(self.n_param, np.float32)
        fn = self._L.tsc_model_get_params if what == 'params' else self._L.tsc_model_get_opt_state
        _lib.check(fn(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    def get_tower_params(self):
        return self.unpack(self.get_flat())

    def grad_tensor(self):
        """The contiguous fp32 gradient buffer as a torch tensor view (for RCCL all-reduce)."""
        if getattr(self, '_grad_t', None) is not None:
            return self._grad_t
        class _Holder:
            pass
        hold = _Holder()
        hold.__cuda_array_interface__ = {'shape': (self.n_param,), 'typestr': '<f4', 'data': (self._grad_ptr, False),
                                         'version': 3, 'strides': None}
        self._grad_t = torch.as_tensor(hold, device=self.device)
        return self._grad_t

    def _view(self, ptr, shape, typestr, dtype):
        class _Holder:
            pass
        hold = _Holder()
        hold.__cuda_array_interface__ = {'shape': tuple(shape), 'typestr': typestr, 'data': (ptr, False), 'version': 3, 'strides': None}
        t = torch.as_tensor(hold, device=self.device)
        assert t.dtype == dtype and t.data_ptr() == ptr
        return t

    def rollout_slots(self):
        """Zero-copy rollouts (include/tsc.h tsc_model_rollout_slot): torch views of the on-policy buffer itself --
        obs [T+1,E,A,SMAX] f32, action [T,E,A] i32, value [T,E,A] f32, reward [T,E,A] f64 (raw), done [T+1,E] u8
        (done[t] = before step t, done[t+1] = after it).  The trainer lets the forward and the env write them in place."""
        if getattr(self, '_slots', None) is None:
            p = (C.c_void_p * 6)()
            _lib.check(self._L.tsc_model_rollout_slot(self._h, 0, p))
            T, E, A = self.n_step, self.E, self.n_agent
            self._slots = dict(obs=self._view(p[0], (T + 1, E, A, self.s_max), '<f4', torch.float32),
                               action=self._view(p[1], (T, E, A), '<i4', torch.int32),
                               value=self._view(p[2], (T, E, A), '<f4', torch.float32),
                               reward=self._view(p[3], (T, E, A), '<f8', torch.float64),
                               done=self._view(p[4], (T + 1, E), '|u1', torch.uint8))
        return self._slots

    # ---- reference API (batched) ------------------------------------------------------------
    def reset(self):
        """agents/models.py:218-220."""
        _lib.check(self._L.tsc_model_reset(self._h))

    def forward(self, obs, done, out_type='pv'):
        """agents/models.py:185-200.  obs f32 [E,A,SMAX] (device), done u8 [E] or bool.
        Returns tensors pi [E,A,AMAX], v [E,A] (or one of them for 'p' / 'v')."""
        if not torch.is_tensor(done):
            done = torch.full((self.E,), int(bool(done)), dtype=torch.uint8, device=self.device)
        adv = 1 if 'p' in out_type else 0
        v_out = self.v if adv else self.v_boot
        _lib.check(self._L.tsc_model_forward(self._h, C.c_void_p(obs.data_ptr()), C.c_void_p(done.data_ptr()),
                                             C.c_void_p(self.pi.data_ptr()), C.c_void_p(v_out.data_ptr()), adv))
        if out_type == 'pv':
            return self.pi, self.v
        return self.pi if out_type == 'p' else v_out

    def forward_sample(self, obs, done, cache=True, v_out=None, action_out=None):
        """forward(obs, done, 'pv') + sample() in one launch -> (pi, v, action).  cache=True keeps this step's
        activations (slot = the transition add_transition fills next) so backward() skips the re-forward.
        v_out / action_out: write the value and the action somewhere else (the rollout slot, rollout_slots())."""
        if not torch.is_tensor(done):
            done = torch.full((self.E,), int(bool(done)), dtype=torch.uint8, device=self.device)
        v = self.v if v_out is None else v_out
        act = self.action if action_out is None else action_out
        _lib.check(self._L.tsc_model_forward_sample(
            self._h, C.c_void_p(obs.data_ptr()), C.c_void_p(done.data_ptr()), C.c_void_p(self.pi.data_ptr()),
            C.c_void_p(v.data_ptr()), C.c_void_p(act.data_ptr()), self.sample_seed, self.sample_step,
            self.cur_t if cache else -1))
        self.sample_step += 1
        return self.pi, v, act

    def commit_transition(self):
        """The transition of slot cur_t was written in place (rollout_slots): only the slot counter moves."""
        self.cur_t += 1

    def sample(self, pi=None):
        """np.random.choice per agent (utils.py:155-157), counter-based RNG."""
        pi = self.pi if pi is None else pi
        _lib.check(self._L.tsc_model_sample(self._h, C.c_void_p(pi.data_ptr()), C.c_void_p(self.action.data_ptr()),
                                            self.sample_seed, self.sample_step))
        self.sample_step += 1
        return self.action

    def add_transition(self, obs, done_pre, actions, rewards, values, done_post):
        """agents/models.py:222-229 (+ the pre-step done the LSTM saw, agents/utils.py:225-226)."""
        if not torch.is_tensor(done_pre):
            done_pre = torch.full((self.E,), int(bool(done_pre)), dtype=torch.uint8, device=self.device)
        _lib.check(self._L.tsc_model_add_transition(
            self._h, self.cur_t, C.c_void_p(obs.data_ptr()), C.c_void_p(done_pre.data_ptr()),
            C.c_void_p(actions.data_ptr()), C.c_void_p(rewards.data_ptr()), C.c_void_p(values.data_ptr()),
            C.c_void_p(done_post.data_ptr())))
        self.cur_t += 1

    def use_stream(self, stream):
        self.stream = stream
        _lib.check(self._L.tsc_model_set_stream(self._h, C.c_void_p(stream.cuda_stream)))

    def compute_grads(self, R):
        """First half of IA2C.backward (agents/models.py:174-183): returns + loss + BPTT -> flat gradient buffer."""
        assert self.cur_t == self.n_step, 'backward() needs a full n_step buffer (T %% n_step == 0, utils.py:121)'
        self._cur_lr = self.lr_scheduler.get(self.n_step)
        self._cur_beta = cur_beta = self.beta_scheduler.get(self.n_step)
        _lib.check(self._L.tsc_model_compute_grads(self._h, C.c_void_p(R.data_ptr()), float(cur_beta)))

    def apply_grads(self, scale=1.0, want_stats=False):
        """Second half: per-agent clip on grad * scale, TF1 RMSProp, states_bw <- states_fw, buffer reset."""
        stats = np.zeros((self.n_agent, 4), np.float64) if want_stats else None
        _lib.check(self._L.tsc_model_apply_grads(self._h, float(self._cur_lr), float(scale),
                                                 stats.ctypes.data_as(C.c_void_p) if want_stats else None))
        self.cur_t = 0
        return stats

    def backward(self, R, summary_writer=None, global_step=None, want_stats=False):
        """agents/models.py:174-183.  R: bootstrap values f32 [E,A] (zeros where terminal)."""
        self.compute_grads(R)
        scale = 1.0
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            # the gradient kernels run on self.stream: issue the collective there too, so that it starts after
            # compute_grads and apply_grads starts after it, whatever torch's current stream is
            with torch.cuda.stream(self.stream):
                scale = allreduce_grads_(self.grad_tensor(), self.pg)   # RCCL over xGMI, one flat buffer
        return self.apply_grads(scale, want_stats)

    # ---- checkpoints (agents/models.py:83-108: `checkpoint-<step>`, highest step wins) -------
    def save(self, model_dir, global_step):
        os.makedirs(model_dir, exist_ok=True)
        np.savez(os.path.join(model_dir, 'checkpoint-%d.npz' % int(global_step)), params=self.get_flat('params'),
                 ms=self.get_flat('ms'), layout=np.array(self.layout.as_tuple() + (self.s_max,), np.int64),
                 dims=np.array([self.n_wave_ls, self.n_w_ls, self.n_f_ls, self.n_a_ls], np.int64),
                 counters=np.array([self.sample_step, self.base_seed, self.lr_scheduler.n, self.beta_scheduler.n], np.int64),
                 format=np.int64(CKPT_FORMAT))

    def load(self, model_dir, checkpoint=None):
        save_file, save_step = None, 0
        if os.path.exists(model_dir):
            if checkpoint is None:
                for f in os.listdir(model_dir):
                    if f.startswith('checkpoint'):
                        tokens = f.split('.')[0].split('-')
                        if len(tokens) != 2:
                            continue
                        if int(tokens[1]) > save_step:
                            save_file, save_step = f, int(tokens[1])
            else:
                save_file = 'checkpoint-%d.npz' % int(checkpoint)
        if save_file is None or not os.path.exists(os.path.join(model_dir, save_file)):
            return False
        z = np.load(os.path.join(model_dir, save_file))
        p = np.ascontiguousarray(z['params'], np.float32)
        ms = np.ascontiguousarray(z['ms'], np.float32)
        want = self.layout.as_tuple() + (self.s_max,)
        dims = np.array([self.n_wave_ls, self.n_w_ls, self.n_f_ls, self.n_a_ls], np.int64)
        if tuple(int(x) for x in z['layout']) != want or p.size != self.n_param or ms.size != self.n_param or \
                ('dims' in z.files and not np.array_equal(z['dims'], dims)):
            raise ValueError('checkpoint %s does not fit this model (layout %r vs %r, %d vs %d parameters)'
                             % (save_file, tuple(int(x) for x in z['layout']), want, p.size, self.n_param))
        _lib.check(self._L.tsc_model_set_params(self._h, p.ctypes.data_as(C.c_void_p)))
        _lib.check(self._L.tsc_model_set_opt_state(self._h, ms.ctypes.data_as(C.c_void_p)))
        if 'counters' in z.files:            # action-RNG stream and lr / beta schedules resume where they stopped; the checkpoint holds
            # the BASE seed, every rank / replica re-derives its own stream from it (independent exploration survives a resume)
            self.sample_step, seed = int(z['counters'][0]), int(z['counters'][1])
            base, self.sample_seed = resume_sample_seed(seed, z['format'] if 'format' in z.files else None, self.rank, self.replica)
            self.base_seed = self.base_seed if base is None else base
            self.lr_scheduler.n, self.beta_scheduler.n = int(z['counters'][2]), int(z['counters'][3])
        return True

    def close(self):
        if getattr(self, '_h', None):
            self._L.tsc_model_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class IA2C:
    """E = 1 duck-type of agents/models.py:132-229 (lists of per-agent arrays in and out)."""
    _name = 'ia2c'

    def __init__(self, n_s_ls, n_a_ls, n_w_ls, total_step, model_config, seed=0, device=0, n_f_ls=None):
        n_f_ls = [0] * len(n_s_ls) if n_f_ls is None else n_f_ls
        s_max = (max(n_s_ls) + 3) // 4 * 4
        self.vec = VecA2C(n_s_ls, n_a_ls, n_w_ls, n_f_ls, 1, s_max, max(n_a_ls), model_config, total_step,
                          device=device, seed=seed, name=self._name)
        self.name, self.n_agent, self.n_step = self._name, len(n_s_ls), self.vec.n_step
        self.n_s_ls, self.n_a_ls = list(n_s_ls), list(n_a_ls)
        self.sess = None
        self._last_done = False
        d = self.vec.device
        self._obs = torch.zeros(1, self.n_agent, s_max, dtype=torch.float32, device=d)

    def _put_obs(self, obs):
        o = np.zeros((1, self.n_agent, self.vec.s_max), np.float32)
        for a, ob in enumerate(obs):
            o[0, a, :len(ob)] = np.asarray(ob, np.float32)          # TF feed casts to float32
        self._obs.copy_(torch.from_numpy(o))
        return self._obs

    def forward(self, obs, done, out_type='pv'):
        out = self.vec.forward(self._put_obs(obs), done, out_type)
        self._last_done = done
        if out_type == 'pv':
            pi, v = out[0][0].cpu().numpy(), out[1][0].cpu().numpy()
            return [pi[a, :n].copy() for a, n in enumerate(self.n_a_ls)], [v[a] for a in range(self.n_agent)]
        if out_type == 'p':
            pi = out[0].cpu().numpy()
            return [pi[a, :n].copy() for a, n in enumerate(self.n_a_ls)]
        v = out[0].cpu().numpy()
        return [v[a] for a in range(self.n_agent)]

    def add_transition(self, obs, actions, rewards, values, done):
        d = self.vec.device
        self.vec.add_transition(self._put_obs(obs), self._last_done,
                                torch.tensor([list(map(int, actions))], dtype=torch.int32, device=d),
                                torch.tensor(np.asarray(rewards, np.float64).reshape(1, -1), device=d),
                                torch.tensor(np.asarray(values, np.float32).reshape(1, -1), device=d),
                                torch.tensor([int(bool(done))], dtype=torch.uint8, device=d))

    def backward(self, R_ls, summary_writer=None, global_step=None):
        R = torch.tensor(np.asarray(R_ls, np.float32).reshape(1, -1), device=self.vec.device)
        return self.vec.backward(R, summary_writer, global_step)

    def reset(self):
        self.vec.reset()

    def save(self, model_dir, global_step):
        self.vec.save(model_dir, global_step)

    def load(self, model_dir, checkpoint=None):
        return self.vec.load(model_dir, checkpoint)


class MA2C(IA2C):
    """agents/models.py:232-261."""
    _name = 'ma2c'

    def __init__(self, n_s_ls, n_a_ls, n_w_ls, n_f_ls, total_step, model_config, seed=0, device=0):
        super().__init__(n_s_ls, n_a_ls, n_w_ls, total_step, model_config, seed=seed, device=device, n_f_ls=n_f_ls)
