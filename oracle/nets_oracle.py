"""CPU restatement of the reference learner -- TEST INFRASTRUCTURE ONLY (oracle/__init__.py).

Follows, in float64 torch (autograd plays the role of tf.gradients):
  fc, lstm .................. agents/utils.py:66-74, 88-116   (gate order i,f,o,u; done-masked state)
  LstmACPolicy / FPLstmAC ... agents/policies.py:99-118, 191-211 (separate pi / v towers;
                              h = [fcw(wave), fcf(fingerprint), fct(wait)])
  heads ..................... agents/policies.py:20-26
  loss ...................... agents/policies.py:41-52
  clip + RMSProp ............ agents/policies.py:54-61 (tf.clip_by_global_norm, TF1 RMSPropOptimizer:
                              ms init 1, ms = a ms + (1-a) g^2, w -= lr g / sqrt(ms + eps))
  returns / advantages ...... agents/utils.py:202-228 (pinned against the reference's own
                              OnPolicyBuffer by tests/golden/learner_known_answers.npz)
TensorFlow 1.12 itself is absent: TF numerics are restated from the graph code and
TF's documented op semantics -- "parity unpinned" for the TF kernels, stated in DESIGN.md.
The env dimension E is the batch dimension; the loss averages over the T*E samples of an agent.
"""
import numpy as np
import torch

DT = torch.float64


def t64(x):
    return torch.as_tensor(np.asarray(x), dtype=DT)


def fc(x, w, b, act=torch.relu):
    z = x @ w + b
    return act(z) if act is not None else z


def lstm(xs, dones, s, wx, wh, b):
    """xs [T,B,n], dones [T,B] (pre-step), s [B,2L] = [c,h] -> (hs [T,B,L], s_new)."""
    L = s.shape[1] // 2
    c, h = s[:, :L], s[:, L:]
    out = []
    for t in range(xs.shape[0]):
        keep = (1.0 - dones[t]).unsqueeze(1)
        c = c * keep
        h = h * keep
        z = xs[t] @ wx + h @ wh + b
        i, f, o, u = z[:, :L], z[:, L:2 * L], z[:, 2 * L:3 * L], z[:, 3 * L:]
        i, f, o, u = torch.sigmoid(i), torch.sigmoid(f), torch.sigmoid(o), torch.tanh(u)
        c = f * c + i * u
        h = o * torch.tanh(c)
        out.append(h)
    return torch.stack(out), torch.cat([c, h], 1)


def tower(p, ob, dones, s, nw, nt, nf):
    """One tower of one agent.  ob [T,B,n_s] in env order [wave | wait | fingerprint]."""
    hs = [fc(ob[..., :nw], p['fcw_w'], p['fcw_b'])]
    if nf:
        hs.append(fc(ob[..., nw + nt:nw + nt + nf], p['fcf_w'], p['fcf_b']))
    if nt:
        hs.append(fc(ob[..., nw:nw + nt], p['fct_w'], p['fct_b']))
    h = torch.cat(hs, -1)
    if 'fc_w' in p:                                   # FcACPolicy._build_net (agents/policies.py:227-235)
        h, s_new = fc(h, p['fc_w'], p['fc_b']), s
    else:
        h, s_new = lstm(h, dones, s, p['lstm_wx'], p['lstm_wh'], p['lstm_b'])
    out = h @ p['out_w'] + p['out_b']
    return out, s_new


def tower_activations(p, ob, done, s, nw, nt, nf):
    """One LSTM step of one tower with every intermediate the HIP rollout forward caches for the update:
    ob [B,n_s], done [B], s [B,2L] -> dict(X1 [B,H], gates [B,4L] post-activation i|f|o|u, c, h, hprev [B,L]).
    Same arithmetic as tower() / lstm() above (agents/policies.py:99-118,191-211; agents/utils.py:88-116)."""
    hs = [fc(ob[..., :nw], p['fcw_w'], p['fcw_b'])]
    if nf:
        hs.append(fc(ob[..., nw + nt:nw + nt + nf], p['fcf_w'], p['fcf_b']))
    if nt:
        hs.append(fc(ob[..., nw:nw + nt], p['fct_w'], p['fct_b']))
    x1 = torch.cat(hs, -1)
    L = s.shape[1] // 2
    keep = (1.0 - done).unsqueeze(1)
    c, h = s[:, :L] * keep, s[:, L:] * keep
    z = x1 @ p['lstm_wx'] + h @ p['lstm_wh'] + p['lstm_b']
    i, f, o, u = (torch.sigmoid(z[:, :L]), torch.sigmoid(z[:, L:2 * L]), torch.sigmoid(z[:, 2 * L:3 * L]),
                  torch.tanh(z[:, 3 * L:]))
    cn = f * c + i * u
    hn = o * torch.tanh(cn)
    return dict(X1=x1, gates=torch.cat([i, f, o, u], 1), c=cn, h=hn, hprev=h)


def to_torch(tower_params, requires_grad=False):
    out = []
    for p in tower_params:
        q = {k: t64(v).clone().requires_grad_(requires_grad) for k, v in p.items()}
        out.append(q)
    return out


class OracleA2C:
    """IA2C / MA2C over E env instances in float64."""

    def __init__(self, tower_params, n_wave_ls, n_w_ls, n_f_ls, n_a_ls, n_env, n_lstm=64, gamma=0.99,
                 reward_norm=2000.0, reward_clip=2.0, value_coef=0.5, max_grad_norm=40.0, alpha=0.99, eps=1e-5,
                 state_f32=False):
        # state_f32: round the carried LSTM state to float32 after every forward -- the reference fetches new_states from
        # the session (float32) and feeds it back (agents/policies.py:130-135); only the Oracle-B replay
        # (tests/test_refnet_oracle.py) needs that to agree with the recorded float64 values to 1e-12
        self.state_f32 = state_f32
        self.p = to_torch(tower_params)
        self.ms = [{k: torch.ones_like(v) for k, v in p.items()} for p in self.p]
        self.nw, self.nt, self.nf, self.na = n_wave_ls, n_w_ls, n_f_ls, n_a_ls
        self.A, self.E, self.L = len(n_a_ls), n_env, n_lstm
        self.gamma, self.rnorm, self.rclip = gamma, reward_norm, reward_clip
        self.vcoef, self.max_norm, self.alpha, self.eps = value_coef, max_grad_norm, alpha, eps
        self.reset()
        self.buf = dict(obs=[], acts=[], rs=[], vs=[], dones=[None])

    def reset(self):                                          # policies.py:120-123
        self.s_fw = [torch.zeros(self.E, 2 * self.L, dtype=DT) for _ in range(2 * self.A)]
        self.s_bw = [s.clone() for s in self.s_fw]

    def _ob(self, obs, a):
        n = self.nw[a] + self.nt[a] + self.nf[a]
        return t64(obs[..., a, :n])

    def forward(self, obs, done, out_type='pv'):
        """obs [E,A,SMAX] ndarray, done [E] -> pi list[A] of [E,n_a], v [E,A]."""
        d = t64(np.broadcast_to(np.asarray(done, np.float64), (self.E,))).unsqueeze(0)
        pis, vs = [], []
        with torch.no_grad():
            for a in range(self.A):
                ob = self._ob(obs, a).unsqueeze(0)
                lo, s0 = tower(self.p[2 * a], ob, d, self.s_fw[2 * a], self.nw[a], self.nt[a], self.nf[a])
                vo, s1 = tower(self.p[2 * a + 1], ob, d, self.s_fw[2 * a + 1], self.nw[a], self.nt[a], self.nf[a])
                if 'p' in out_type:                            # policies.py:127-135
                    if self.state_f32:
                        s0, s1 = s0.float().double(), s1.float().double()
                    self.s_fw[2 * a], self.s_fw[2 * a + 1] = s0, s1
                pis.append(torch.softmax(lo[0], -1).numpy())
                vs.append(vo[0, :, 0].numpy())
        return pis, np.stack(vs, 1)

    def add_transition(self, obs, done_pre, actions, rewards, values, done_post):
        r = np.asarray(rewards, np.float64)
        if self.rnorm:
            r = r / self.rnorm                                   # models.py:223-224
        if self.rclip:
            r = np.clip(r, -self.rclip, self.rclip)              # :225-226
        b = self.buf
        if b['dones'][0] is None or len(b['obs']) == 0:
            b['dones'][0] = np.asarray(done_pre, np.float64).copy()
        b['obs'].append(np.array(obs, np.float64)); b['acts'].append(np.array(actions))
        b['rs'].append(r); b['vs'].append(np.asarray(values, np.float32).astype(np.float64))
        b['dones'].append(np.asarray(done_post, np.float64).copy())

    @staticmethod
    def returns_advs(rs, vs, dones, R, gamma):
        """agents/utils.py:202-214: rs, vs [T,...]; dones [T+1,...] (index t+1 = post-step done)."""
        T = len(rs)
        Rs, Advs = [None] * T, [None] * T
        R = np.asarray(R, np.float64)
        for t in range(T - 1, -1, -1):
            R = rs[t] + gamma * R * (1. - dones[t + 1])
            Rs[t], Advs[t] = R, R - vs[t]
        return (np.array(Rs, np.float64).astype(np.float32), np.array(Advs, np.float64).astype(np.float32))

    def compute_grads(self, R_boot, beta):
        """-> (grads list[2A] of dicts, stats [A,3]) ; float64 autograd of policies.py:41-52."""
        b = self.buf
        T = len(b['obs'])
        obs = np.stack(b['obs'])                                 # [T,E,A,S]
        dones = np.stack(b['dones'])                             # [T+1,E]
        dpost = dones[:, :, None] * np.ones((1, 1, self.A))
        Rs, Advs = self.returns_advs(np.stack(b['rs']), np.stack(b['vs']), dpost, R_boot, self.gamma)
        self.Rs, self.Advs = Rs, Advs
        acts = np.stack(b['acts'])
        dpre = t64(dones[:-1])
        grads, stats = [], []
        P = to_torch([{k: v.numpy() for k, v in p.items()} for p in self.p], requires_grad=True)
        # ReLU kinks: hidden units with a pre-activation within 1e-5 of zero for some sample.  There a float32
        # evaluation may land on the other side of the kink than float64 and the unit's weight-gradient column then
        # differs by that sample's whole contribution -- a property of relu, not an error of either side.  Parity
        # tests compare those columns with a loose bound (self.kink_cols[tower][layer] = bool per hidden unit).
        self.kink_cols = []
        with torch.no_grad():
            for g_, p in enumerate(self.p):
                a = g_ // 2
                ob = self._ob(obs, a)
                nw, nt, nf = self.nw[a], self.nt[a], self.nf[a]
                kc = {'fcw': ((ob[..., :nw] @ p['fcw_w'] + p['fcw_b']).abs() < 1e-5).reshape(-1, p['fcw_b'].shape[0]).any(0).numpy()}
                hs = [fc(ob[..., :nw], p['fcw_w'], p['fcw_b'])]
                if nf:
                    kc['fcf'] = ((ob[..., nw + nt:nw + nt + nf] @ p['fcf_w'] + p['fcf_b']).abs() < 1e-5).reshape(-1, p['fcf_b'].shape[0]).any(0).numpy()
                    hs.append(fc(ob[..., nw + nt:nw + nt + nf], p['fcf_w'], p['fcf_b']))
                if nt:
                    kc['fct'] = ((ob[..., nw:nw + nt] @ p['fct_w'] + p['fct_b']).abs() < 1e-5).reshape(-1, p['fct_b'].shape[0]).any(0).numpy()
                    hs.append(fc(ob[..., nw:nw + nt], p['fct_w'], p['fct_b']))
                if 'fc_w' in p:
                    kc['fc'] = ((torch.cat(hs, -1) @ p['fc_w'] + p['fc_b']).abs() < 1e-5).reshape(-1, p['fc_b'].shape[0]).any(0).numpy()
                self.kink_cols.append(kc)
        for a in range(self.A):
            ob = self._ob(obs, a)
            lo, _ = tower(P[2 * a], ob, dpre, self.s_bw[2 * a], self.nw[a], self.nt[a], self.nf[a])
            vo, _ = tower(P[2 * a + 1], ob, dpre, self.s_bw[2 * a + 1], self.nw[a], self.nt[a], self.nf[a])
            pi = torch.softmax(lo, -1).reshape(T * self.E, -1)
            v = vo.reshape(T * self.E)
            A_ = torch.as_tensor(acts[:, :, a].reshape(-1), dtype=torch.long)
            ADV, R = t64(Advs[:, :, a].reshape(-1)), t64(Rs[:, :, a].reshape(-1))
            log_pi = torch.log(torch.clamp(pi, 1e-10, 1.0))
            entropy = -(pi * log_pi).sum(1)
            entropy_loss = -entropy.mean() * beta
            policy_loss = -(log_pi.gather(1, A_[:, None])[:, 0] * ADV).mean()
            value_loss = ((R - v) ** 2).mean() * 0.5 * self.vcoef
            loss = policy_loss + value_loss + entropy_loss
            loss.backward()
            stats.append([policy_loss.item(), value_loss.item(), entropy_loss.item()])
        for q in P:
            grads.append({k: (v.grad if v.grad is not None else torch.zeros_like(v)).detach() for k, v in q.items()})
        return grads, np.array(stats)

    def apply_grads(self, grads, lr, grad_scale=1.0):
        norms = []
        for a in range(self.A):
            gl = [g * grad_scale for t in (2 * a, 2 * a + 1) for g in grads[t].values()]
            norm = torch.sqrt(sum((g ** 2).sum() for g in gl))   # tf.clip_by_global_norm
            norms.append(norm.item())
            sc = self.max_norm / max(norm.item(), self.max_norm) if self.max_norm > 0 else 1.0
            for t in (2 * a, 2 * a + 1):
                for k, g in grads[t].items():
                    g = g * grad_scale * sc
                    ms = self.alpha * self.ms[t][k] + (1 - self.alpha) * g * g
                    self.ms[t][k] = ms
                    self.p[t][k] = self.p[t][k] - lr * g / torch.sqrt(ms + self.eps)
        self.s_bw = [s.clone() for s in self.s_fw]               # policies.py:153
        last = self.buf['dones'][-1]
        self.buf = dict(obs=[], acts=[], rs=[], vs=[], dones=[last])   # utils.py:227
        return np.array(norms)

    def tower_params(self):
        return [{k: v.numpy().astype(np.float32) for k, v in p.items()} for p in self.p]

    def tower_params_f64(self):
        return [{k: v.numpy().copy() for k, v in p.items()} for p in self.p]


def choice_from_uniform(pi, u):
    """np.random.choice(n, p=pi) given its uniform draw u (numpy mtrand: cdf = cumsum(p);
    cdf /= cdf[-1]; searchsorted(cdf, u, side='right'))."""
    cdf = np.cumsum(np.asarray(pi, np.float64))
    cdf /= cdf[-1]
    return int(np.searchsorted(cdf, u, side='right'))


def splitmix64(x):
    M = (1 << 64) - 1
    x = (x + 0x9E3779B97F4A7C15) & M
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & M
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & M
    return x ^ (x >> 31)


def sample_uniform(seed, step, idx):
    """The counter-based uniform the HIP sampler documents (include/tsc.h tsc_model_sample)."""
    M = (1 << 64) - 1
    h = splitmix64((splitmix64(seed ^ ((step * 0xD1B54A32D192ED03) & M)) + idx) & M)
    return (h >> 11) * (1.0 / 9007199254740992.0)
