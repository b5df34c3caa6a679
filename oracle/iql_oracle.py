"""CPU restatement of the reference's independent Q-learning agents -- TEST INFRASTRUCTURE ONLY
(oracle/__init__.py): the checker of csrc/tsc_iql.hip (tests/test_iql_gpu.py).

Follows:
  ReplayBuffer .............. agents/utils.py:231-263 (ring of `buffer_size` tuples, random.sample minibatch;
                              pinned against the reference class by tests/golden/iql_known_answers.npz)
  LRQPolicy / DeepQPolicy ... agents/policies.py:285-389: q = fc(S -> n_a, linear)   (LR)
                              h = [relu(fc(wave -> n_fc0)), relu(fc(wait -> n_fc0/4))] -> relu(fc(. -> n_fc)) -> fc(-> n_a)
  loss ...................... agents/policies.py:305-328: mean((Q(s)[a] - stop_grad(done ? r : r + gamma max Q(s')))^2),
                              the SAME network for Q(s') (no target network), tf.clip_by_global_norm,
                              tf.train.AdamOptimizer defaults (beta1 .9, beta2 .999, eps 1e-8;
                              lr_t = lr sqrt(1 - b2^t) / (1 - b1^t); w -= lr_t m / (sqrt(v) + eps))
  IQL.forward / backward .... agents/models.py:332-363: epsilon-greedy, 10 minibatch updates per rollout once
                              the buffer holds a batch; reward norm / clip as in add_transition :369-376
TensorFlow itself is absent: "parity unpinned" for the TF kernels, as for the A2C learner.
"""
import random

import numpy as np
import torch

DT = torch.float64


class ReplayBuffer:
    """agents/utils.py:231-263."""

    def __init__(self, buffer_size, batch_size):
        self.buffer_size, self.batch_size = buffer_size, batch_size
        self.cum_size = 0
        self.buffer = []

    def add_transition(self, ob, a, r, next_ob, done):
        item = (ob, a, r, next_ob, done)
        if self.cum_size < self.buffer_size:
            self.buffer.append(item)
        else:
            self.buffer[int(self.cum_size % self.buffer_size)] = item
        self.cum_size += 1

    def reset(self):
        self.buffer, self.cum_size = [], 0

    def sample_transition(self, rng=random):
        mb = rng.sample(self.buffer, self.batch_size)
        return tuple(np.asarray([d[i] for d in mb]) for i in (0, 1, 3, 2, 4))   # obs, acts, next_obs, rs, dones

    @property
    def size(self):
        return min(self.buffer_size, self.cum_size)


def q_net(p, S, n_s, n_w):
    """p: dict of float64 tensors.  'lr': {q_w, q_b};  'dqn': {fcw_w, fcw_b, [fct_w, fct_b], fc0_w, fc0_b, q_w, q_b}."""
    if 'fcw_w' not in p:
        return S @ p['q_w'] + p['q_b']
    h = torch.relu(S[:, :n_s] @ p['fcw_w'] + p['fcw_b'])
    if n_w:
        h = torch.cat([h, torch.relu(S[:, n_s:] @ p['fct_w'] + p['fct_b'])], 1)
    h = torch.relu(h @ p['fc0_w'] + p['fc0_b'])
    return h @ p['q_w'] + p['q_b']


class OracleQ:
    """One agent's Q-learner (QPolicy.prepare_loss + AdamOptimizer), float64."""

    def __init__(self, params, n_s, n_w, gamma=0.99, max_grad_norm=40.0):
        self.p = {k: torch.as_tensor(np.asarray(v), dtype=DT).clone() for k, v in params.items()}
        self.n_s, self.n_w, self.gamma, self.max_norm = n_s, n_w, gamma, max_grad_norm
        self.m = {k: torch.zeros_like(v) for k, v in self.p.items()}
        self.v = {k: torch.zeros_like(v) for k, v in self.p.items()}
        self.t = 0

    def forward(self, ob):
        with torch.no_grad():
            return q_net(self.p, torch.as_tensor(np.asarray(ob)[None], dtype=DT), self.n_s, self.n_w)[0].numpy()

    def loss_and_grads(self, obs, acts, next_obs, dones, rs):
        P = {k: v.clone().requires_grad_(True) for k, v in self.p.items()}
        S, S1 = torch.as_tensor(np.asarray(obs), dtype=DT), torch.as_tensor(np.asarray(next_obs), dtype=DT)
        q0 = q_net(P, S, self.n_s, self.n_w).gather(1, torch.as_tensor(np.asarray(acts), dtype=torch.long)[:, None])[:, 0]
        with torch.no_grad():
            q1 = q_net(self.p, S1, self.n_s, self.n_w).max(1).values
            r = torch.as_tensor(np.asarray(rs), dtype=DT)
            d = torch.as_tensor(np.asarray(dones).astype(bool))
            tq = torch.where(d, r, r + self.gamma * q1)
        loss = ((q0 - tq) ** 2).mean()
        loss.backward()
        return loss.item(), {k: v.grad.detach() for k, v in P.items()}

    def backward(self, obs, acts, next_obs, dones, rs, lr):
        loss, g = self.loss_and_grads(obs, acts, next_obs, dones, rs)
        norm = torch.sqrt(sum((x ** 2).sum() for x in g.values())).item()
        if self.max_norm > 0:                                    # tf.clip_by_global_norm
            sc = self.max_norm / max(norm, self.max_norm)
            g = {k: x * sc for k, x in g.items()}
        self.t += 1
        b1, b2, eps = 0.9, 0.999, 1e-8
        lr_t = lr * np.sqrt(1.0 - b2 ** self.t) / (1.0 - b1 ** self.t)
        for k in self.p:
            self.m[k] = b1 * self.m[k] + (1 - b1) * g[k]
            self.v[k] = b2 * self.v[k] + (1 - b2) * g[k] * g[k]
            self.p[k] = self.p[k] - lr_t * self.m[k] / (torch.sqrt(self.v[k]) + eps)
        return loss, norm


def act_epsilon_greedy(qs, eps, u_explore, u_action):
    """IQL.forward 'explore' (agents/models.py:339-345) given its two uniform draws:
    np.random.random() < eps -> np.random.randint(n_a) (= floor(u_action * n_a)), else argmax."""
    if u_explore < eps:
        return int(u_action * len(qs))
    return int(np.argmax(qs))


def floyd_sample(size, batch, uniform):
    """`batch` distinct indices in [0, size) -- the documented replacement of random.sample (include/tsc.h
    tsc_iql_compute_grads): for i in [0, B): j = size - B + i; t = floor(uniform(i) * (j + 1)); pick t, or j if t was
    picked before."""
    out = []
    for i in range(batch):
        j = size - batch + i
        t = min(int(uniform(i) * (j + 1)), j)
        out.append(j if t in out else t)
    return out


class OracleIQL:
    """IQL over E env instances (agents/models.py:264-376): one OracleQ per agent, one ring per (instance, agent),
    minibatch = the rows every instance's ring contributes, loss = mean over them (E = 1: the reference)."""

    def __init__(self, agent_params, n_wave_ls, n_w_ls, n_a_ls, n_env, batch_size=20, buffer_size=1000, gamma=0.99,
                 reward_norm=3000.0, reward_clip=2.0, max_grad_norm=40.0, replay_seed=0):
        self.qs = [OracleQ(p, nw, nt, gamma, max_grad_norm) for p, nw, nt in zip(agent_params, n_wave_ls, n_w_ls)]
        self.nw, self.nt, self.na = list(n_wave_ls), list(n_w_ls), list(n_a_ls)
        self.A, self.E, self.B, self.cap = len(n_a_ls), n_env, batch_size, int(buffer_size)
        self.rnorm, self.rclip, self.replay_seed = reward_norm, reward_clip, replay_seed
        self.rings = [[ReplayBuffer(self.cap, batch_size) for _ in range(self.A)] for _ in range(n_env)]
        self.update_step = 0
        self.last_idx = None

    def forward(self, obs):
        """obs [E,A,SMAX] -> list[A] of q [E, n_a] (float64)."""
        out = []
        for a, q in enumerate(self.qs):
            n = self.nw[a] + self.nt[a]
            with torch.no_grad():
                out.append(q_net(q.p, torch.as_tensor(np.asarray(obs)[:, a, :n], dtype=DT), q.n_s, q.n_w).numpy())
        return out

    def add_transition(self, obs, actions, rewards, next_obs, done):
        r = np.asarray(rewards, np.float64)
        if self.rnorm:
            r = r / self.rnorm
        if self.rclip:
            r = np.clip(r, -self.rclip, self.rclip)
        for e in range(self.E):
            for a in range(self.A):
                n = self.nw[a] + self.nt[a]
                self.rings[e][a].add_transition(np.array(obs[e, a, :n], np.float64), int(actions[e, a]), float(np.float32(r[e, a])),
                                                np.array(next_obs[e, a, :n], np.float64), bool(done[e]))

    def minibatch_step(self, lr, idx_given=None):
        """-> (per-agent loss, per-agent grad norm, grads list[A] of dict) ; also applies the Adam step.
        idx_given [E, A, B]: the caller's draw (a recorded reference run) instead of the documented Floyd sampling."""
        from oracle.nets_oracle import sample_uniform
        size = self.rings[0][0].size
        idx = np.zeros((self.E, self.A, self.B), np.int32)
        losses, norms, grads = [], [], []
        for a, q in enumerate(self.qs):
            obs, acts, nobs, rs, dones = [], [], [], [], []
            for e in range(self.E):
                p = e * self.A + a
                ids = (floyd_sample(size, self.B, lambda i: sample_uniform(self.replay_seed, self.update_step, p * self.B + i))
                       if idx_given is None else [int(x) for x in idx_given[e, a]])
                idx[e, a] = ids
                for s in ids:
                    ob, ac, r, nob, d = self.rings[e][a].buffer[s]
                    obs.append(ob); acts.append(ac); rs.append(r); nobs.append(nob); dones.append(d)
            loss, g = q.loss_and_grads(obs, acts, nobs, dones, rs)
            grads.append({k: v.numpy().copy() for k, v in g.items()})
            l2, norm = q.backward(obs, acts, nobs, dones, rs, lr)
            losses.append(loss); norms.append(norm)
        self.update_step += 1
        self.last_idx = idx
        return np.array(losses), np.array(norms), grads

    def agent_params(self):
        return [{k: v.numpy().astype(np.float32) for k, v in q.p.items()} for q in self.qs]
