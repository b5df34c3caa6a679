"""CPU restatement of the reference's independent Q-learning agents -- TEST INFRASTRUCTURE ONLY
(oracle/__init__.py).  Groundwork for the SURVEY 8(f) rank-1 row; no HIP path exists yet.

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
