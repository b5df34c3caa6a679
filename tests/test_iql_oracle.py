"""Groundwork for the IQL row: the CPU oracle of the reference's Q-learning agents, pinned
where the reference can run here (ReplayBuffer over a seeded `random`) and self-checked where it cannot
(TF graph code restated: gradient vs finite differences, TF1 Adam closed form)."""
import os
import random

import numpy as np
import torch

from oracle.iql_oracle import OracleQ, ReplayBuffer, act_epsilon_greedy, q_net


def test_replay_buffer_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, 'iql_known_answers.npz'))
    buf = ReplayBuffer(1000, 20)
    random.seed(5)
    sizes, draws = [], []
    for i in range(1500):
        buf.add_transition(np.array([float(i)]), i % 5, -0.001 * i, np.array([float(i + 1)]), (i % 720) == 719)
        if i in (10, 19, 20, 999, 1000, 1499):
            sizes.append([i, buf.size, buf.cum_size])
        if i >= 19 and i % 97 == 0:
            obs, acts, nobs, rs, dones = buf.sample_transition()
            draws.append(np.concatenate([[i], obs[:, 0], acts, nobs[:, 0], rs, dones.astype(np.float64)]))
    np.testing.assert_array_equal(np.array(sizes), g['sizes'])
    np.testing.assert_array_equal(np.array(draws), g['draws'])
    np.testing.assert_array_equal(np.array([t[0][0] for t in buf.buffer]), g['content'])


def _params(rng, n_s, n_w, n_a, kind):
    if kind == 'lr':
        return {'q_w': rng.randn(n_s + n_w, n_a) * 0.3, 'q_b': rng.randn(n_a) * 0.1}
    p = {'fcw_w': rng.randn(n_s, 16) * 0.3, 'fcw_b': rng.randn(16) * 0.1,
         'fc0_w': rng.randn(16 + (4 if n_w else 0), 8) * 0.3, 'fc0_b': rng.randn(8) * 0.1,
         'q_w': rng.randn(8, n_a) * 0.3, 'q_b': rng.randn(n_a) * 0.1}
    if n_w:
        p.update({'fct_w': rng.randn(n_w, 4) * 0.3, 'fct_b': rng.randn(4) * 0.1})
    return p


def test_q_loss_gradient_and_adam_step():
    rng = np.random.RandomState(0)
    for kind, n_w in (('lr', 0), ('dqn', 3), ('dqn', 0)):
        n_s, n_a, B = 7, 4, 20
        q = OracleQ(_params(rng, n_s, n_w, n_a, kind), n_s, n_w, gamma=0.9, max_grad_norm=0.5)
        obs, nobs = rng.rand(B, n_s + n_w), rng.rand(B, n_s + n_w)
        acts, rs = rng.randint(0, n_a, B), -rng.rand(B)
        dones = rng.rand(B) < 0.2
        loss, grads = q.loss_and_grads(obs, acts, nobs, dones, rs)
        # finite differences on a few coordinates; the target is a constant (stop_gradient, same network)
        with torch.no_grad():
            tq = torch.where(torch.as_tensor(dones), torch.as_tensor(rs),
                             torch.as_tensor(rs) + 0.9 * q_net(q.p, torch.as_tensor(nobs), n_s, n_w).max(1).values)
        for k in ('q_w', 'q_b'):
            flat = q.p[k].reshape(-1)
            for idx in (0, flat.numel() - 1):
                old = flat[idx].item()
                vals = []
                for d in (1e-6, -1e-6):
                    flat[idx] = old + d
                    with torch.no_grad():
                        q0 = q_net(q.p, torch.as_tensor(obs), n_s, n_w).gather(1, torch.as_tensor(acts)[:, None])[:, 0]
                        vals.append(((q0 - tq) ** 2).mean().item())
                flat[idx] = old
                assert abs((vals[0] - vals[1]) / 2e-6 - grads[k].reshape(-1)[idx].item()) < 1e-5
        # one Adam step of TF1: with m = (1-b1) g, v = (1-b2) g^2 the first update is -lr * g / (|g| + eps')
        before = {k: v.clone() for k, v in q.p.items()}
        norm = np.sqrt(sum((g ** 2).sum().item() for g in grads.values()))
        sc = 0.5 / max(norm, 0.5)
        q.backward(obs, acts, nobs, dones, rs, 1e-3)
        for k in before:
            g = grads[k] * sc
            lr_t = 1e-3 * np.sqrt(1 - 0.999) / (1 - 0.9)
            want = before[k] - lr_t * (0.1 * g) / (torch.sqrt(0.001 * g * g) + 1e-8)
            np.testing.assert_allclose(q.p[k].numpy(), want.numpy(), rtol=1e-12, atol=1e-15)


def test_epsilon_greedy_rule():
    qs = np.array([0.1, 0.7, 0.7, -1.0])
    assert act_epsilon_greedy(qs, 0.3, 0.31, 0.99) == 1          # exploit: first argmax
    assert act_epsilon_greedy(qs, 0.3, 0.29, 0.99) == 3          # explore: floor(u * n_a)
    assert act_epsilon_greedy(qs, 0.3, 0.0, 0.0) == 0
