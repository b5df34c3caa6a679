#!/usr/bin/env python
"""Learning curve of the HIP path: large_grid MA2C (config/config_ma2c_large.ini of the reference), E env instances on one
MI355X, `--episodes` training episodes of 720 control steps = 6 A2C updates each (Trainer.run, utils.py:255-308).

Writes the rows of the reference's train_reward.csv (one per episode: mean / std over the episode's control steps of the
global reward, averaged over the env instances -- the quantity of figs/large_grid_train.png) plus the episode's wall time:

    python tools/learning_curve.py --episodes 50 --envs 1024 --out profiles/r03_learning_curve.json

Every instance sees its own demand seed per episode (seed0 + e, stride E per episode), every instance explores with its
own action stream, all share one set of weights.  The simulator underneath is this repo's microsim spec (MICROSIM_SPEC.md):
absolute reward levels are not SUMO's, the trend is what this shows."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np          # noqa: E402
import torch                # noqa: E402


def run(episodes, n_env, scenario='large_grid', agent='ma2c', seed=0, lr=None, log=None, policy='lstm', scn_kw=None, test_seeds=None):
    from deeprl_signal_control_amd.agents import VecA2C
    from deeprl_signal_control_amd.env import VecTrafficEnv
    from deeprl_signal_control_amd.scenario import build_scenario
    from deeprl_signal_control_amd.trainer import VecTrainer
    scn = build_scenario(scenario, agent, **(scn_kw or {}))
    if scenario == 'large_grid':
        mcfg, seed0 = dict(reward_norm=2000.0 if agent == 'ma2c' else 3000.0, batch_size=120), 12
    else:
        mcfg, seed0 = dict(reward_norm=1.0, batch_size=40), 42
    if lr is not None:
        mcfg['lr_init'] = lr
    env = VecTrafficEnv(scn, n_env, device=0, seed=seed0, **({'test_seeds': tuple(test_seeds)} if test_seeds else {}))
    is_q = agent in ('iqld', 'iqll')
    if is_q:
        # config/config_iql{d,l}_large.ini; total_step = the run's control steps (epsilon decays linearly to its floor over the first half)
        from deeprl_signal_control_amd.iql import VecIQL
        qcfg = dict(batch_size=20, buffer_size=1000, reward_norm=3000.0 if scenario == 'large_grid' else 1.0)
        if lr is not None:
            qcfg['lr_init'] = lr
        model = VecIQL(scn.n_s_ls, scn.n_a_ls, scn.n_w_ls, n_env, scn.s_max, int(scn.green_tab.shape[1]), qcfg,
                       total_step=episodes * int(env.T), device=0, seed=seed, model_type='dqn' if agent == 'iqld' else 'lr')
    else:
        model = VecA2C(scn.n_s_ls, scn.n_a_ls, scn.n_w_ls, scn.n_f_ls, n_env, scn.s_max, int(scn.green_tab.shape[1]), mcfg,
                       device=0, seed=seed, name=agent, policy=policy)
    tr = VecTrainer(env, model, log_rewards=True)
    rows = []
    T = int(env.T)
    for ep in range(episodes):
        t0 = time.perf_counter()
        tr.start_episode()
        tr._ep_rewards = []
        while True:
            finished, R = tr.explore()
            model.backward() if is_q else model.backward(R)
            if finished:
                env.terminate()
                break
        r = torch.stack(tr._ep_rewards)                                  # [T, E] global reward per control step
        torch.cuda.synchronize()
        row = dict(episode=ep, step=(ep + 1) * T, avg_reward=float(r.mean(0).mean().item()),
                   std_reward=float(r.std(0, unbiased=False).mean().item()),
                   spread_over_instances=float(r.mean(0).std().item()), wall_s=time.perf_counter() - t0)
        rows.append(row)
        if log:
            log('episode %3d  avg step reward %9.2f  (std over steps %.1f, over instances %.1f)  %.2f s'
                % (ep, row['avg_reward'], row['std_reward'], row['spread_over_instances'], row['wall_s']))
    # the reference's evaluation (Trainer.run's test block, utils.py:257-275): every test seed, sampled and argmax policy
    ev = {pt: tr.evaluate(policy_type=pt) for pt in ('default', 'deterministic')}
    if log:
        for pt, rws in ev.items():
            log('evaluation (%s policy): %s' % (pt, ', '.join('seed %d: %.1f' % (r['test_id'], r['avg_reward']) for r in rws)))
    # the scenario's greedy controller on the same test seeds (the reference's baseline, envs/*_env.py controllers on the device)
    class _Greedy:
        name, n_step = 'greedy', 1
        def forward(self, ob, *a, **k): return env.greedy_actions(ob)
        def reset(self): pass
    was = env.train_mode
    env.train_mode = False
    gtr = VecTrainer(env, _Greedy())
    gtr.agent = 'greedy'
    gm, _ = gtr.perform(np.arange(n_env) % env.test_num, 'default')
    env.train_mode = was
    ev['greedy'] = [dict(test_id=int(k), avg_reward=float(np.mean(gm[np.arange(n_env) % env.test_num == k]))) for k in range(env.test_num)]
    if log:
        log('greedy controller: %s' % ', '.join('seed %d: %.1f' % (r['test_id'], r['avg_reward']) for r in ev['greedy']))
    env.close(); model.close()
    return rows, ev


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--episodes', type=int, default=50)
    ap.add_argument('--envs', type=int, default=1024)
    ap.add_argument('--scenario', default='large_grid')
    ap.add_argument('--agent', default='ma2c', help='ma2c | ia2c | iqld | iqll (the IQL agents: epsilon-greedy exploration, replay rings, Adam)')
    ap.add_argument('--lr', type=float, default=None)
    ap.add_argument('--policy', default='lstm', choices=['lstm', 'fc'], help='fc = FcACPolicy (ia2c only; BASELINE configs[1])')
    ap.add_argument('--lane-change', type=int, default=None, help='large_grid: 1 / 0 = with / without MICROSIM_SPEC.md rule 10 (default: the scenario default)')
    ap.add_argument('--test-seeds', default=None, help='comma-separated evaluation seeds (default: the config\'s test_seeds)')
    ap.add_argument('--out', default=None)
    args = ap.parse_args()
    kw = {} if args.lane_change is None else {'lane_change': bool(args.lane_change)}
    rows, ev = run(args.episodes, args.envs, args.scenario, args.agent, lr=args.lr, log=print, policy=args.policy, scn_kw=kw,
                   test_seeds=[int(x) for x in args.test_seeds.split(',')] if args.test_seeds else None)
    first, last = np.mean([r['avg_reward'] for r in rows[:5]]), np.mean([r['avg_reward'] for r in rows[-5:]])
    out = dict(scenario=args.scenario, agent=args.agent, policy=args.policy, envs=args.envs, episodes=args.episodes, scenario_options=kw,
               control_steps_per_episode=rows[0]['step'], first5_mean=first, last5_mean=last, rows=rows, evaluation=ev,
               note='mean over env instances of the per-episode mean global step reward (train_reward.csv avg_reward); '
                    'this repo\'s microsim spec underneath, not SUMO')
    print('first 5 episodes %.2f -> last 5 episodes %.2f' % (first, last))
    if args.out:
        with open(args.out, 'w') as f:
            json.dump(out, f, indent=1)


if __name__ == '__main__':
    main()
