#!/usr/bin/env python
"""SURVEY 8(d) CPU baseline (i): "Oracle-A" -- the REFERENCE's own env class (envs/large_grid_env.py + envs/env.py, imported
unmodified from /root/reference) driving oracle/fake_traci.py over the C microsim, E = 1, one core -- timed.  It is the closest
runnable stand-in for "the reference env on a CPU": the reference's real Python / TraCI-call overhead with SUMO's process
replaced by an in-process C library (so it is an UPPER bound on what the reference's env sustains with a real SUMO behind a
socket).  Second figure: the reference's whole training loop (agents/models.py MA2C + utils.py Trainer.run, unmodified) over
oracle/fake_tf.py (float64 torch standing in for TensorFlow 1.12) on that env, one shortened episode.

Build container only (/root/reference does not exist on the GPU box): writes profiles/r05_oracle_a.json, which bench.py
quotes inside `cpu_baseline`.

    python tools/time_oracle_a.py [control_steps=240] [episode_sec=600]
"""
import json
import os
import platform
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def time_env(n_ctrl):
    from oracle import fake_traci
    env = fake_traci.ref_env('large_grid', 'ma2c')
    rng = np.random.RandomState(0)
    ob = env.reset()
    A = len(ob)
    # warm the network up to the demand ramp with untimed steps (t = 0 .. 600 s), then time n_ctrl control steps
    t_step = t_fp = 0.0
    live = []
    for k in range(120 + n_ctrl):
        pol = list(rng.dirichlet(np.ones(5), size=A).astype(np.float32))
        act = [int(a) for a in rng.randint(0, 5, A)]
        t0 = time.perf_counter()
        env.update_fingerprint(pol)
        t1 = time.perf_counter()
        ob, r, done, g = env.step(act)
        t2 = time.perf_counter()
        if k >= 120:
            t_fp += t1 - t0; t_step += t2 - t1
            live.append(fake_traci._SCN_FOR_CONNECT['last'].ms.totals()['live'])
        if done:
            break
    env.terminate()
    n = len(live)
    return dict(value=A * 5 * n / (t_step + t_fp), unit='env-steps/s', control_steps=n, seconds=t_step + t_fp,
                ms_per_control_step=1e3 * (t_step + t_fp) / n, mean_live_vehicles=float(np.mean(live)),
                what='reference LargeGridEnv.step + update_fingerprint (envs/env.py:566-635, MA2C config) over oracle/fake_traci.py + '
                     'oracle/microsim.c, E = 1, 1 core, random actions, t = 600 s .. %d s of the episode' % (600 + 5 * n))


def time_loop(episode_sec):
    from oracle import refnet
    t0 = time.perf_counter()
    fx = refnet.run_reference_a2c('large_grid', 'ma2c', seed_w=0, episode_sec=episode_sec)
    dt = time.perf_counter() - t0
    n_ctrl = len(fx['actions'])
    return dict(value=25 * 5 * n_ctrl / dt, unit='env-steps/s', control_steps=n_ctrl, updates=int(fx['n_backward']), seconds=dt,
                what='reference MA2C + Trainer.run (agents/models.py, agents/policies.py, utils.py, unmodified) over oracle/fake_tf.py '
                     '(float64 torch in place of TensorFlow 1.12; instrumented recorder attached) on the env above: one episode of %d s, '
                     'E = 1, incl. graph construction' % episode_sec)


if __name__ == '__main__':
    n_ctrl = int(sys.argv[1]) if len(sys.argv) > 1 else 240
    ep = int(sys.argv[2]) if len(sys.argv) > 2 else 600
    import torch
    torch.set_num_threads(1)
    out = dict(host=dict(machine=platform.machine(), cpus=os.cpu_count(), python=platform.python_version(),
                         note='build container (the GPU box has no /root/reference); one core'),
               oracle_a_env=time_env(n_ctrl), oracle_a_training_loop=time_loop(ep))
    path = os.path.join(ROOT, 'profiles', 'r05_oracle_a.json')
    json.dump(out, open(path, 'w'), indent=1)
    print(json.dumps(out, indent=1))
