#!/usr/bin/env python
"""Sim-only timing of tsc_env_step at a realistic traffic state, one line per run (A/B runs over the TSC_ENV_* knobs).
    python tools/bench_env_quick.py [E] [scenario] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from deeprl_signal_control_amd.env import VecTrafficEnv
from deeprl_signal_control_amd.scenario import build_scenario

E = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
name = sys.argv[2] if len(sys.argv) > 2 else 'large_grid'
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 240
scn = build_scenario(name, 'ma2c')
env = VecTrafficEnv(scn, E, seed=12)
env.reset()
g = torch.Generator(device='cuda'); g.manual_seed(0)
na = torch.as_tensor(scn.n_a_ls, device='cuda')
acts = [(torch.rand(E, scn.n_agent, generator=g, device='cuda') * na).to(torch.int32).contiguous() for _ in range(16)]
out = []
for phase, n in (('warm', 240), ('timed', steps)):
    env.live_vehicle_mean(1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        env.step(acts[i % 16])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    out.append('%s %.1f us/step V=%.0f' % (phase, 1e6 * dt / n, env.live_vehicle_mean(n)))
print('E=%d %s threads=%s kf=%s: %s' % (E, name, os.environ.get('TSC_ENV_THREADS', '-'), os.environ.get('TSC_ENV_KF', '-'), '; '.join(out)))
env.close()
