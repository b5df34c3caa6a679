#!/usr/bin/env python
"""Sim-only timing of the env kernel (tsc_env_step) at a realistic traffic state.
    python tools/bench_env.py [E] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from deeprl_signal_control_amd.env import VecTrafficEnv
from deeprl_signal_control_amd.scenario import build_large_grid
from deeprl_signal_control_amd.trainer import greedy_actions_large_grid

E = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
scn = build_large_grid('ma2c')
for chunk in ('1',):
    env = VecTrafficEnv(scn, E, seed=12)
    obs = env.reset()
    g = torch.Generator(device='cuda'); g.manual_seed(0)
    acts = [torch.randint(0, 5, (E, 25), generator=g, device='cuda', dtype=torch.int32) for _ in range(16)]
    for phase, n in (('warm-up to t=1500s (random actions)', 300), ('timed', steps)):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(n):
            obs, r, d, gr = env.step(acts[i % 16])
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print('chunk=%s %s: %.1f us/control-step, %.0f live veh/env, %.3g env-steps/s (sim only)'
              % (chunk, phase, 1e6 * dt / n, env.mean_live_vehicles(), 25 * E * 5 * n / dt))
    env.close()
