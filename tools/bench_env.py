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




def phase_names(n):
    """Labels of the shader-clock stamps of step_kernel (with / without the flat phase)."""
    per = ['H%d', 'bar', 'F%d', 'bar', 'B%d', 'bar'] if n >= 1 + 5 * 6 + 4 else ['A%d', 'bar', 'B%d', 'bar']
    names = ['prologue'] + sum([[p % k if '%' in p else p for p in per] for k in range(5)], [])
    names += ['detectors', 'bar', 'obs', 'reward']
    return names + ['?'] * max(0, n - len(names))


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
    import ctypes as C
    from deeprl_signal_control_amd import _lib
    st = (C.c_int64 * 64)()
    _lib.check(env._L.tsc_env_debug_clock(env._h, 1, None))
    env.step(acts[0]); env.step(acts[1])
    _lib.check(env._L.tsc_env_debug_clock(env._h, 1, st))
    n = st[63]
    d = [st[i + 1] - st[i] for i in range(n - 1)]
    names = phase_names(len(d))
    print('workgroup 0 / thread 0 shader-clock cycles per phase (total %d):' % (st[n - 1] - st[0]))
    print('  ' + ', '.join('%s=%d' % (names[i], d[i]) for i in range(len(d)) if names[i] != 'bar'))
    env.close()

# --- cache / interference experiment: how much slower is env_step when other kernels run in between?
from deeprl_signal_control_amd import _lib as _l
env = VecTrafficEnv(scn, E, seed=12)
env.reset()
for i in range(300):
    env.step(acts[i % 16])
big = torch.zeros(64 << 20, device='cuda')           # 256 MB
for label, fn in (('nothing between', lambda: None), ('256 MB memset between', lambda: big.zero_()),
                  ('16 MB memset between', lambda: big[:4 << 20].zero_())):
    _l.profile(enable=True, reset=True)
    for i in range(60):
        env.step(acts[i % 16]); fn()
    p = _l.profile()
    _l.profile(enable=False)
    print('env_step avg %.1f us with %s' % (1e3 * p['env_step'][0] / p['env_step'][1], label))

# --- phase stamps with and without an interfering kernel
import ctypes as C
st = (C.c_int64 * 64)()
_l.check(env._L.tsc_env_debug_clock(env._h, 1, None))
for label, fn in (('nothing between', lambda: None), ('16 MB memset between', lambda: big[:4 << 20].zero_())):
    for i in range(4):
        env.step(acts[i]); fn()
    torch.cuda.synchronize()
    _l.check(env._L.tsc_env_debug_clock(env._h, 1, st))
    n = st[63]
    d = [st[i + 1] - st[i] for i in range(n - 1)]
    names = phase_names(len(d))
    print('%s: total %d cycles: ' % (label, st[n - 1] - st[0]) + ', '.join('%s=%d' % (names[i], d[i]) for i in range(len(d)) if names[i] != 'bar'))

# --- per-workgroup wall-clock (100 MHz) start/end: is it the blocks or the dispatch that gets slower?
import numpy as np
buf = (C.c_int64 * (64 + 2 * E))()
for label, fn in (('nothing between', lambda: None), ('16 MB memset between', lambda: big[:4 << 20].zero_())):
    for i in range(4):
        env.step(acts[i]); fn()
    torch.cuda.synchronize()
    _l.check(env._L.tsc_env_debug_clock(env._h, 2, buf))
    w = np.array(buf[64:], dtype=np.int64).reshape(E, 2)
    tag = (w[:, 1] >> 48) & 0xFFFF
    w[:, 1] &= 0xFFFFFFFFFFFF
    w[:, 0] &= 0xFFFFFFFFFFFF
    st0, en0 = w[:, 0] - w[:, 0].min(), w[:, 1] - w[:, 0].min()
    dur = (w[:, 1] - w[:, 0]) / 100.0
    print('%s: kernel span %.1f us; block start p50/p99/max %.1f/%.1f/%.1f us; block duration min/p50/max %.1f/%.1f/%.1f us'
          % (label, en0.max() / 100.0, np.percentile(st0, 50) / 100.0, np.percentile(st0, 99) / 100.0, st0.max() / 100.0,
             dur.min(), np.median(dur), dur.max()))
    late = np.where(st0 > 5000)[0]
    print('   late blocks: %d; (block, start_us, xcc, hwid) %s' % (len(late), [(int(b), float(st0[b]) / 100.0, int(tag[b] >> 12), hex(int(tag[b] & 0xFFF))) for b in late[:12]]))
    import collections
    cnt = collections.Counter((int(t >> 12), int(t & 0xFFF) >> 4) for t in tag)
    print('   blocks per (xcc, hwid>>4): max %d min %d distinct %d' % (max(cnt.values()), min(cnt.values()), len(cnt)))
