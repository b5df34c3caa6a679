#!/usr/bin/env python
"""Timing / phase breakdown of the fused rollout forward as the trainer runs it
(tsc_model_forward_sample with the activation cache on)."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from deeprl_signal_control_amd import _lib
from deeprl_signal_control_amd.agents import VecA2C
from deeprl_signal_control_amd.scenario import build_large_grid

E = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
scn = build_large_grid('ma2c')
m = VecA2C(scn.n_s_ls, scn.n_a_ls, scn.n_w_ls, scn.n_f_ls, E, scn.s_max, 5, {}, device=0, seed=0, name='ma2c')
obs = torch.rand(E, 25, scn.s_max, device='cuda')
done = torch.zeros(E, dtype=torch.uint8, device='cuda')
big = torch.zeros(16 << 20, device='cuda')
for label, fn in (('back to back', lambda: None), ('64 MB memset between', lambda: big.zero_())):
    for _ in range(5):
        m.forward_sample(obs, done, cache=True); fn()
    _lib.profile(enable=True, reset=True)
    for _ in range(50):
        m.forward_sample(obs, done, cache=True); fn()
    p = _lib.profile(); _lib.profile(enable=False)
    print('fused forward %s: %.1f us' % (label, 1e3 * p['policy_fwd_fused'][0] / p['policy_fwd_fused'][1]))
    nblk = 8 * ((m.G + 7) // 8) * ((E + 63) // 64)
    buf = (C.c_int64 * (64 + 2 * nblk))()
    _lib.check(m._L.tsc_model_debug_clock(m._h, 1, None, 0))
    for _ in range(3):
        m.forward_sample(obs, done, cache=True); fn()
    _lib.check(m._L.tsc_model_debug_clock(m._h, 1, buf, 64 + 2 * nblk))
    n = buf[63]
    names = ['obs->LDS', 'fc (X1)', 'state', 'gates MFMA', 'barrier', 'cell', 'head']
    if n > 9:                                            # weight-stationary kernel: prologue, then 6 stamps per tile
        names = ['prologue', 'fill'] + sum([['gates+head%d' % i, 'bar%d' % i, 'cell%d' % i, 'fc%d' % i, 'bar2_%d' % i] for i in range(12)], [])
    print('   phases (cycles): ' + ', '.join('%s=%d' % (names[i], buf[i + 1] - buf[i]) for i in range(n - 1)) + '  total=%d' % (buf[n - 1] - buf[0]))
    w = np.array(buf[64:], dtype=np.int64).reshape(nblk, 2)
    live = w[:, 1] > 0
    if not live.any():
        continue
    if n > 9:          # weight-stationary kernel: which workgroups are the long ones (13 or 12 half tiles of 16 instances)
        S = max(1, 256 // m.G)
        d_all = (w[:, 1] - w[:, 0]) / 100.0
        e_all = (w[:, 1] - w[live, 0].min()) / 100.0
        order = np.argsort(-e_all)[:8]
        print('   last workgroups to end (block id: start, duration, end us): ' +
              ', '.join('%d: %.1f %.1f %.1f' % (b, (w[b, 0] - w[live, 0].min()) / 100.0, d_all[b], e_all[b]) for b in order))
        hist = np.histogram(d_all[live], bins=8)
        print('   duration histogram (us): ' + ', '.join('%.1f-%.1f: %d' % (hist[1][i], hist[1][i + 1], hist[0][i]) for i in range(8)))
    st = (w[live, 0] - w[live, 0].min()) / 100.0
    en = (w[live, 1] - w[live, 0].min()) / 100.0
    dur = en - st
    print('   %d working blocks: span %.1f us; start p50/p90/max %.1f/%.1f/%.1f; duration min/p50/max %.1f/%.1f/%.1f us'
          % (live.sum(), en.max(), np.percentile(st, 50), np.percentile(st, 90), st.max(), dur.min(), np.median(dur), dur.max()))
