#!/usr/bin/env python
"""What does the rollout forward's activation cache cost?  (the blocked cache layout [N/32][8][32][8] -- or the
measurement that rejects it.)  tsc_model_forward_sample at E = 1024 with cache=True (the step's X1 / gate / h / c rows are streamed
into the update's buffers: 193 MB per launch) against cache=False (the same kernel without those stores), alternating, 100 launches each.
A layout change can at most recover the difference.

    python tools/bench_fwd_cache.py"""
import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from deeprl_signal_control_amd.agents import VecA2C
from deeprl_signal_control_amd.scenario import build_large_grid
E = 1024
scn = build_large_grid('ma2c')
m = VecA2C(scn.n_s_ls, scn.n_a_ls, scn.n_w_ls, scn.n_f_ls, E, scn.s_max, 5, {}, device=0, seed=0, name='ma2c')
obs = torch.rand(E, 25, scn.s_max, device='cuda')
done = torch.zeros(E, dtype=torch.uint8, device='cuda')
for cache in (True, False, True, False):
    for _ in range(10):
        m.forward_sample(obs, done, cache=cache)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(100):
        m.forward_sample(obs, done, cache=cache)
    b.record(); torch.cuda.synchronize()
    print('cache', cache, '%.1f us' % (a.elapsed_time(b) * 10))
