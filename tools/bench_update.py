#!/usr/bin/env python
"""Time the A2C update's kernels alone (no simulator): fill one rollout of the benchmark shape through the fused forward
(random observations, E env instances, T = n_step), then run compute_grads `--reps` times with HIP-event timing on.
    python tools/bench_update.py [--envs 1024] [--agent ma2c] [--reps 3]
Environment knobs of the library (TSC_UNFUSED_DW, TSC_UNFUSED_DX) select kernel variants for A/B runs."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np      # noqa: E402
import torch            # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--envs', type=int, default=1024)
    ap.add_argument('--agent', default='ma2c')
    ap.add_argument('--scenario', default='large_grid')
    ap.add_argument('--reps', type=int, default=3)
    args = ap.parse_args()
    from deeprl_signal_control_amd import _lib
    from deeprl_signal_control_amd.agents import VecA2C
    from deeprl_signal_control_amd.scenario import build_scenario
    scn = build_scenario(args.scenario, args.agent)
    T = 120 if args.scenario == 'large_grid' else 40
    m = VecA2C(scn.n_s_ls, scn.n_a_ls, scn.n_w_ls, scn.n_f_ls, args.envs, scn.s_max, int(scn.green_tab.shape[1]),
               dict(batch_size=T), device=0, seed=0, name=args.agent)
    sl = m.rollout_slots()
    g = torch.Generator(device='cuda'); g.manual_seed(0)
    out = {}
    for rep in range(args.reps + 1):
        m.reset()
        sl['obs'].copy_(torch.rand(sl['obs'].shape, generator=g, device='cuda') * 2)
        sl['done'].zero_(); sl['done'][0].fill_(1)
        sl['reward'].copy_(-torch.rand(sl['reward'].shape, generator=g, device='cuda', dtype=torch.float64) * 4000)
        m.cur_t = 0
        for t in range(T):
            m.forward_sample(sl['obs'][t], sl['done'][t], v_out=sl['value'][t], action_out=sl['action'][t])
            m.commit_transition()
        R = m.forward(sl['obs'][T], False, 'v')
        torch.cuda.synchronize()
        if rep == 1:
            _lib.profile(enable=1, reset=True)
        m.backward(R)
    torch.cuda.synchronize()
    prof = _lib.profile()
    _lib.profile(enable=False)
    tot = 0.0
    for k, (ms, cnt) in sorted(prof.items(), key=lambda kv: -kv[1][0]):
        if k in ('policy_fwd_fused',):
            continue
        per = ms / args.reps
        tot += per
        print('%-18s %8.3f ms per update  (%d launches)' % (k, per, cnt))
    print('%-18s %8.3f ms' % ('update total', tot))


if __name__ == '__main__':
    main()
