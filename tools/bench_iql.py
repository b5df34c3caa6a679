#!/usr/bin/env python
"""Times the IQL-DNN learner's launches in isolation (large_grid, E = 1024 by default): the minibatch gradient
(tsc_iql_compute_grads: sample + fused gradient + reduce) and the acting forward, with HIP events on the launch stream.

    python tools/bench_iql.py [--envs 1024] [--reps 50] [--scenario large_grid]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np      # noqa: E402
import torch            # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--envs', type=int, default=1024)
    ap.add_argument('--reps', type=int, default=50)
    ap.add_argument('--scenario', default='large_grid')
    ap.add_argument('--stamps', action='store_true', help='phase stamps of workgroup 0 and the start / end of every workgroup (tsc_iql_debug_clock)')
    args = ap.parse_args()
    from deeprl_signal_control_amd import _lib
    from deeprl_signal_control_amd.iql import VecIQL
    from deeprl_signal_control_amd.scenario import build_scenario
    scn = build_scenario(args.scenario, 'iqld')
    E, A = args.envs, scn.n_agent
    m = VecIQL(scn.n_s_ls, scn.n_a_ls, scn.n_w_ls, E, scn.s_max, int(scn.green_tab.shape[1]),
               dict(batch_size=20, buffer_size=1000, reward_norm=3000.0), total_step=10 ** 6, seed=0, model_type='dqn')
    g = torch.Generator(device='cuda'); g.manual_seed(0)
    mask = torch.zeros(A, scn.s_max, device='cuda')
    for a, n in enumerate(scn.n_s_ls):
        mask[a, :n] = 1
    obs = torch.rand(E, A, scn.s_max, generator=g, device='cuda') * 2 * mask
    for t in range(40):
        nobs = torch.rand(E, A, scn.s_max, generator=g, device='cuda') * 2 * mask
        act = (torch.rand(E, A, generator=g, device='cuda') * torch.as_tensor(scn.n_a_ls, device='cuda')).to(torch.int32)
        rew = -torch.rand(E, A, generator=g, device='cuda', dtype=torch.float64) * 6000.0
        done = (torch.rand(E, generator=g, device='cuda') < 0.05).to(torch.uint8)
        m.add_transition(obs, act, rew, nobs, done)
        obs = nobs

    def timed(fn, reps):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a_.record()
        for _ in range(reps):
            fn()
        b_.record()
        torch.cuda.synchronize()
        return a_.elapsed_time(b_) / reps * 1e3

    step = [0]

    def grads():
        _lib.check(m._L.tsc_iql_compute_grads(m._h, 7, step[0]))
        step[0] += 1
    out = {'fused': m.fused, 'E': E, 'compute_grads_us': timed(grads, args.reps),
           'forward_us': timed(lambda: m.forward(obs, mode='explore'), args.reps),
           'minibatch_step_us': timed(lambda: m.minibatch_step(1e-4), args.reps)}
    if args.stamps and m.fused:
        import ctypes as C
        n = 64 + 2 * 4096
        buf = np.zeros(n, np.int64)
        _lib.check(m._L.tsc_iql_debug_clock(m._h, 1, None, 0))
        for _ in range(3):
            grads()
        _lib.check(m._L.tsc_iql_debug_clock(m._h, 1, buf.ctypes.data_as(C.c_void_p), n))
        st = buf[:64].reshape(4, 16)[:, :11]
        names = ['nets', 'td+stage', 'dX1', 'bar1', 'B', 'bar2', 'C', 'bar3', 'D', 'bar4']
        out['phase_cycles_per_wave'] = {nm: [int(st[w, k + 1] - st[w, k]) for w in range(4)] for k, nm in enumerate(names)}
        out['chunk_cycles'] = [int(st[w, 10] - st[w, 0]) for w in range(4)]
        fine = buf[:64].reshape(4, 16)
        if fine[0, 11]:
            out['nets_fine'] = {'L1': [int(fine[w, 11] - fine[w, 0]) for w in range(4)], 'relu1': [int(fine[w, 12] - fine[w, 11]) for w in range(4)], 'L2': [int(fine[w, 13] - fine[w, 12]) for w in range(4)], 'relu2': [int(fine[w, 14] - fine[w, 13]) for w in range(4)], 'Q': [int(fine[w, 1] - fine[w, 14]) for w in range(4)]}
        wg = buf[64:].reshape(-1, 2)
        wg = wg[wg[:, 0] > 0]
        t0 = wg[:, 0].min()
        out['workgroups'] = int(len(wg))
        out['wg_us'] = {'first_start': 0.0, 'last_start': float((wg[:, 0].max() - t0) / 100.0), 'min_dur': float((wg[:, 1] - wg[:, 0]).min() / 100.0),
                        'median_dur': float(np.median(wg[:, 1] - wg[:, 0]) / 100.0), 'max_dur': float((wg[:, 1] - wg[:, 0]).max() / 100.0),
                        'span': float((wg[:, 1].max() - t0) / 100.0)}
    gsum = float(m.grad_tensor().double().abs().sum().item())
    out['grad_abs_sum'] = gsum
    print(json.dumps(out))
    m.close()


if __name__ == '__main__':
    main()
