#!/usr/bin/env python
"""Round-6 experiment behind MICROSIM_SPEC.md "Krauss car following" (VERDICT r05 item 9): SUMO's default car-following model
(large_grid/data/build_file.py:279 names none, so SUMO runs Krauss with sigma = 0.5) against this repo's spec (IDM acceleration
clamped by the Krauss safe speed, no dawdling), on the CPU oracle only (oracle/microsim.c ms_set_krauss, OFF by default and not part
of the spec), under the reference's greedy controllers -- the aggregates tests/test_microsim_anchors.py gates on.

    python tools/sweep_krauss.py        -> profiles/r06_krauss_sweep.json
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deeprl_signal_control_amd.scenario import build_large_grid, build_real_net      # noqa: E402
from deeprl_signal_control_amd.trainer import greedy_actions                         # noqa: E402
from oracle.env_oracle import OracleEnv, greedy_large_grid                           # noqa: E402


def episode(scn, seed, krauss, sigma):
    env = OracleEnv(scn, seed=seed, train_mode=False, test_seeds=(seed,))
    env.ms.L.ms_set_krauss(int(krauss), float(sigma))
    try:
        ob = env.reset(0)
        L = scn.agent_lanes.shape[1]
        rs = []
        while True:
            if scn.name == 'large_grid':
                act = [greedy_large_grid(o[:6]) for o in ob]
            else:
                w = np.zeros((scn.n_agent, L))
                for a, o in enumerate(ob):
                    w[a, :len(o)] = o
                act = list(greedy_actions(scn, w))
            ob, r, done, g = env.step(act)
            rs.append(g)
            if done:
                break
        tot = env.ms.totals()
    finally:
        env.ms.L.ms_set_krauss(0, 0.5)
    return dict(reward=float(np.mean(rs)), arrived=int(tot['arrived']), departed=int(tot['departed']), pending=int(tot['pending']),
                teleported=int(tot['teleported']))


if __name__ == '__main__':
    out = {}
    scns = {'large_grid': (build_large_grid('greedy', norm_wave=1.0, norm_wait=1.0, clip_wave=-1.0, clip_wait=-1.0), (10000, 20000, 30000, 40000)),
            'real_net': (build_real_net('greedy', norm_wave=1.0, clip_wave=-1.0), (10000, 20000, 30000))}
    variants = [('spec (IDM + safe speed)', 0, 0.0), ('Krauss sigma 0', 1, 0.0), ('Krauss sigma 0.25', 1, 0.25), ('Krauss sigma 0.5 (SUMO default)', 1, 0.5)]
    for name, (scn, seeds) in scns.items():
        for label, k, sg in variants:
            rows = [episode(scn, sd, k, sg) for sd in seeds]
            out['%s | %s' % (name, label)] = rows
            print('%-10s %-34s reward %s  arrived %s  teleported %s' % (
                name, label, ' / '.join('%.1f' % r['reward'] for r in rows), ' / '.join(str(r['arrived']) for r in rows),
                ' / '.join(str(r['teleported']) for r in rows)), flush=True)
    json.dump(dict(anchors=dict(large_grid=-972.28, real_net=-41.8), results=out),
              open(os.path.join(ROOT, 'profiles', 'r06_krauss_sweep.json'), 'w'), indent=1)
