#!/usr/bin/env python
"""Round-4 experiment behind MICROSIM_SPEC.md ("junction interiors"): does a junction-blocking mechanism move the greedy
large_grid run towards the authors' SUMO figure (-972.28, result_plot.ipynb:188) while Monaco stays in its band (-41.8)?

The mechanism lives in the CPU oracle only (oracle/microsim.c ms_set_box, OFF by default and not part of the spec): a lane head
with an open signal and no room in its target lane stands IN the junction with probability p (drawn once per vehicle) and
blocks every other approach of that node for as long as it stands there.  One greedy episode per (scenario, p, seed).

    python tools/sweep_junction_box.py            -> table on stdout + profiles/r04_junction_box_sweep.json
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


def episode(scn, seed, p):
    env = OracleEnv(scn, seed=seed, train_mode=False, test_seeds=(seed,))
    ob = env.reset(0)
    env.ms.set_box(p)
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
    return dict(reward=float(np.mean(rs)), arrived=int(tot['arrived']), departed=int(tot['departed']), pending=int(tot['pending']),
                teleported=int(tot['teleported']), box_seconds=int(env.ms.L.ms_box_count(env.ms.h)))


if __name__ == '__main__':
    out = {}
    scns = {'large_grid': build_large_grid('greedy', norm_wave=1.0, norm_wait=1.0, clip_wave=-1.0, clip_wait=-1.0),
            'real_net': build_real_net('greedy', norm_wave=1.0, clip_wave=-1.0)}
    ps = [0.0, 0.02, 0.05, 0.1, 0.2, 0.5, 1.0]
    for name, scn in scns.items():
        for p in ps:
            rows = [episode(scn, sd, p) for sd in (10000, 20000)]
            out['%s p=%g' % (name, p)] = rows
            print('%-10s p=%-5g reward %s  arrived %s  teleported %s  box-seconds %s' % (
                name, p, ' / '.join('%.1f' % r['reward'] for r in rows), ' / '.join(str(r['arrived']) for r in rows),
                ' / '.join(str(r['teleported']) for r in rows), ' / '.join(str(r['box_seconds']) for r in rows)), flush=True)
    json.dump(dict(anchors=dict(large_grid=-972.28, real_net=-41.8), seeds=[10000, 20000], results=out),
              open(os.path.join(ROOT, 'profiles', 'r04_junction_box_sweep.json'), 'w'), indent=1)
