#!/usr/bin/env python
"""Round-5 experiment behind MICROSIM_SPEC.md ("lane changing"): does lane choice on large_grid's two-lane streets -- hand-offs
that enter the lane the junction's connection leads to, and a gap-acceptance lane change inside the 200-m edge -- move the greedy
large_grid run from this spec's -66 towards the authors' -972 (result_plot.ipynb:188)?  CPU oracle only (oracle/microsim.c,
ms_set_lanechange: off by default, the spec is unchanged).  One greedy episode per (gaps, seed).

    python tools/sweep_lane_change.py            -> profiles/r05_lane_change_sweep.json
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deeprl_signal_control_amd.scenario import build_large_grid                      # noqa: E402
from oracle.env_oracle import OracleEnv, greedy_large_grid                           # noqa: E402


def episode(scn, seed, gaps):
    env = OracleEnv(scn, seed=seed, train_mode=False, test_seeds=(seed,))
    ob = env.reset(0)
    if gaps is not None:
        env.ms.set_lanechange(*gaps)
    rs = []
    while True:
        ob, r, done, g = env.step([greedy_large_grid(o[:6]) for o in ob])
        rs.append(g)
        if done:
            break
    tot = env.ms.totals()
    assert env.ms.check() == 0
    lc = env.ms.lanechange_counts()
    return dict(reward=float(np.mean(rs)), arrived=int(tot['arrived']), departed=int(tot['departed']), pending=int(tot['pending']),
                teleported=int(tot['teleported']), mean_trip=float(tot['sum_trip']) / max(1, tot['arrived']),
                lane_changes=int(lc['changes']), blocked_vehicle_seconds=int(lc['blocked_seconds']))


if __name__ == '__main__':
    kw = dict(norm_wave=1.0, norm_wait=1.0, clip_wave=-1.0, clip_wait=-1.0)
    scn = build_large_grid('greedy', lane_change=False, **kw)            # the rounds 1 - 4 tables: the experiment hooks sit on top of them
    scn10 = build_large_grid('greedy', lane_change=True, **kw)           # rule 10 as adopted (compiled tables + lane_sib)
    out = {}
    cases = [('rounds 1-4 (needed lane at edge entry, no lane change)', None), ('gap acceptance anywhere, gaps 0 / 0 m (any free slot)', (0.0, 0.0)),
             ('gap acceptance anywhere, gaps 2 / 2 m', (2.0, 2.0)), ('gap acceptance anywhere, gaps 2.5 / 5 m', (2.5, 5.0)),
             ('gap acceptance anywhere, gaps 5 / 10 m', (5.0, 10.0)), ('gap acceptance anywhere, gaps 10 / 20 m', (10.0, 20.0)),
             ('behind the sibling tail only, gap 0 m', (0.0, -1.0)), ('behind the sibling tail only, gap 2 m', (2.0, -1.0)),
             ('behind the sibling tail only, gap 5 m', (5.0, -1.0)), ('RULE 10 (adopted)', 'rule10')]
    for name, gaps in cases:
        if gaps == 'rule10':
            rows = [episode(scn10, sd, None) for sd in (10000, 20000, 30000, 40000)]
        else:
            rows = [episode(scn, sd, gaps) for sd in (10000, 20000)]
        out[name] = rows
        print('%-52s reward %s  arrived %s  teleported %s  trip %s  changes %s  blocked veh-s %s' % (
            name, ' / '.join('%.1f' % r['reward'] for r in rows), ' / '.join(str(r['arrived']) for r in rows),
            ' / '.join(str(r['teleported']) for r in rows), ' / '.join('%.0f' % r['mean_trip'] for r in rows),
            ' / '.join(str(r['lane_changes']) for r in rows), ' / '.join(str(r['blocked_vehicle_seconds']) for r in rows)), flush=True)
    json.dump(dict(anchor=dict(large_grid=-972.28), seeds=[10000, 20000], seeds_rule10=[10000, 20000, 30000, 40000], results=out),
              open(os.path.join(ROOT, 'profiles', 'r05_lane_change_sweep.json'), 'w'), indent=1)
