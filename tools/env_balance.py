#!/usr/bin/env python
"""Does the assignment of env instances to workgroups matter for tsc_env_step?  (the launch lasts as long as its
heaviest workgroup.)  Warm the instances up to mid-episode traffic, then time the step under different block orders built from the
per-instance vehicle counts, and report where the workgroups land (XCC / CU from HW_ID) and how their durations relate to the load.

    python tools/env_balance.py [E] [steps]"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np      # noqa: E402
import torch            # noqa: E402

from deeprl_signal_control_amd import _lib                      # noqa: E402
from deeprl_signal_control_amd.env import VecTrafficEnv         # noqa: E402
from deeprl_signal_control_amd.scenario import build_large_grid  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 120
scn = build_large_grid('ma2c')
env = VecTrafficEnv(scn, E, seed=12)
env.reset()
g = torch.Generator(device='cuda'); g.manual_seed(0)
acts = [torch.randint(0, 5, (E, 25), generator=g, device='cuda', dtype=torch.int32) for _ in range(16)]
for i in range(300):
    env.step(acts[i % 16])
torch.cuda.synchronize()


def counts():
    c = np.zeros(E, np.int32)
    _lib.check(env._L.tsc_env_vehicle_counts(env._h, c.ctypes.data_as(C.c_void_p)))
    return c


def timed(n):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i in range(5):
        env.step(acts[i % 16])
    a.record()
    for i in range(n):
        env.step(acts[i % 16])
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def set_order(o):
    o = None if o is None else np.ascontiguousarray(o, np.int32)
    _lib.check(env._L.tsc_env_set_block_order(env._h, None if o is None else o.ctypes.data_as(C.c_void_p)))


def blocks():
    buf = (C.c_int64 * (64 + 2 * E))()
    _lib.check(env._L.tsc_env_debug_clock(env._h, 2, None))
    for i in range(3):
        env.step(acts[i])
    torch.cuda.synchronize()
    _lib.check(env._L.tsc_env_debug_clock(env._h, 2, buf))
    w = np.array(buf[64:], dtype=np.int64).reshape(E, 2)
    tag = (w[:, 1] >> 48) & 0xFFFF
    w &= 0xFFFFFFFFFFFF
    return w, tag


out = {'E': E}
c = counts()
out['vehicles'] = {'mean': float(c.mean()), 'min': int(c.min()), 'p50': float(np.median(c)), 'max': int(c.max())}
w, tag = blocks()
dur = (w[:, 1] - w[:, 0]) / 100.0
xcc, hwid = tag >> 12, tag & 0xFFF
cu = (hwid >> 8) & 0xF          # HW_ID: wave 3:0 simd 5:4 pipe 7:6 cu 11:8 sh 12 se 15:13 (only the low 12 bits were kept)
out['identity'] = {'span_us': float((w[:, 1].max() - w[:, 0].min()) / 100.0), 'dur_min_p50_max': [float(dur.min()), float(np.median(dur)), float(dur.max())],
                   'corr_dur_vehicles': float(np.corrcoef(dur, c)[0, 1])}
# which blocks share (xcc, hwid >> 4)?  print the block ids of a few groups
groups = {}
for b in range(E):
    groups.setdefault((int(xcc[b]), int(hwid[b]) >> 6), []).append(b)
out['placement_groups'] = len(groups)
out['placement_examples'] = [v for _, v in sorted(groups.items())[:6]]
out['b_mod8_is_xcc'] = bool(all(int(xcc[b]) == int(xcc[b % 8]) for b in range(E)))

# per-workgroup cycles by phase kind: what makes the slow workgroups slow?
buf = (C.c_int64 * (64 + 7 * E))()
_lib.check(env._L.tsc_env_debug_clock(env._h, 3, None))
for i in range(3):
    env.step(acts[i])
torch.cuda.synchronize()
_lib.check(env._L.tsc_env_debug_clock(env._h, 3, buf))
ph = np.array(buf[64 + 2 * E:], dtype=np.int64).reshape(E, 5)
tot = ph.sum(1)
order_ = np.argsort(tot)
names = ['pro+epilogue', 'head walk', 'flat phase', 'gather', 'barriers']
sl = {'fastest 10 %': order_[:E // 10], 'median 10 %': order_[E // 2 - E // 20:E // 2 + E // 20], 'slowest 10 %': order_[-E // 10:]}
out['phase_cycles'] = {k: {n_: float(ph[idx, j].mean()) for j, n_ in enumerate(names)} | {'total': float(tot[idx].mean()), 'vehicles': float(counts()[idx].mean())}
                       for k, idx in sl.items()}
out['phase_corr_with_total'] = {n_: float(np.corrcoef(ph[:, j], tot)[0, 1]) for j, n_ in enumerate(names)}

rank = np.argsort(-c, kind='stable')            # heaviest first
res = {}
res['identity'] = timed(steps)
set_order(rank); res['heaviest_first'] = timed(steps)
set_order(rank[::-1].copy()); res['lightest_first'] = timed(steps)
# snake over S slots-of-a-CU assuming blocks b, b + E/4, b + 2E/4, b + 3E/4 share a CU
q = E // 4
snake = np.zeros(E, np.int64)
snake[0:q] = rank[0:q]; snake[q:2 * q] = rank[q:2 * q][::-1]; snake[2 * q:3 * q] = rank[2 * q:3 * q]; snake[3 * q:] = rank[3 * q:][::-1]
c = counts(); rank = np.argsort(-c, kind='stable')
set_order(snake); res['snake_quarters'] = timed(steps)
# snake assuming 4 consecutive blocks of one XCC share a CU: blocks 8k + x for k in 4j .. 4j + 3
o = np.zeros(E, np.int64)
grp = [[8 * (4 * j + i) + x for i in range(4)] for x in range(8) for j in range(E // 32)]
nG = len(grp)
for gi, blks in enumerate(grp):
    picks = [rank[gi], rank[2 * nG - 1 - gi], rank[2 * nG + gi], rank[4 * nG - 1 - gi]]
    for bb, inst in zip(blks, picks):
        o[bb] = inst
set_order(o); res['snake_consecutive_on_xcc'] = timed(steps)
rng = np.random.RandomState(0)
set_order(rng.permutation(E)); res['random'] = timed(steps)
set_order(None); res['identity_again'] = timed(steps)
out['us_per_step'] = res
print(json.dumps(out))
print(json.dumps(out['phase_cycles'], indent=1), file=sys.stderr)
env.close()
