#!/usr/bin/env python
"""Fold rocprofv3 --pmc passes (rocpd SQLite databases, one counter set per pass) into the per-kernel
HBM traffic table bench.py reads (profiles/rNN_pmc.json) and, for an SQ pass, a utilisation CSV.

    python tools/rocpd_pmc.py traffic gpurun_out/pmc_fetch/*.db gpurun_out/pmc_write/*.db profiles/rNN_pmc.json
    python tools/rocpd_pmc.py sq gpurun_out/pmc_sq/*.db profiles/r01_pmc_sq.csv

FETCH_SIZE / WRITE_SIZE are rocprofv3's derived counters (KB per dispatch, from the L2's memory-side request
counters, MI355X_MICROARCH.md "HBM / rocprofv3"); they are collected in separate passes from any trace.
"""
import csv
import json
import sqlite3
import sys

NAMES = [('step_kernel', 'env_step'), ('iql_fused_grad', 'iql_grad'), ('iql_fused_act', 'iql_act'), ('iql_fused_reduce', 'iql_reduce'), ('fc_bwd_reduce', 'dw1_gemm'), ('fc_bwd_kernel', 'dx1_gemm'), ('policy_fwd_fc_mfma', 'policy_fwd_fused'), ('head_bwd2_reduce', 'dwo_gemm'), ('head_bwd2', 'head_bwd'), ('register_order', 'register_order'),
         ('grad_norm_fold', 'grad_norm'), ('policy_fwd_fused', 'policy_fwd_fused'), ('policy_fwd_ws', 'policy_fwd_fused'), ('dwxh_kernel', 'dwx_gemm'),
         ('dx1w1_kernel', 'dx1_gemm'), ('lstm_bwd', 'lstm_bwd'), ('lstm_fwd', 'lstm_fwd'), ('head_bwd', 'head_bwd'),
         ('head_fwd', 'head_fwd'), ('add_transition', 'add_transition'), ('dwxh_reduce', 'dwh_gemm'),
         ('dx1w1_reduce', 'dw1_gemm'), ('returns_kernel', 'returns'), ('rmsprop', 'rmsprop'), ('grad_norm', 'grad_norm'),
         ('interleave_gates', 'interleave_gates'), ('transpose_wx', 'transpose_wx'),
         ('gemm_grouped_kernel<true', 'gemm_tn'), ('gemm_grouped_kernel<false', 'gemm_nn'), ('splitk_reduce', 'splitk_reduce')]


def short(name):
    for key, s in NAMES:
        if key in name:
            return s
    return None


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    out = {}
    for name, val in c.execute('select kernel_name, value from counters_collection where counter_name = ?', (counter,)):
        k = short(name)
        if k is None:
            continue
        tot, n = out.get(k, (0.0, 0))
        out[k] = (tot + float(val), n + 1)
    return out


def traffic(fetch_db, write_db, out_path, bench_line=None, command=None):
    """bench_line: the JSON line bench.py printed under the FETCH pass -- its window-mean vehicles per instance and its
    workload are recorded next to the counters (bench.py:pmc_traffic compares that figure with its own run's)."""
    f, w = per_kernel(fetch_db, 'FETCH_SIZE'), per_kernel(write_db, 'WRITE_SIZE')
    kern = {}
    for k in sorted(set(f) | set(w)):
        ft, fn = f.get(k, (0.0, 0))
        wt, wn = w.get(k, (0.0, 0))
        kern[k] = {'fetch_kb': ft / max(fn, 1), 'write_kb': wt / max(wn, 1), 'launches': max(fn, wn)}
    doc = {'command': 'tools/profile_round.sh: rocprofv3 --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) -- %s; folded by '
                      'tools/rocpd_pmc.py' % (command or 'python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra --no-profile'),
           'units': 'KB per launch (average over launches)', 'kernels': kern}
    if bench_line:
        line = [l for l in open(bench_line) if l.startswith('{')][-1]
        b = json.loads(line)
        doc['mean_live_vehicles_per_env'] = b['config']['mean_live_vehicles_per_env']
        doc['live_vehicles'] = ('window mean over the timed iterations of the FETCH pass (one whole episode; the warm-up iterations are '
                                'another whole episode with the same statistics, and the counters average over both)')
        doc['workload'] = b['config']['workload']
    json.dump(doc, open(out_path, 'w'), indent=1)
    print('%d kernels -> %s' % (len(kern), out_path))


def sq(db, out_path):
    c = sqlite3.connect(db)
    names = [r[0] for r in c.execute('select distinct counter_name from counters_collection')]
    agg = {}
    for name, cn, val in c.execute('select kernel_name, counter_name, value from counters_collection'):
        k = short(name)
        if k is None:
            continue
        d = agg.setdefault(k, {})
        d[cn] = d.get(cn, 0.0) + float(val)
        d['_n_' + cn] = d.get('_n_' + cn, 0) + 1
    with open(out_path, 'w', newline='') as fh:
        w = csv.writer(fh)
        w.writerow(['kernel', 'dispatches'] + names + ['wait_any_frac', 'active_inst_frac', 'mfma_busy_per_simd_frac'])
        for k, d in sorted(agg.items()):
            wc = d.get('SQ_WAVE_CYCLES', 0.0)
            row = [k, d.get('_n_' + names[0], 0)] + ['%.0f' % d.get(n, 0.0) for n in names]
            row.append('%.3f' % (d.get('SQ_WAIT_ANY', 0.0) / wc) if wc else '')
            row.append('%.3f' % (d.get('SQ_ACTIVE_INST_ANY', 0.0) / wc) if wc else '')
            ga = d.get('GRBM_GUI_ACTIVE', 0.0)
            row.append('%.3f' % (d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (128.0 * ga)) if ga else '')
            w.writerow(row)
    print('%d kernels -> %s' % (len(agg), out_path))


if __name__ == '__main__':
    if sys.argv[1] == 'traffic':
        traffic(sys.argv[2], sys.argv[3], sys.argv[4], *(sys.argv[5:7]))
    else:
        sq(sys.argv[2], sys.argv[3])
