"""Reads a bench.py JSON line on stdin and prints value, iteration time and the per-launch averages of the largest kernels
(A/B runs on the GPU box: `TSC_ENV_KF=3 python bench.py --no-extra --no-cpu-baseline | python tools/bench_brief.py kf3`)."""
import json
import sys

d = json.loads(sys.stdin.read().strip().splitlines()[-1])
ks = d.get('kernels', {})
print(' '.join(sys.argv[1:]), '%.1f M' % (d['value'] / 1e6), '%.2f ms' % d['ms_per_step'],
      ' '.join('%s=%.1fus' % (k, v['ms_total'] / v['launches'] * 1e3) for k, v in list(ks.items())[:5]))
