#!/bin/bash
# Bench lines + rocprofv3 kernel traces of the other BASELINE configs (c2: IA2C FC 256 envs, c5: Monaco MA2C 512 envs),
# on the GPU box (run through gpurun from the repo root):   tools/profile_configs.sh r04
set -u
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in c2 c5; do
  python $ROOT/bench.py --config $C > $OUT/${TAG}_bench_${C}.json 2> $OUT/${TAG}_bench_${C}.err
  rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_trace_${C} -o ${TAG}${C} -- python $ROOT/bench.py --config $C --no-cpu-baseline --no-extra --steps 3 --warmup 1 --no-profile > /dev/null 2> $OUT/${TAG}_trace_${C}.err
done
find $OUT -name "*.db" | xargs ls -la
