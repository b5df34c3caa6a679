#!/bin/bash
# Profiles of one round, on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r04
# 1. rocprofv3 --kernel-trace --stats of `python bench.py` (every launch timed by the profiler)
# 2. PMC passes, each in its own run without any trace (MI355X_MICROARCH.md "HBM / rocprofv3"): FETCH_SIZE, WRITE_SIZE, SQ
# Output: rocpd databases under gpurun_out/<tag>_*; fold them with tools/rocpd_stats.py / tools/rocpd_pmc.py into profiles/.
set -u
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $ROOT/bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err       # the driver line (c3 + extra.configs c2 / c5 + cpu_baseline)
B="python $ROOT/bench.py --no-cpu-baseline --no-extra"
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_trace -o ${TAG} -- $B --steps 3 --warmup 1 --no-profile > $OUT/${TAG}_bench_rocprof.json 2> $OUT/${TAG}_trace.err
rocprofv3 --pmc FETCH_SIZE -d $OUT/${TAG}_pmc_fetch -o f -- $B --steps 1 --warmup 1 --no-profile > /dev/null 2> $OUT/${TAG}_pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE -d $OUT/${TAG}_pmc_write -o w -- $B --steps 1 --warmup 1 --no-profile > /dev/null 2> $OUT/${TAG}_pmc_write.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/${TAG}_pmc_sq -o s -- $B --steps 1 --warmup 1 --no-profile > /dev/null 2> $OUT/${TAG}_pmc_sq.err
find $OUT -name "*.db" | xargs ls -la
