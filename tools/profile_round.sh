#!/bin/bash
# Profiles of one round, on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r06 [cal c2 c5 q1 c3]
# per configuration (c3 = the default workload; suffix "" / _c2 / _c5 / _q1):
# 1. (run last, see below) the bench line (c3: the driver line `python bench.py` with extra.configs and cpu_baseline)
# 2. rocprofv3 --kernel-trace --stats of the same command (every launch timed by the profiler; bench.py's default window = whole
#    episodes, so the per-kernel averages are episode averages like the live HIP-event figures)
# 3. PMC passes, each in its own run without any trace (MI355X_MICROARCH.md "HBM / rocprofv3"): FETCH_SIZE, WRITE_SIZE, SQ.
#    The FETCH / WRITE passes cover WHOLE EPISODES (one warm-up episode + one timed episode), so that the per-launch averages
#    belong to the episode-mean vehicle count the bench line of the pass reports (bench.py refuses a summary whose count is
#    more than 15 % off the run's).
# Output: rocpd databases under gpurun_out/<tag>_*; folded into gpurun_out/<tag>/ by tools/rocpd_stats.py / tools/rocpd_pmc.py
# (copy that directory's files into profiles/).
set -u
TAG=${1:-r06}
shift
CFGS=${@:-cal c2 c5 q1 c3}        # c3 last: its driver line also carries c2 / c5 / q1 and quotes their summaries
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
RES=$OUT/$TAG
mkdir -p $RES
cd /tmp && export TMPDIR=/tmp
for C in $CFGS; do
  if [ $C = cal ]; then
    # 0. the counters on known byte counts (tools/fetch_calib.hip: 1-GiB streams, dword and 16 B per lane), one counter per pass
    [ -x $ROOT/tools/fetch_calib ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $ROOT/tools/fetch_calib $ROOT/tools/fetch_calib.hip
    rocprofv3 --pmc FETCH_SIZE -d $OUT/${TAG}_cal_f -o f -- $ROOT/tools/fetch_calib > /dev/null 2> $OUT/${TAG}_cal_f.err
    rocprofv3 --pmc WRITE_SIZE -d $OUT/${TAG}_cal_w -o w -- $ROOT/tools/fetch_calib > /dev/null 2> $OUT/${TAG}_cal_w.err
    python $ROOT/tools/fetch_calib.py $(find $OUT/${TAG}_cal_f -name "*.db" | head -1) $(find $OUT/${TAG}_cal_w -name "*.db" | head -1) $RES/${TAG}_fetch_calibration.json
    cp $RES/${TAG}_fetch_calibration.json $ROOT/profiles/ 2>/dev/null
    rm -rf $OUT/${TAG}_cal_f $OUT/${TAG}_cal_w
    continue
  fi
  case $C in
    c3) SUF=""; EP=6;;        # iterations per episode: 720 control steps / n_step
    c2) SUF="_c2"; EP=6;;
    c5) SUF="_c5"; EP=18;;
    q1) SUF="_q1"; EP=36;;
  esac
  B="python $ROOT/bench.py --config $C --no-cpu-baseline --no-extra"
  rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_trace${SUF} -o t -- $B --no-profile > $RES/${TAG}_bench_rocprof${SUF}.json 2> $OUT/${TAG}_trace${SUF}.err
  python $ROOT/tools/rocpd_stats.py $(find $OUT/${TAG}_trace${SUF} -name "*.db" | head -1) $RES/${TAG}_kernel_stats${SUF}.csv
  rm -rf $OUT/${TAG}_trace${SUF}                 # the rocpd databases are tens of MB each; gpurun copies back at most 64 MiB
  if [ $C != q1 ]; then
    rocprofv3 --pmc FETCH_SIZE -d $OUT/${TAG}_pmc_fetch${SUF} -o f -- $B --steps $EP --warmup $EP --no-profile > $OUT/${TAG}_pmc_fetch${SUF}.json 2> $OUT/${TAG}_pmc_fetch${SUF}.err
    rocprofv3 --pmc WRITE_SIZE -d $OUT/${TAG}_pmc_write${SUF} -o w -- $B --steps $EP --warmup $EP --no-profile > $OUT/${TAG}_pmc_write${SUF}.json 2> $OUT/${TAG}_pmc_write${SUF}.err
    python $ROOT/tools/rocpd_pmc.py traffic $(find $OUT/${TAG}_pmc_fetch${SUF} -name "*.db" | head -1) $(find $OUT/${TAG}_pmc_write${SUF} -name "*.db" | head -1) \
           $RES/${TAG}_pmc${SUF}.json $OUT/${TAG}_pmc_fetch${SUF}.json "$B --steps $EP --warmup $EP --no-profile"
    rm -rf $OUT/${TAG}_pmc_fetch${SUF} $OUT/${TAG}_pmc_write${SUF}
  fi
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/${TAG}_pmc_sq${SUF} -o s -- $B --steps 1 --warmup $EP --no-profile > /dev/null 2> $OUT/${TAG}_pmc_sq${SUF}.err
  python $ROOT/tools/rocpd_pmc.py sq $(find $OUT/${TAG}_pmc_sq${SUF} -name "*.db" | head -1) $RES/${TAG}_pmc_sq${SUF}.csv
  rm -rf $OUT/${TAG}_pmc_sq${SUF}
  # the bench line LAST, after this collection's summaries replaced the box's copy of profiles/: the line quotes the profiler's
  # averages and the PMC traffic of the same collection (the files that are then committed)
  cp $RES/${TAG}_kernel_stats${SUF}.csv $ROOT/profiles/ 2>/dev/null
  [ -f $RES/${TAG}_pmc${SUF}.json ] && cp $RES/${TAG}_pmc${SUF}.json $ROOT/profiles/
  if [ $C = c3 ]; then
    python $ROOT/bench.py > $RES/${TAG}_bench.json 2> $OUT/${TAG}_bench.err           # the driver line
  else
    python $ROOT/bench.py --config $C --no-cpu-baseline > $RES/${TAG}_bench${SUF}.json 2> $OUT/${TAG}_bench${SUF}.err
  fi
done
ls -la $RES
