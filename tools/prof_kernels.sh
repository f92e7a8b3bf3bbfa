#!/bin/bash
# Per-kernel profiles of every decode kernel family:
# kernel stats, FETCH_SIZE, WRITE_SIZE and LDS / wait counters in separate
# rocprofv3 passes, program directly after `--`.
# usage (GPU box, repo root): bash tools/prof_kernels.sh <tag> [gib]
set -u
TAG=${1:-rXX}
GIB=${2:-4}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
run() {   # name, rocprof args...
  local name=$1; shift
  rm -rf $OUT/$name
  timeout 900 rocprofv3 "$@" -d $OUT/$name -o p --output-format csv -- python3 $R/tools/prof_formats.py $GIB 3 > $OUT/$name.log 2>&1
  find $OUT/$name -type f ! -name '*.csv' -delete
}
run stats --kernel-trace --stats
find $OUT/stats -name '*kernel_trace.csv' -delete
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
run sq --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
cd $R
python3 tools/summarize_kernels.py $TAG $OUT
