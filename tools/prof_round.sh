#!/bin/bash
# Profiles bench.py the way the bench contract asks: one --kernel-trace --stats
# pass over the whole default run (headline + api_read + cfg3 + other_configs),
# then FETCH_SIZE and WRITE_SIZE in separate --pmc passes of the headline leg
# (program directly after `--`), then the plain bench line.
# usage (GPU box, from the repo root): bash tools/prof_round.sh [tag]
set -u
TAG=${1:-prof}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rm -rf $OUT/prof_stats $OUT/prof_head $OUT/prof_fetch $OUT/prof_write
# headline leg alone: the decode kernel's row of this pass holds headline launches only
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof_head -o hd --output-format csv -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --traffic none --no-extra-legs --detail $OUT/bench_headline_under_rocprof_detail.json > $OUT/bench_headline_under_rocprof.json 2> $OUT/prof_head.err
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -o st --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --traffic none --detail $OUT/bench_under_rocprof_detail.json > $OUT/bench_under_rocprof.json 2> $OUT/prof_stats.err
timeout 900 rocprofv3 --pmc FETCH_SIZE -d $OUT/prof_fetch -o f --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --pmc-child > /dev/null 2> $OUT/prof_fetch.err
timeout 900 rocprofv3 --pmc WRITE_SIZE -d $OUT/prof_write -o w --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --pmc-child > /dev/null 2> $OUT/prof_write.err
# keep only the small CSVs (gpurun_out is capped)
find $OUT/prof_stats $OUT/prof_head $OUT/prof_fetch $OUT/prof_write -type f ! -name '*.csv' -delete
find $OUT/prof_stats $OUT/prof_head -name '*kernel_trace.csv' -delete
cd $R
# the driver's command; the compact line (stdout) and the full record of the legs
python3 bench.py --gpus 1 --steps 20 --warmup 5 --detail $OUT/bench_detail.json > $OUT/bench_plain.json 2> $OUT/bench_plain.err
echo "rc $? bytes of the last stdout line: $(tail -n 1 $OUT/bench_plain.json | wc -c)"
tail -n 1 $OUT/bench_plain.json
