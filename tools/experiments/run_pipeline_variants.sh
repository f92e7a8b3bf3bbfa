#!/bin/bash
# bench.py's pipeline leg under the round-5 switches, twice each (a noisy host):
# does the background file sink or the arena's background growth change the READ rates?
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
for v in "BB_WRITE_ASYNC=1 BB_ARENA_PREPARE=1" "BB_WRITE_ASYNC=0 BB_ARENA_PREPARE=1" "BB_WRITE_ASYNC=1 BB_ARENA_PREPARE=0" "BB_WRITE_ASYNC=0 BB_ARENA_PREPARE=0"; do
  echo "## $v (rep $rep)"
  env $v python3 $R/tools/run_pipeline_leg.py 2.0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
for f in d['formats']:
    print('  %-52s read best %.1f median %.1f GB/s   writer %.2f GB/s' % (f['case'][:52], f.get('file_GBps_best',0), f.get('file_GBps_median',0), f.get('writer_GBps',0)))
"
done
done
