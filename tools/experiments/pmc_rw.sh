#!/bin/bash
# usage (GPU box, repo root): bash tools/experiments/pmc_rw.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_rw
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for set in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum" \
           "TCC_BUSY_sum TCC_REQ_sum TCC_EA0_WRREQ_IO_CREDIT_STALL_sum TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set -d $OUT/p$i -o p --output-format csv -- python3 $R/tools/experiments/pmc_rw.py > $OUT/p$i.out 2> $OUT/p$i.err
  tail -2 $OUT/p$i.err
done
find $OUT -type f ! -name '*.csv' ! -name '*.out' ! -name '*.err' -delete
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if 'k_decode_flat_lds' in r['Kernel_Name']]
    by = collections.defaultdict(dict)
    for r in rows:
        by[r['Dispatch_Id']][r['Counter_Name']] = float(r['Counter_Value'])
    print(f.split('/')[-3])
    for k in sorted(by, key=int):
        print('  dispatch', k, {n: '%.4g' % v for n, v in sorted(by[k].items())})
PY
