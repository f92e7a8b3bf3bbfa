#!/bin/bash
# counter passes over tools/experiments/prof_alloc.py (program directly after `--`)
set -u
TAG=${1:-r02af}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
python3 $R/tools/experiments/prof_alloc.py > $OUT/plain0.log 2>&1      # (first process on a fresh box places differently)
python3 $R/tools/experiments/prof_alloc.py > $OUT/plain1.log 2>&1
run() {
  local name=$1; shift
  rm -rf $OUT/$name
  # (bounded: a GRBM_* pass once aborted inside rocprofv3 and sat in its finaliser for 40 minutes)
  timeout 300 rocprofv3 "$@" -d $OUT/$name -o p --output-format csv -- python3 $R/tools/experiments/prof_alloc.py > $OUT/$name.log 2>&1
  find $OUT/$name -type f ! -name '*.csv' -delete
}
run trace --kernel-trace
run utcl --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_PENDING_STALL_CYCLES_sum
run eawr --pmc TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_LEVEL_sum
run tcc --pmc TCC_BUSY_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_LEVEL_sum
run lat --pmc TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum
python3 $R/tools/experiments/prof_alloc.py > $OUT/plain2.log 2>&1
cd $R
python3 - <<PY
import csv, glob, json, collections
out = "$OUT"
for name in ("plain0", "plain1", "plain2", "trace", "utcl", "eawr", "tcc", "lat"):
    print("==", name)
    for l in open(out + "/" + name + ".log"):
        if l.startswith("{"): print("  ", l.strip()[:200])
for name in ("utcl", "eawr", "tcc", "lat"):
    files = glob.glob(out + "/" + name + "/**/*counter_collection.csv", recursive=True)
    if not files:
        print(name, "no csv"); continue
    rows = [r for r in csv.DictReader(open(files[0])) if "k_decode_flat" in r["Kernel_Name"]]
    per = collections.OrderedDict()
    for r in rows:
        per.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
    print("==", name, "per decode dispatch (in launch order, 3 per size)")
    for d, v in per.items():
        print("  ", d, {k: "%.4g" % x for k, x in v.items()})
PY
