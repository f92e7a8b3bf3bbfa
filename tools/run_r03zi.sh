# translation counters for a slow and a fast arena block (tools/prof_arena_blocks.py)
export TMPDIR=/tmp
R=$(pwd); OUT=$R/gpurun_out/r03zi; mkdir -p $OUT; cd /tmp
python3 $R/tools/prof_arena_blocks.py > $OUT/plain.log 2>&1; tail -4 $OUT/plain.log
run() { local name=$1; shift; rm -rf $OUT/$name; timeout 300 rocprofv3 "$@" -d $OUT/$name -o p --output-format csv -- python3 $R/tools/prof_arena_blocks.py > $OUT/$name.log 2>&1; find $OUT/$name -type f ! -name '*.csv' -delete; tail -4 $OUT/$name.log | cut -c1-120; }
run t1 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum
run t2 --pmc GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE
run t3 --pmc TCP_UTCL1_SERIALIZATION_STALL TCP_UTCL1_THRASHING_STALL TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS
cd $R
python3 - <<PY
import csv, glob, collections
for name in ("t1", "t2", "t3"):
    f = glob.glob("gpurun_out/r03zi/%s/**/*counter_collection.csv" % name, recursive=True)
    if not f:
        print(name, "no csv"); continue
    rows = [r for r in csv.DictReader(open(f[0])) if "k_decode_flat_lds" in r["Kernel_Name"]]
    by = collections.defaultdict(list)
    for r in rows:
        by[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for c, v in by.items():
        d = collections.defaultdict(float)
        for i, x in v: d[i] += x
        print(name, c, [round(d[k]) for k in sorted(d)])
PY
