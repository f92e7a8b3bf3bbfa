export TMPDIR=/tmp
R=$(pwd); OUT=$R/gpurun_out/r03r; mkdir -p $OUT; cd /tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
grep -n -A6 "TCC_EA0_WRREQ\b\|TCC_EA0_WRREQ$" $OUT/counters_list.txt | head -40
python3 $R/tools/prof_arena_blocks.py > $OUT/plain.log 2>&1; cat $OUT/plain.log | tail -4
run() { local name=$1; shift; rm -rf $OUT/$name; timeout 300 rocprofv3 "$@" -d $OUT/$name -o p --output-format csv -- python3 $R/tools/prof_arena_blocks.py > $OUT/$name.log 2>&1; find $OUT/$name -type f ! -name '*.csv' -delete; tail -4 $OUT/$name.log | cut -c1-120; }
run wr --pmc TCC_EA0_WRREQ TCC_EA0_WRREQ_DRAM_CREDIT_STALL
run wr2 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum
run rd --pmc TCC_EA0_RDREQ TCC_EA0_RDREQ_DRAM_CREDIT_STALL
cd $R
f=$(find $OUT/wr -name "*counter_collection.csv" | head -1); head -3 $f | cut -c1-400; grep -c k_decode $f
