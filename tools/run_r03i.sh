python -m pytest tests -m gpu -x -q > gpurun_out/r03i_pytest_gpu.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03i_pytest_gpu.log
tail -3 gpurun_out/r03i_pytest_gpu.log
timeout 300 python tools/prof_pipeline_windows.py 2 64 > gpurun_out/r03i_prof_pipeline_windows.log 2>&1
grep -v first_windows gpurun_out/r03i_prof_pipeline_windows.log | cut -c1-300; grep rep gpurun_out/r03i_prof_pipeline_windows.log | cut -c1-420
timeout 300 python tools/bench_pipeline.py 2 > gpurun_out/r03i_bench_pipeline.jsonl 2>&1; cat gpurun_out/r03i_bench_pipeline.jsonl | cut -c1-300
BB_EXPERIMENTS=1 timeout 600 python tools/exp_lds.py > gpurun_out/r03i_exp_lds.log 2>&1; cat gpurun_out/r03i_exp_lds.log | cut -c1-400
(time python bench.py) > gpurun_out/r03i_bench.json 2> gpurun_out/r03i_bench.err; tail -5 gpurun_out/r03i_bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03i_bench.json') if l.startswith('{')][-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d['roofline']['kernel'])
print(json.dumps(d.get('mid_size'))[:3000])
print([ (c['case'][:30], c['frac']) for c in d.get('other_configs',[]) if 'case' in c])
print(d.get('cfg3',{}).get('roofline'))
PY
