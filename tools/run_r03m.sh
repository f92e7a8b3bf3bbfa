python -m pytest tests -m gpu -q --maxfail=8 > gpurun_out/r03m_pytest_gpu.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03m_pytest_gpu.log
tail -6 gpurun_out/r03m_pytest_gpu.log
BB_EXPERIMENTS=1 python -m pytest tests/test_kernels_gpu.py tests/test_abi.py tests/test_bounds_gpu.py -q --maxfail=5 > gpurun_out/r03m_pytest_exp.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03m_pytest_exp.log
tail -3 gpurun_out/r03m_pytest_exp.log
timeout 300 python tools/prof_pipeline_windows.py 2 64 > gpurun_out/r03m_prof_pipeline_windows.log 2>&1
grep rep gpurun_out/r03m_prof_pipeline_windows.log | cut -c1-240
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/prof_round.sh r03m > gpurun_out/r03m_prof_round.log 2>&1; tail -c 400 gpurun_out/r03m_prof_round.log
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03m/bench_plain.json') if l.startswith('{')][-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d['roofline']['kernel'], d['roofline'].get('traffic'))
for r in d['mid_size']['sizes']:
    print(r['frames'], 'torch', r['torch_empty']['frac_min'], r['torch_empty']['frac_median'], r['torch_empty']['frac_max'], 'arena', r['arena']['frac_min'], r['arena']['frac_median'], r['arena']['frac_max'], 'api', r['api_read'].get('frac'))
print(d['mid_size']['guppi_cf_8GiB_in'])
print([ (c['case'][:30], c['frac']) for c in d.get('other_configs',[]) if 'case' in c])
print(d.get('cfg3',{}).get('roofline'))
print(d['api_read']['ms_over_kernel_leg'], d['cpu_baseline']['value'], d['cpu_baseline']['calibration']['reference_as_written_estimate_Msps'])
PY
