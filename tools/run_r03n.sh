python -m pytest tests -m gpu -q --maxfail=8 > gpurun_out/r03n_pytest_gpu.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03n_pytest_gpu.log
tail -4 gpurun_out/r03n_pytest_gpu.log
BB_EXPERIMENTS=1 timeout 900 python tools/exp_lds.py > gpurun_out/r03n_exp_lds.log 2>&1; cat gpurun_out/r03n_exp_lds.log | cut -c1-330
bash tools/prof_round.sh r03n > gpurun_out/r03n_prof_round.log 2>&1
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03n/bench_plain.json') if l.startswith('{')][-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d['roofline']['kernel'], d['roofline'].get('traffic'))
for r in d['mid_size']['sizes']:
    print(r['frames'], 'torch', r['torch_empty']['frac_min'], r['torch_empty']['frac_median'], r['torch_empty']['frac_max'], 'arena', r['arena']['frac_min'], r['arena']['frac_median'], r['arena']['frac_max'], 'api', r['api_read'].get('frac'))
print(d['mid_size']['guppi_cf_8GiB_in'])
print([ (c['case'][:30], c['frac']) for c in d.get('other_configs',[]) if 'case' in c])
print(d.get('cfg3',{}).get('roofline'))
PY
