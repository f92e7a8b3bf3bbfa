for i in 1 2; do
python bench.py > gpurun_out/r03o_bench_$i.json 2> gpurun_out/r03o_bench_$i.err
python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r03o_bench_$i.json') if l.startswith('{')][-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'])
m=d['mid_size']
for r in m['sizes']:
    print(r['frames'], 'torch', r['torch_empty']['frac_min'], r['torch_empty']['frac_median'], r['torch_empty']['frac_max'], 'arena', r['arena']['frac_min'], r['arena']['frac_median'], r['arena']['frac_max'], 'api', r['api_read'].get('frac'))
g=m['guppi_cf_8GiB_in']; print('guppi torch', g['torch_empty']['frac_min'], g['torch_empty']['frac_median'], 'arena', g['arena']['frac_min'], g['arena']['frac_median'])
print({k: m['arena_after'][k] for k in ('bytes_backed','steps','probes','last_probe_gbps','grow_ms')})
PY
done
timeout 400 python tools/exp_arena.py 250 arena:0 > gpurun_out/r03o_exp_arena.log 2>&1; grep '"kind": "arena\|arena_stats_at_end\|torch.empty' gpurun_out/r03o_exp_arena.log | cut -c1-400
timeout 300 python tools/exp_arena_history.py > gpurun_out/r03o_exp_arena_history.log 2>&1; cat gpurun_out/r03o_exp_arena_history.log | cut -c1-200
python -m pytest tests/test_arena_gpu.py -m gpu -q 2>&1 | tail -3
