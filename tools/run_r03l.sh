timeout 400 python tools/exp_arena.py 250 arena_step48:0 > gpurun_out/r03l_exp_arena_step48.log 2>&1
BB_ARENA_STEP_GIB=8 BB_ARENA_KEEP_EVERY=6 timeout 400 python tools/exp_arena.py 250 arena_step8_of_48:0 > gpurun_out/r03l_exp_arena_step8_keep6.log 2>&1
BB_ARENA_STEP_GIB=8 timeout 400 python tools/exp_arena.py 250 arena_step8:0 > gpurun_out/r03l_exp_arena_step8.log 2>&1
BB_ARENA_STEP_GIB=16 BB_ARENA_KEEP_EVERY=6 timeout 400 python tools/exp_arena.py 250 arena_step16_of_96:0 > gpurun_out/r03l_exp_arena_step16_keep6.log 2>&1
BB_ARENA_STEP_GIB=24 timeout 400 python tools/exp_arena.py 250 arena_step24:0 > gpurun_out/r03l_exp_arena_step24.log 2>&1
for f in gpurun_out/r03l_exp_arena_*.log; do echo == $f; grep '"kind": "arena\|torch.empty' $f | cut -c1-300; done
python -m pytest tests/test_arena_gpu.py tests/test_bytefmt_gpu.py -m gpu -q --maxfail=5 2>&1 | tail -5
timeout 300 python tools/prof_pipeline_windows.py 2 64 > gpurun_out/r03l_prof_pipeline_windows.log 2>&1
grep rep gpurun_out/r03l_prof_pipeline_windows.log | cut -c1-300
timeout 300 python tools/bench_subset_blocks.py > gpurun_out/r03l_bench_subset_blocks.jsonl 2>&1; cat gpurun_out/r03l_bench_subset_blocks.jsonl | cut -c1-330
