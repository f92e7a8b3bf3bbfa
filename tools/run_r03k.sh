timeout 600 python tools/exp_arena_history.py > gpurun_out/r03k_exp_arena_history.log 2>&1; cat gpurun_out/r03k_exp_arena_history.log | cut -c1-250
