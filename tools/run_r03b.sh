timeout 1500 tools/arena_probe2 base pool sizes steer > gpurun_out/r03b_arena_probe2.log 2> gpurun_out/r03b_arena_probe2.err
tail -5 gpurun_out/r03b_arena_probe2.log; tail -3 gpurun_out/r03b_arena_probe2.err
