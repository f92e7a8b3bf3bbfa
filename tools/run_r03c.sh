timeout 1500 tools/arena_probe3 > gpurun_out/r03c_arena_probe3.log 2> gpurun_out/r03c_arena_probe3.err
tail -5 gpurun_out/r03c_arena_probe3.log; tail -3 gpurun_out/r03c_arena_probe3.err
