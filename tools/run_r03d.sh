timeout 1500 tools/arena_probe4 > gpurun_out/r03d_arena_probe4.log 2> gpurun_out/r03d_arena_probe4.err
tail -5 gpurun_out/r03d_arena_probe4.log | cut -c1-300; tail -3 gpurun_out/r03d_arena_probe4.err
