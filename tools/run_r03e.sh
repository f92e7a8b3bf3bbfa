timeout 900 tools/arena_probe5 > gpurun_out/r03e_arena_probe5.log 2> gpurun_out/r03e_arena_probe5.err
tail -2 gpurun_out/r03e_arena_probe5.log | cut -c1-300; tail -3 gpurun_out/r03e_arena_probe5.err
