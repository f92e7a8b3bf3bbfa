cd tools && python exp_cai.py > ../gpurun_out/r03a_exp_cai.log 2>&1; cd ..
timeout 900 tools/arena_probe base pairs layouts > gpurun_out/r03a_arena_probe.log 2> gpurun_out/r03a_arena_probe.err
tail -3 gpurun_out/r03a_arena_probe.log; cat gpurun_out/r03a_exp_cai.log
