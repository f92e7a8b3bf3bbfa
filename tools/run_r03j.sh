python -m pytest tests -m gpu -q --maxfail=8 > gpurun_out/r03j_pytest_gpu.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03j_pytest_gpu.log
tail -12 gpurun_out/r03j_pytest_gpu.log
BB_EXPERIMENTS=1 python -m pytest tests/test_kernels_gpu.py tests/test_abi.py tests/test_bounds_gpu.py -q --maxfail=5 > gpurun_out/r03j_pytest_exp.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03j_pytest_exp.log
tail -3 gpurun_out/r03j_pytest_exp.log
timeout 300 python tools/prof_pipeline_windows.py 2 64 > gpurun_out/r03j_prof_pipeline_windows.log 2>&1
grep rep gpurun_out/r03j_prof_pipeline_windows.log | cut -c1-330
timeout 300 python tools/bench_pipeline.py 2 > gpurun_out/r03j_bench_pipeline.jsonl 2>&1; cat gpurun_out/r03j_bench_pipeline.jsonl | cut -c1-300
BB_EXPERIMENTS=1 timeout 900 python tools/exp_lds.py > gpurun_out/r03j_exp_lds.log 2>&1; cat gpurun_out/r03j_exp_lds.log | cut -c1-420
