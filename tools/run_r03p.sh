bash tools/prof_kernels.sh r03pk 8 > gpurun_out/r03pk_prof_kernels.log 2>&1; tail -5 gpurun_out/r03pk_prof_kernels.log
ls gpurun_out/r03pk | head; ls profiles | grep r03pk
