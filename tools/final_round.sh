# end of round: the -m gpu suite, then every profile of the round with the final library (one gpurun call)
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/final_tests.log 2>&1 || { tail -15 gpurun_out/final_tests.log; exit 1; }
tail -1 gpurun_out/final_tests.log
bash tools/prof_r02.sh r02_h adc_mfma_kernel > gpurun_out/prof_r02_h.log 2>&1 || exit 1
bash tools/prof_r02_rest.sh r02_h > gpurun_out/prof_r02_h_rest.log 2>&1 || exit 1
timeout -k 10 300 python bench.py --force-sharded --no-secondary > gpurun_out/r02_h_rest/r02_h_bench_sharded_world1.json 2> gpurun_out/shard.err || exit 1
grep -c "P=" gpurun_out/r02_h_rest/sweep_10m.log gpurun_out/r02_h_rest/sweep_c5.log
