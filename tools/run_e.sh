python -m pytest tests -m gpu -q > gpurun_out/r3_tests_e.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_tests_e.log; tail -6 gpurun_out/r3_tests_e.log
python bench.py --no-cpu-baseline --no-roofline > gpurun_out/r3_bench_e.json 2> gpurun_out/r3_bench_e.err; grep "host enqueue" gpurun_out/r3_bench_e.err
python bench.py --workload cfg4 --no-cpu-baseline --no-roofline > gpurun_out/r3_bench_e_cfg4.json 2> gpurun_out/r3_bench_e_cfg4.err; grep "host enqueue" gpurun_out/r3_bench_e_cfg4.err
python bench.py --workload cfg4 --step-impl autograd --no-cpu-baseline --no-roofline > gpurun_out/r3_bench_e_cfg4a.json 2> gpurun_out/r3_bench_e_cfg4a.err; grep "host enqueue" gpurun_out/r3_bench_e_cfg4a.err
python bench.py --workload cfg5 --no-cpu-baseline --no-roofline --steps 10 --warmup 3 > gpurun_out/r3_bench_e_cfg5.json 2> gpurun_out/r3_bench_e_cfg5.err; grep "host enqueue" gpurun_out/r3_bench_e_cfg5.err
