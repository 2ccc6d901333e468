python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err; tail -3 gpurun_out/r03_bench_default.err; cut -c1-300 gpurun_out/r03_bench_default.json
python bench.py --encoder resnet18 --no-cpu-baseline > gpurun_out/r03_bench_resnet18.json 2> gpurun_out/r03_bench_resnet18.err; cut -c1-200 gpurun_out/r03_bench_resnet18.json
python bench.py --workload cfg4 --cpu-steps 3 > gpurun_out/r03_bench_cfg4.json 2> gpurun_out/r03_bench_cfg4.err; cut -c1-200 gpurun_out/r03_bench_cfg4.json
python bench.py --workload cfg5 --steps 10 --warmup 3 > gpurun_out/r03_bench_cfg5.json 2> gpurun_out/r03_bench_cfg5.err; cut -c1-200 gpurun_out/r03_bench_cfg5.json
