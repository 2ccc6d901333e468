python -m pytest tests/test_gpu_kernels.py tests/test_gpu_seqgan.py -m gpu -q -x 2>&1 | tail -6
python bench.py --workload cfg5 --no-cpu-baseline --no-roofline --steps 10 --warmup 3 > gpurun_out/r3_bench_h_cfg5.json 2> gpurun_out/r3_bench_h_cfg5.err; grep "host enqueue" gpurun_out/r3_bench_h_cfg5.err
GIC_NO_FUSED_GUMBELMAX=1 python bench.py --workload cfg5 --no-cpu-baseline --no-roofline --steps 10 --warmup 3 > gpurun_out/r3_bench_h_cfg5_unfused.json 2> gpurun_out/r3_bench_h_cfg5_unfused.err; grep "host enqueue" gpurun_out/r3_bench_h_cfg5_unfused.err
