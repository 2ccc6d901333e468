python -m pytest tests/test_gpu_attention.py tests/test_gpu_step.py -m gpu -q 2>&1 | tail -5
python bench.py --workload cfg4 --no-cpu-baseline --no-roofline > gpurun_out/r3_bench_m_cfg4.json 2> gpurun_out/r3_bench_m_cfg4.err; grep "host enqueue" gpurun_out/r3_bench_m_cfg4.err
GIC_NO_STEP_GRAPH=1 python bench.py --workload cfg4 --no-cpu-baseline --no-roofline > gpurun_out/r3_bench_m_cfg4e.json 2> gpurun_out/r3_bench_m_cfg4e.err; grep "host enqueue" gpurun_out/r3_bench_m_cfg4e.err
