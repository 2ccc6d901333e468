python -m pytest tests -m gpu -q -x > gpurun_out/r3_tests_c.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_tests_c.log; tail -8 gpurun_out/r3_tests_c.log
for v in graph eager; do
  if [ $v = eager ]; then export GIC_NO_STEP_GRAPH=1; fi
  python bench.py --no-cpu-baseline --no-roofline > gpurun_out/r3_bench_c_$v.json 2> gpurun_out/r3_bench_c_$v.err; tail -2 gpurun_out/r3_bench_c_$v.err | head -1; cut -c1-120 gpurun_out/r3_bench_c_$v.json
done
GIC_LIB_VARIANT=nt python bench.py --no-cpu-baseline --no-roofline > gpurun_out/r3_bench_c_nt.json 2> gpurun_out/r3_bench_c_nt.err; tail -2 gpurun_out/r3_bench_c_nt.err | head -1; cut -c1-120 gpurun_out/r3_bench_c_nt.json
unset GIC_NO_STEP_GRAPH
GIC_LIB_VARIANT=stamps python tools/conv_stamps.py 64,14,1024,256,1,1,0 64,7,2048,512,1,1,0 64,14,1024,512,1,1,0 64,7,512,2048,1,1,0,1 64,14,256,256,3,1,1,1 64,7,512,512,3,1,1,1 64,28,128,128,3,1,1,1 64,28,256,256,3,2,1 64,14,512,512,3,2,1 > gpurun_out/r3_stamps_c.txt 2>&1; grep "per launch" gpurun_out/r3_stamps_c.txt
