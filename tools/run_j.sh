python -m pytest tests/test_gpu_encoder.py -m gpu -q -x -k "block_output or trunk_forward_bf16_at_bench" 2>&1 | tail -4
for cfg in "A" "B GIC_RES_IN_MIN_ROWS=0 GIC_RES_IN_MAX_COUT=512 GIC_TILE8_RES_NS_SMALL=1" "C GIC_RES_IN_MIN_ROWS=0 GIC_RES_IN_MAX_COUT=512" "D GIC_RES_IN_MIN_ROWS=0 GIC_RES_IN_MAX_COUT=256" "E GIC_RES_IN_MIN_ROWS=0 GIC_RES_IN_MAX_COUT=512 GIC_TILE8_RES_NS_SMALL=2"; do
  set -- $cfg; name=$1; shift
  env "$@" python bench.py --no-cpu-baseline --no-roofline --steps 40 > gpurun_out/r3_bench_j_$name.json 2> gpurun_out/r3_bench_j_$name.err
  echo "$name $* : $(python -c "import json;print(json.load(open('gpurun_out/r3_bench_j_$name.json'))['ms_per_step'])")"
done
