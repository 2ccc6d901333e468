python -m pytest tests/test_gpu_attention.py -m gpu -q 2>&1 | tail -3
bash tools/collect_profiles.sh r03 > gpurun_out/r03_collect.log 2>&1; tail -30 gpurun_out/r03_collect.log
python tools/step_timeline.py > gpurun_out/r03_step_timeline.txt 2>&1; tail -20 gpurun_out/r03_step_timeline.txt
