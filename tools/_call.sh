timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r05_pytest_gpu4.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error|Error" gpurun_out/r05_pytest_gpu4.log | tail -5
for i in 1 2; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --sustain-seconds 1 > gpurun_out/r05_line.$i.json 2>/dev/null; python tools/bench_brief.py gpurun_out/r05_line.$i.json; done
bash tools/frame_ab.sh r05_c4c "config4_4k_smaa_taa config3_4k_4096lights" wg4:GR_LIGHTING_WGS_PER_CU=4 wg5 -- --steps 100 --warmup 10 --sustain-seconds 1
