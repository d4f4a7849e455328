g++ -O2 -std=c++17 -pthread -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tools/host_bench.cpp -o /tmp/host_bench -Lgranite_amd/lib -lgranite_host -lgranite_hip -Wl,-rpath,$PWD/granite_amd/lib -Wl,-rpath,/opt/rocm/lib -L/opt/rocm/lib -lamdhip64 && /tmp/host_bench
python bench.py --no-cpu-baseline > gpurun_out/bench_h.json; python tools/bench_brief.py gpurun_out/bench_h.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_h20.json; python tools/bench_brief.py gpurun_out/bench_h20.json
timeout 600 python -m pytest tests/test_gpu_app.py tests/test_gpu_aa.py -q -m gpu -x 2>&1 | grep -E "passed|failed|rror|assert" | head -5
