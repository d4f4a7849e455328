#!/bin/bash
# Round 4, seventh GPU batch: binning with 64 lights per wave (lib) vs 32 (lib_oldbin): parity, kernel alone, frame.
O=gpurun_out/r04g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_lighting.py tests/test_gpu_app.py tests/test_gpu_fullsize.py -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -8 | cut -c1-300
cat > /tmp/binning_only.py <<'PY'
import os, sys
ROOT=os.environ.get('GRAFT_REPO_ROOT','/root/repo'); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests'))
from granite_amd import capi
from gpu_scene import Scene
gr=capi.Context(0); sc=Scene(3840,2160,4096); dev=sc.build_clusters_gpu(gr); prm=sc.cluster_params_struct()
for _ in range(5): gr.check(gr.lib.gr_cluster_binning(gr.handle,None,dev["transforms"].ptr,dev["setup"].ptr,dev["bitmask"].ptr,prm))
gr.sync(); gr.timing_enable(True); gr.timing_reset()
for _ in range(50): gr.check(gr.lib.gr_cluster_binning(gr.handle,None,dev["transforms"].ptr,dev["setup"].ptr,dev["bitmask"].ptr,prm))
gr.sync(); q=gr.timing_query(); print({k:round(1000*ms/c,1) for k,(c,ms) in q.items()})
PY
for lib in lib lib_oldbin lib lib_oldbin; do
  GRANITE_LIB_DIR=$lib timeout 120 python /tmp/binning_only.py 2>/dev/null | sed "s/^/binning alone $lib /"
  for i in 1 2; do GRANITE_LIB_DIR=$lib timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$lib.$i.json 2>/dev/null; python tools/bench_brief.py $O/bench_$lib.$i.json | sed "s/^/$lib /"; done
  GRANITE_LIB_DIR=$lib timeout 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --sustain-seconds 0 > $O/bench200_$lib.json 2>/dev/null; python tools/bench_brief.py $O/bench200_$lib.json | sed "s/^/$lib 200 steps /"
done
