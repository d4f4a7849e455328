O=gpurun_out/r04q; mkdir -p $O
GRANITE_LIB_DIR=lib_bh8 timeout 600 python -m pytest tests/test_gpu_aa.py -q -m gpu -x 2>&1 | tail -2
for lib in lib lib_bh8; do AA_TIME_INPUT=card GRANITE_LIB_DIR=$lib timeout 200 python tools/aa_time.py 2>/dev/null | grep -i "taa\|Ultra\|fxaa\|x2160" | sed "s/^/$lib /" | cut -c1-110; done
for lib in lib lib_bh8 lib lib_bh8; do
  GRANITE_LIB_DIR=$lib timeout 300 python bench.py --workload config4_4k_smaa_taa --no-cpu-baseline > $O/bench_config4_$lib.json 2>/dev/null; python tools/bench_brief.py $O/bench_config4_$lib.json | sed "s/^/$lib config4 /"
done
