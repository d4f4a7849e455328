#!/bin/bash
# Round 2, first GPU batch: parity at full size, issue-rate microbenchmark, baseline bench + rocprof stats, quick A/Bs.
O=gpurun_out/r02b1; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -15 ) > $O/pytest.txt; cat $O/pytest.txt
hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_bench tools/valu_bench.hip && timeout 120 /tmp/valu_bench > $O/valu_bench.txt 2>&1; cat $O/valu_bench.txt
timeout 300 python tools/ulp_hist.py > $O/ulp_hist.json 2>$O/ulp_hist.err; cat $O/ulp_hist.json
timeout 300 python bench.py --steps 200 --warmup 20 > $O/bench_base.json 2>$O/bench_base.err; tail -c 1500 $O/bench_base.json
timeout 120 python tools/lighting_only.py > $O/light_base.txt 2>&1; tail -1 $O/light_base.txt
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o base --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 20 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_prof.json 2>$GRAFT_REPO_ROOT/$O/bench_prof.err )
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_base.csv
# A/B 1: post kernels at wave priority 3
touch granite_amd/csrc/post.hip; make -s -j8 -C granite_amd/csrc EXTRA_post=-DGR_POST_WAVE_PRIORITY=3 2>&1 | tail -3
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_prio3.json 2>$O/bench_prio3.err; python tools/bench_brief.py $O/bench_prio3.json
touch granite_amd/csrc/post.hip; make -s -j8 -C granite_amd/csrc 2>&1 | tail -3
# A/B 2: lighting without the SLP vectoriser
touch granite_amd/csrc/lighting.hip; make -s -j8 -C granite_amd/csrc EXTRA_lighting=-fno-slp-vectorize 2>&1 | tail -3
timeout 120 python tools/lighting_only.py > $O/light_noslp.txt 2>&1; tail -1 $O/light_noslp.txt
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_noslp.json 2>$O/bench_noslp.err; python tools/bench_brief.py $O/bench_noslp.json
touch granite_amd/csrc/lighting.hip; make -s -j8 -C granite_amd/csrc 2>&1 | tail -3
python tools/bench_brief.py $O/bench_base.json
