# builds lighting.hip with each switch combination and times the kernel alone (4 and 5 workgroups per CU)
O=gpurun_out/r02b6; mkdir -p $O
run() { touch granite_amd/csrc/lighting.hip; make -s -j8 -C granite_amd/csrc EXTRA_lighting="$1" 2>&1 | grep -v warning | tail -2
  for w in 4 5; do echo "variant [$1] wgs $w $(GR_LIGHTING_WGS_PER_CU=$w timeout 120 python tools/lighting_only.py 2>&1 | tail -1)"; done; }
( run ""; run "-DLV_LOOP=0"; run "-DLV_NEAR_TEST=0"; run "-DLV_CONE_CULL=0"; run "" ) 2>&1 | tee $O/variants.txt
make -s -j8 -C granite_amd/csrc 2>&1 | tail -1
timeout 300 python -m pytest tests/test_gpu_lighting.py -q -m gpu -x 2>&1 | tail -2 | tee -a $O/variants.txt
