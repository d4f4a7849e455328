bash tools/frame_ab.sh r05_up "config3_4k_4096lights config4_4k_smaa_taa config5_8k" fused nofuse:GR_NO_UP_FUSION=1 -- --steps 100 --warmup 10 --sustain-seconds 1
for l in lib; do GRANITE_LIB_DIR=$l timeout 120 python tools/post_only.py 2>/dev/null | sed "s/^/fused /"; GR_NO_UP_FUSION=1 timeout 120 python tools/post_only.py 2>/dev/null | sed "s/^/nofuse /"; done
