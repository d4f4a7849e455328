for i in 1 2; do
 for v in "" "GR_LIGHTING_WGS_PER_CU=5" "GR_LIGHTING_PX=1 GR_LIGHTING_WGS_PER_CU=7" "GR_LIGHTING_PX=1 GR_LIGHTING_WGS_PER_CU=8"; do env $v timeout 120 python tools/lighting_only.py | sed "s/^/[$v] /"; done
done
bash tools/frame_ab.sh r05_px "config3_4k_4096lights" wg5:GR_LIGHTING_WGS_PER_CU=5 px1w8:GR_LIGHTING_PX=1,GR_LIGHTING_WGS_PER_CU=8 px1w7:GR_LIGHTING_PX=1,GR_LIGHTING_WGS_PER_CU=7
bash tools/frame_ab.sh r05_c4 "config4_4k_smaa_taa config2_1080p_256lights" base wg5:GR_LIGHTING_WGS_PER_CU=5
