O=gpurun_out/r02b3; mkdir -p $O
( python tools/frame_parts.py full; python tools/frame_parts.py postonly; python tools/frame_parts.py hdr10
  for w in 3 5; do GR_LIGHTING_WGS_PER_CU=$w python tools/frame_parts.py full; done
  GRANITE_STREAM_PRIORITIES=hhh python tools/frame_parts.py full
  GRANITE_STREAM_PRIORITIES=lll python tools/frame_parts.py full ) 2>&1 | grep -v "^\[granite" | tee $O/frame_parts.txt
