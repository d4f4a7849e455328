#!/usr/bin/env python3
"""Launch durations of one kernel over a run, from a GR_TIMING_DUMP file ("name start_us stop_us" per bracket).
usage: span_ramp.py spans.txt [kernel name = lighting]"""
import sys

name = sys.argv[2] if len(sys.argv) > 2 else "lighting"
rows = [l.split() for l in open(sys.argv[1]) if l.startswith(name + " ")]
spans = [(float(a), float(b)) for _, a, b in rows]
print(f"{name}: {len(spans)} bracketed launches")
prev_end = None
for i, (a, b) in enumerate(spans):
    gap = "" if prev_end is None else f"  {a - prev_end:9.1f} us after the previous bracketed launch ended"
    print(f"  #{i:3d}  start {a:10.1f} us  duration {b - a:7.1f} us{gap}")
    prev_end = b
