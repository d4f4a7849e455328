"""What Granite's tools/bench_aa.py does (it cannot travel to the GPU box): tools/aa-bench-headless once per AA method with two
input images, the --stat documents collected into one results file of bench_aa.py's shape (map_result_to_json).
usage: python tools/aa_bench_sweep.py [width height frames out.json]"""
import json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from granite_amd import png, synth

w, h, frames = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (3840, 2160, 100)
out = sys.argv[4] if len(sys.argv) > 4 else "aa_bench_results.json"
tmp = tempfile.mkdtemp()
images = []
for i in range(2):
    images.append(os.path.join(tmp, f"image{i}.png"))
    png.write_png(images[-1], synth.make_ldr_pattern(1920, 1080, synth.SEED + i))
methods = ["none", "fxaa", "smaaLow", "smaaMedium", "smaaHigh", "smaaUltra", "taaLow", "taaMedium", "taaHigh"]  # the live ones of bench_aa.py:158
runs = []
for method in methods:
    stat = os.path.join(tmp, "stat.json")
    subprocess.check_call([os.path.join(ROOT, "tools", "aa-bench-headless"), "--frames", str(frames), "--width", str(w), "--height", str(h),
                           "--input-images", *images, "--stat", stat, "--aa-method", method], stdout=subprocess.DEVNULL)
    parsed = json.load(open(stat))
    passes = {k: round(v["timePerAccumulationUs"], 1) for k, v in parsed.get("performance", {}).items()}
    runs.append({"method": method, "avg": parsed["averageFrameTimeUs"], "stdev": 0.0, "width": w, "height": h, "gpu": parsed["gpu"],
                 "version": parsed["driverVersion"], "passesUs": passes})
    print(method, f"{parsed['averageFrameTimeUs']:.1f} us/frame", passes)
json.dump({"runs": runs}, open(out, "w"), indent=1)
