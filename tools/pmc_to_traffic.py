#!/usr/bin/env python3
"""gpurun_out/<tag>/summary.json (tools/pmc_passes.sh + tools/pmc_summary.py) -> profiles/pmc_traffic.json, the per-launch PMC
figures bench.py quotes in `roofline.traffic` / `roofline.valu_issue`.  HBM bytes = 2 * FETCH_SIZE + WRITE_SIZE (KB):
rocprofv3 on gfx950 tallies 128-byte read requests at 64 bytes (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is taken as
reported.  usage: pmc_to_traffic.py <summary.json> <out.json> <round>"""
import hashlib, json, os, sys

summary, out, rnd = sys.argv[1], sys.argv[2], int(sys.argv[3])
table = json.load(open(summary))
# kernel names as tools/pmc_summary.py shortens them; the first dispatched instantiation whose name starts with the prefix
want = {"lighting": "k_lighting<2, false", "tonemap": "k_tonemap<true, true, true>", "bloom_threshold": "k_bloom_threshold_2to1<true>"}
correction = ("FETCH_SIZE doubled (gfx950 rocprofv3 tallies 128-B read requests at 64 B: MI355X_MICROARCH.md, HBM section); "
              "WRITE_SIZE as reported")
classes = {"fma_f32": "SQ_INSTS_VALU_FMA_F32", "mul_f32": "SQ_INSTS_VALU_MUL_F32", "add_f32": "SQ_INSTS_VALU_ADD_F32",
           "transcendental_f32": "SQ_INSTS_VALU_TRANS_F32", "cvt": "SQ_INSTS_VALU_CVT", "int32": "SQ_INSTS_VALU_INT32"}
# the translation unit each kernel comes from: bench.py refuses these figures once the source has changed since they were taken
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sources = {"lighting": "granite_amd/csrc/lighting.hip", "tonemap": "granite_amd/csrc/post.hip", "bloom_threshold": "granite_amd/csrc/post.hip"}
def sha256(path):
    return hashlib.sha256(open(os.path.join(ROOT, path), "rb").read()).hexdigest()
kernels = {}
for key, name in want.items():
    matches = sorted(k for k in table if k.startswith(name))
    if not matches:
        continue
    name, vals = matches[0], table[matches[0]]
    entry = {"kernel": name, "source_file": sources[key], "source_sha256": sha256(sources[key]), "FETCH_SIZE_KB": vals.get("FETCH_SIZE"), "WRITE_SIZE_KB": vals.get("WRITE_SIZE"), "correction": correction}
    if vals.get("FETCH_SIZE") is not None and vals.get("WRITE_SIZE") is not None:
        entry["hbm_bytes_per_launch"] = (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0
    for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE"):
        entry[c] = vals.get(c)
    if vals.get("SQ_INSTS_VALU") and all(vals.get(c) is not None for c in classes.values()):
        hist = {k: vals[c] for k, c in classes.items()}
        hist["other (mov, cmp, min/max/med3, cndmask, dpp, readlane)"] = vals["SQ_INSTS_VALU"] - sum(hist.values())
        entry["valu_class_histogram"] = hist
    kernels[key] = entry
json.dump({"source": "rocprofv3 --pmc, separate passes (tools/pmc_passes.sh): bench.py --steps 3 --warmup 1, config3_4k_4096lights; mean per dispatch",
           "round": rnd, "kernels": kernels}, open(out, "w"), indent=1)
print(json.dumps({k: (v.get("hbm_bytes_per_launch"), v.get("SQ_INSTS_VALU")) for k, v in kernels.items()}))
