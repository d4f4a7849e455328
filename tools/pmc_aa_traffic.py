#!/usr/bin/env python3
"""gpurun_out/pmc_aa/summary.json (tools/pmc_aa.sh) -> aa_traffic.json: HBM-side bytes of the anti-aliasing kernels at 3840x2160 per launch
(2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction of MI355X_MICROARCH.md) against their algorithmic reads and writes, and the ratio.
bench.py quotes the file in the config-4 line (`roofline.aa_kernels`) while the kernels' sources still hash to what is recorded here.
usage: pmc_aa_traffic.py <summary.json> <out.json>"""
import hashlib, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCES = ["granite_amd/csrc/aa.hip", "granite_amd/csrc/aa_core.hpp", "granite_amd/csrc/aa_fast_kernels.hpp", "granite_amd/csrc/smaa_weights.hpp"]
PX = 3840 * 2160
# algorithmic bytes per pixel: (reads, writes) -- SURVEY 8d's per-pixel figures for the AA rows
KERNELS = {
    "fxaa": ("k_fxaa_fast", 4, 4, "RGBA8 in, RGBA8 out"),
    "smaa_edge_detection": ("k_smaa_edges_fast", 4, 2, "RGBA8 in, RG8 out"),
    "smaa_blend_weight": ("k_smaa_weights_bits", 2, 4, "RG8 edges in (+ 2 x 2 B of bit planes, + LUTs), RGBA8 out"),
    "smaa_neighbor_blend": ("k_smaa_blend_fast", 8, 4, "RGBA8 colour + RGBA8 weights in, RGBA8 out"),
    "taa_resolve_high": ("k_taa_fast<2, true>", 24, 16, "RGBA16F current + D32F + RG16F motion + RGBA16F history in, RGBA16F colour + history out"),
}
table = json.load(open(sys.argv[1]))
out = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over tools/aa_time.py 3840 2160 (tools/pmc_aa.sh); mean per dispatch",
       "correction": "FETCH_SIZE doubled (gfx950 rocprofv3 tallies 128-B read requests at 64 B: MI355X_MICROARCH.md, HBM section); WRITE_SIZE as reported",
       "sources_sha256": {p: hashlib.sha256(open(os.path.join(ROOT, p), "rb").read()).hexdigest() for p in SOURCES}, "kernels": {}}
for key, (prefix, rd, wr, what) in KERNELS.items():
    match = sorted(k for k in table if k.startswith(prefix))
    if not match or table[match[0]].get("FETCH_SIZE") is None or table[match[0]].get("WRITE_SIZE") is None:
        continue
    v = table[match[0]]
    fetch, write = 2.0 * v["FETCH_SIZE"] * 1024.0, v["WRITE_SIZE"] * 1024.0
    out["kernels"][key] = {"kernel": match[0], "what": what, "algorithmic_read_bytes": rd * PX, "algorithmic_write_bytes": wr * PX, "fetched_bytes": fetch,
                           "written_bytes": write, "fetch_over_algorithmic_reads": fetch / (rd * PX), "write_over_algorithmic_writes": write / (wr * PX),
                           "hbm_bytes_per_launch": fetch + write}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps({k: round(v["fetch_over_algorithmic_reads"], 2) for k, v in out["kernels"].items()}))
