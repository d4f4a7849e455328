#!/usr/bin/env python3
"""gpurun_out/<tag>/summary.txt (tools/pmc_passes.sh + tools/pmc_summary.py) -> profiles/pmc_traffic.json, the per-launch PMC
figures bench.py quotes in `roofline.traffic`.  HBM bytes = 2 * FETCH_SIZE + WRITE_SIZE (KB): rocprofv3 on gfx950 tallies
128-byte read requests at 64 bytes (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is taken as reported."""
import json, sys

summary, out = sys.argv[1], sys.argv[2]
rows = [l.rstrip("\n") for l in open(summary) if l.strip()]
header = rows[0]
# the summary prints the last 18 characters of each counter name in 20-character right-aligned columns after a 42-character kernel column
full = ["FETCH_SIZE", "GRBM_GUI_ACTIVE", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM", "SQ_BUSY_CYCLES",
        "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_VALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INST_CYCLES_SALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
        "SQ_WAVES", "SQ_WAVE_CYCLES", "WRITE_SIZE"]
cols = header[42:].split()
assert len(cols) == len(full) and all(f.endswith(c) for f, c in zip(full, cols)), cols
want = {"lighting": "k_lighting<2, false>", "tonemap": "k_tonemap<true, true, true>", "bloom_threshold": "k_bloom_threshold<true>"}
correction = ("FETCH_SIZE doubled (gfx950 rocprofv3 tallies 128-B read requests at 64 B: MI355X_MICROARCH.md, HBM section); "
              "WRITE_SIZE as reported")
kernels = {}
for key, name in want.items():
    line = next(r for r in rows[1:] if r[:42].strip() == name)
    vals = dict(zip(full, (float(v) if v != "-" else None for v in line[42:].split())))
    kernels[key] = {"kernel": name, "FETCH_SIZE_KB": vals["FETCH_SIZE"], "WRITE_SIZE_KB": vals["WRITE_SIZE"],
                    "hbm_bytes_per_launch": (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0, "correction": correction,
                    **{c: vals[c] for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "GRBM_GUI_ACTIVE")}}
json.dump({"source": "rocprofv3 --pmc, separate passes (tools/pmc_passes.sh): bench.py --steps 3 --warmup 1, config3_4k_4096lights; mean per dispatch",
           "round": 1, "kernels": kernels}, open(out, "w"), indent=1)
print(json.dumps({k: (v["hbm_bytes_per_launch"], v["SQ_INSTS_VALU"]) for k, v in kernels.items()}))
