#!/usr/bin/env python3
"""Static opcode-class histogram of the kernels in a gfx950 assembly listing (hipcc -S --cuda-device-only).
usage: isa_histogram.py file.s [substring of a kernel name ...]"""
import collections, re, sys

CLASSES = [
    ("fp32 fma/mul/add", ("v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mac_f32", "v_mad_f32", "v_fma_mix", "v_mul_legacy")),
    ("packed fp32", ("v_pk_",)),
    ("transcendental", ("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")),
    ("div helpers", ("v_div_",)),
    ("floor/rndne/fract/trunc", ("v_floor", "v_rndne", "v_fract", "v_trunc", "v_ceil")),
    ("cvt", ("v_cvt",)),
    ("min/max/med3", ("v_min", "v_max", "v_med3")),
    ("cmp", ("v_cmp",)),
    ("cndmask", ("v_cndmask",)),
    ("mov", ("v_mov", "v_accvgpr")),
    ("cross-lane", ("v_readlane", "v_readfirstlane", "v_writelane", "v_permlane", "v_mbcnt", "ds_bpermute", "ds_permute", "ds_swizzle")),
    ("int32/bit", ("v_",)),
    ("lds", ("ds_",)),
    ("vmem", ("global_", "buffer_", "flat_", "scratch_")),
    ("smem", ("s_load", "s_buffer_load")),
    ("branch", ("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_call")),
    ("waitcnt/nop", ("s_waitcnt", "s_nop", "s_barrier", "s_sleep")),
    ("salu", ("s_",)),
]


def classify(op):
    for name, prefixes in CLASSES:
        if op.startswith(prefixes):
            return name
    return "other"


def kernels(text):
    for m in re.finditer(r"^([A-Za-z_][\w$.]*):\s*; @\1\n(.*?)^\s*\.end_amdhsa_kernel|^([A-Za-z_][\w$.]*):\s*; @\3\n(.*?)\n\.Lfunc_end", text, re.S | re.M):
        name, body = (m.group(1), m.group(2)) if m.group(1) else (m.group(3), m.group(4))
        yield name, body


def main():
    text = open(sys.argv[1]).read()
    want = sys.argv[2:]
    for name, body in kernels(text):
        if want and not any(w in name for w in want):
            continue
        ops = [l.split()[0] for l in body.split("\n") if l.startswith("\t") and not l.strip().startswith((".", ";"))]
        hist = collections.Counter(classify(o) for o in ops)
        valu = sum(v for k, v in hist.items() if k not in ("lds", "vmem", "smem", "branch", "waitcnt/nop", "salu", "other"))
        print(f"{name[:90]}: {len(ops)} instructions, {valu} VALU")
        for cname, _ in CLASSES:
            if hist.get(cname):
                print(f"    {cname:26s} {hist[cname]:6d}")


if __name__ == "__main__":
    main()
