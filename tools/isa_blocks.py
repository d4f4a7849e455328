#!/usr/bin/env python3
"""Basic blocks of one kernel in a gfx950 assembly listing with their opcode-class histograms, largest first; loops (a block
that branches back to itself or to an earlier label) are marked.  Used for profiles/r03_lighting_isa_histogram.txt.
usage: isa_blocks.py file.s kernel-substring [min-instructions]"""
import collections, re, sys
sys.path.insert(0, __import__("os").path.dirname(__file__))
from isa_histogram import classify, kernels, CLASSES

text = open(sys.argv[1]).read()
want = sys.argv[2]
min_ins = int(sys.argv[3]) if len(sys.argv) > 3 else 12
for name, body in kernels(text):
    if want not in name:
        continue
    blocks, label, cur = [], "entry", []
    order = {}
    for line in body.split("\n"):
        m = re.match(r"^(\.LBB[\w]+):", line)
        if m:
            blocks.append((label, cur))
            label, cur = m.group(1), []
            continue
        if line.startswith("\t") and not line.strip().startswith((".", ";")):
            cur.append(line.strip())
    blocks.append((label, cur))
    for i, (l, _) in enumerate(blocks):
        order[l] = i
    print(f"{name}: {sum(len(b) for _, b in blocks)} instructions in {len(blocks)} blocks")
    for i, (l, ins) in enumerate(blocks):
        if len(ins) < min_ins:
            continue
        back = [x.split()[-1] for x in ins if x.startswith(("s_cbranch", "s_branch")) and order.get(x.split()[-1], 1 << 30) <= i]
        hist = collections.Counter(classify(x.split()[0]) for x in ins)
        valu = sum(v for k, v in hist.items() if k not in ("lds", "vmem", "smem", "branch", "waitcnt/nop", "salu", "other"))
        tag = f"  LOOP back to {back[0]}" if back else ""
        print(f"  block {l} (#{i}): {len(ins)} instructions, {valu} VALU{tag}")
        print("     " + ", ".join(f"{c} {hist[c]}" for c, _ in CLASSES if hist.get(c)))
