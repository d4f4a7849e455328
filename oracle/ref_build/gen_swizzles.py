#!/usr/bin/env python3
"""ORACLE — TEST INFRASTRUCTURE ONLY.  Writes the swizzle members of the CPU GLSL vectors (every 2-, 3- and 4-letter selection
from xyzw / rgba) into swizzles_{2,3,4}.inc of the build scratch directory.  usage: gen_swizzles.py <output directory>"""
import itertools
import os
import sys

out = sys.argv[1]
os.makedirs(out, exist_ok=True)
for n in (2, 3, 4):
    lines = []
    for names in ("xyzw"[:n], "rgba"[:n]):
        for k in (2, 3, 4):
            for sel in itertools.product(range(n), repeat=k):
                name = "".join(names[i] for i in sel)
                lines.append("\t\tswz%d<T, %d, %s> %s;" % (k, n, ", ".join(map(str, sel)), name))
    open(os.path.join(out, "swizzles_%d.inc" % n), "w").write("\n".join(lines) + "\n")
