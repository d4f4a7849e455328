#!/usr/bin/env python3
"""ORACLE — TEST INFRASTRUCTURE ONLY.

Re-spells a GLSL shader of the reference so that it compiles as C++ against glsl_cpu.hpp: the shader's statements,
expressions, constants and their order are left exactly as written; only what has no meaning outside a GPU pipeline is
touched (version / precision lines, layout() qualifiers, interface-block syntax, parameter qualifiers) and floating literals
get an `f` suffix, because a GLSL literal is fp32 while a C++ literal is a double.  #include directives are expanded from the
reference tree.  The output goes to the build scratch directory (removed after the compile): reference text never enters this repository.

usage: glsl2cpp.py <shader path under the reference> <output .inc>"""
import os
import re
import sys

FLOAT_LITERAL = re.compile(r"(?<![\w.])((?:\d+\.\d*|\.\d+)(?:[eE][-+]?\d+)?|\d+[eE][-+]?\d+)(?![\w.]|f\b)")
BLOCK = re.compile(r"\b(?:uniform|buffer)\s+(\w+)\s*\{([^{}]*)\}\s*(\w*)\s*;", re.S)
PARAM_OUT = re.compile(r"\b(?:inout|out)\s+((?:(?:mediump|highp|lowp)\s+)?)(\w+)\s+(\w+)(?=\s*([,)\[]))")
PARAM_IN = re.compile(r"(?<=[(,])\s*in\s+(?=(?:(?:mediump|highp|lowp)\s+)?\w+\s+\w+\s*[,)])")


def expand_includes(path: str, stack=()) -> str:
    """Textual #include expansion.  Every inclusion is expanded -- a header may first appear inside a conditional block that
    ends up disabled -- and the headers' own #ifndef guards do the de-duplication when the C++ preprocessor runs."""
    real = os.path.realpath(path)
    if real in stack:
        return ""
    out = []
    for line in open(path, encoding="utf-8", errors="replace").read().splitlines():
        m = re.match(r'\s*#\s*include\s+"([^"]+)"', line)
        if m:
            out.append(expand_includes(os.path.join(os.path.dirname(path), m.group(1)), stack + (real,)))
        else:
            out.append(line)
    return "\n".join(out)


def respell(text: str) -> str:
    lines = []
    for line in text.splitlines():
        s = line.strip()
        if s.startswith("#version") or s.startswith("#extension") or re.match(r"precision\s+\w+\s+\w+\s*;", s):
            continue
        lines.append(line)
    text = "\n".join(lines)
    text = re.sub(r"layout\s*\([^()]*\)", "", text)
    text = re.sub(r"^\s*in\s*;\s*$", "", text, flags=re.M)  # what is left of layout(local_size...) in;

    def block(m):
        name, body, instance = m.group(1), m.group(2), m.group(3)
        body = re.sub(r"(\w+)\s+(\w+)\s*\[\s*\]\s*;", r"\1 *\2;", body)  # run-time sized array: a pointer
        if instance:
            return "struct %s_block {%s} %s;" % (name, body, instance)
        return body  # no instance name: the members are globals

    text = BLOCK.sub(block, text)
    # stage inputs / outputs become per-invocation globals
    text = re.sub(r"^[ \t]*(?:in|out)[ \t]+((?:(?:mediump|highp|lowp|flat)[ \t]+)*\w+[ \t]+\w+[ \t]*;)", r"thread_local \1", text, flags=re.M)
    text = text.replace("flat ", "")
    # out / inout parameters are references; arrays already are
    text = PARAM_OUT.sub(lambda m: "%s%s %s%s" % (m.group(1), m.group(2), "" if m.group(4) == "[" else "&", m.group(3)), text)
    text = re.sub(r"\bvec4\s*\[\s*\]\s*\(", "glsl::array_of_vec4(", text)  # array constructor
    text = PARAM_IN.sub(" ", text)
    # qualifier macros (ffx_a.h: "#define outAF2 out AF2"): the same meaning, spelled for C++
    text = re.sub(r"^([ \t]*#[ \t]*define[ \t]+\w+)[ \t]+(?:out|inout)[ \t]+(\w+)[ \t]*$", r"\1 \2 &", text, flags=re.M)
    text = re.sub(r"^([ \t]*#[ \t]*define[ \t]+\w+)[ \t]+in[ \t]+(\w+)[ \t]*$", r"\1 \2", text, flags=re.M)
    text = FLOAT_LITERAL.sub(lambda m: m.group(1) + "f", text)
    return text


def main():
    src, dst = sys.argv[1], sys.argv[2]
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    with open(dst, "w") as f:
        f.write("// GENERATED from %s by oracle/ref_build/glsl2cpp.py -- reference text, do not commit.\n" % src)
        text = respell(expand_includes(src))
        f.write(text)
        f.write("\n")
        # Include guards are macros: forget them, so that another variant of the shader can be compiled in another namespace.
        for guard in sorted(set(re.findall(r"^\s*#\s*ifndef\s+(\w+)\s*\n\s*#\s*define\s+\1\b", text, flags=re.M))):
            f.write("#undef %s\n" % guard)


if __name__ == "__main__":
    main()
