"""Random frame graphs EXECUTED on the three-stream executor (tests/cpp/graph_cases.cpp --execute): every pass hashes its inputs
into its outputs (gr_debug_mix), four frames are enqueued back to back with no host synchronisation, and the swapchain images
must equal those of a serial run of the same graph (every pass on the graphics queue, nothing hoisted, nothing aliased).  A pass
that ran before a producer, a recycled allocation still in use, or a wrong copy of a hand-over ring changes the hash.
(Sensitivity, measured once: with the executor's cross-stream waits switched off -- GRANITE_UNSAFE_NO_CROSS_SYNC=1 -- 33 of the
40 graphs produce different frames.)"""
import json
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "granite_amd", "lib")


def test_pipelined_execution_of_random_graphs_equals_serial_execution(tmp_path):
    exe = str(tmp_path / "graph_cases")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "graph_cases.cpp"), "-o", exe, "-L" + LIB, "-lgranite_host",
                           "-lgranite_hip", "-Wl,-rpath," + LIB, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64"])
    out = subprocess.check_output([exe, "--execute", "40"], text=True, timeout=300)
    cases = list(map(json.loads, out.strip().splitlines()))
    assert len(cases) == 40
    seen = set()
    for c in cases:
        assert c["pipelined"] == c["serial"], c["case"]
        assert len(set(c["pipelined"])) == 4, c["case"]   # the frame number is part of every pass's salt
        seen.update(c["pipelined"])
    assert len(seen) == 160
