"""CPU: the C++ host layer built with -fsanitize=thread and -fsanitize=address (granite_amd/csrc/Makefile: SANITIZE=...) and
driven by a device-less program that runs the clusterer's threaded one-frame-ahead refresh for 60 frames (predictions that hold,
predictions that do not, a scene edited between frames): no sanitizer report, and every refresh packs what a synchronous one
packs.  The reference carries the same switches (CMakeLists.txt:44-46,104-117)."""
import json
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "granite_amd", "csrc")


@pytest.mark.parametrize("sanitizer", ["thread", "address"])
def test_threaded_light_refresh_under_sanitizer(tmp_path, sanitizer):
    out = str(tmp_path / "lib")
    os.makedirs(os.path.join(out, "obj"))
    # the kernels are not rebuilt: the instrumented host library links the stock libgranite_hip.so
    stock = os.path.join(ROOT, "granite_amd", "lib", "libgranite_hip.so")
    if not os.path.exists(stock):
        pytest.skip("libgranite_hip.so not built")
    shutil.copy(stock, out)
    subprocess.check_call(["make", "-s", "-j8", "-C", CSRC, f"OUT={out}", f"SANITIZE={sanitizer}", os.path.join(out, "libgranite_host.so")])
    exe = str(tmp_path / "clusterer_threads")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", f"-fsanitize={sanitizer}", "-fno-omit-frame-pointer", "-D__HIP_PLATFORM_AMD__",
                           "-I/opt/rocm/include", "-I" + CSRC, os.path.join(ROOT, "tests", "cpp", "clusterer_threads.cpp"), "-o", exe,
                           "-L" + out, "-lgranite_host", "-lgranite_hip", "-Wl,-rpath," + out, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib",
                           "-lamdhip64", "-pthread"])
    trace = str(tmp_path / "timeline.json")
    # GRANITE_TIMELINE_TRACE: the CPU timeline writer (host/timeline_trace.hpp) records from the same threads, under the same sanitizer
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66", ASAN_OPTIONS="detect_leaks=0 exitcode=66", GRANITE_TIMELINE_TRACE=trace)
    r = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stderr[-3000:], r.stdout[-500:])
    result = json.loads(r.stdout.strip().splitlines()[-1])
    assert result["mismatches"] == 0 and result["prefetch_hits"] >= 20, result
    assert "WARNING: ThreadSanitizer" not in r.stderr and "ERROR: AddressSanitizer" not in r.stderr
    check_timeline(trace, frames=60, prefetch_hits=result["prefetch_hits"])


def check_timeline(path, frames, prefetch_hits):
    """The trace is chrome://tracing JSON in the reference's event shape (util/timeline_trace_file.cpp:126-131: "B" / "E" pairs, tid and
    pid as strings, ts in microseconds): every refresh of the main thread, the helper threads' jobs under their own thread names."""
    events = json.load(open(path))
    spans = {}
    for tid in {e["tid"] for e in events}:
        stack, last = [], -1.0
        for e in (e for e in events if e["tid"] == tid and e["ph"] in "BE"):
            assert isinstance(e["pid"], str) and isinstance(e["ts"], float)
            assert e["ts"] >= last, (tid, e, last)  # time order per thread: a parent's "B" stands before its children's
            last = e["ts"]
            if e["ph"] == "B":
                stack.append(e)
            else:
                b = stack.pop()
                assert b["name"] == e["name"] and e["ts"] >= b["ts"] >= 0.0
                spans.setdefault((tid, e["name"]), []).append((b["ts"], e["ts"]))
        assert not stack, (tid, stack)
    names = {name for _, name in spans}
    assert {"clusterer-refresh", "light-sort-and-pack", "prefetch-next-frame-lights", "light-sort"} <= names, names
    # the program refreshes two clusterers per frame: the threaded one and a synchronous one it compares with
    assert len(spans[("main", "clusterer-refresh")]) == 2 * frames
    assert len(spans[("main", "light-sort-and-pack")]) == 2 * frames - prefetch_hits  # every refresh that did not adopt a prefetched result
    assert len(spans[("light-worker-0", "light-sort")]) >= prefetch_hits
    assert any(tid.startswith("light-worker-") and tid != "light-worker-0" for tid, _ in spans)
