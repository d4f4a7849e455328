"""CPU: the restated RenderGraph declaration / bake API driven from C++ exactly as a Granite application would drive it
(tests/cpp/graph_cases.cpp, no device): the topology of the reference's tests/render_graph_sandbox.cpp, dead-pass culling,
read-modify-write aliasing with InputRelative sizes, and the std::logic_error contract of bake()/validation
(render_graph.cpp:562-622,2842-2846,3001-3012)."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "granite_amd", "lib")


@pytest.fixture(scope="module")
def cases(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("graph_cases") / "graph_cases")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           os.path.join(ROOT, "tests", "cpp", "graph_cases.cpp"), "-o", exe, "-L" + LIB, "-lgranite_host",
                           "-lgranite_hip", "-Wl,-rpath," + LIB, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64"])
    out = subprocess.check_output([exe], text=True)
    return {c["case"]: c for c in map(json.loads, out.strip().splitlines())}


def test_sandbox_topology_bakes_in_declaration_order(cases):
    g = cases["sandbox"]["graph"]
    assert [p["name"] for p in g["passes"]] == ["depth", "first", "compute", "final"]
    res = {r["name"]: r for r in g["resources"]}
    # SwapchainRelative targets take the backbuffer size; the Absolute storage image keeps 1280x720
    assert (res["depth"]["width"], res["depth"]["height"]) == (1920, 1080)
    assert (res["image"]["width"], res["image"]["height"], res["image"]["format"]) == (1280, 720, 37)
    # "back" has the swapchain's geometry and format: it IS the swapchain image
    assert g["swapchain_phys"] == res["back"]["phys"]
    queues = {p["name"]: p["queue"] for p in g["passes"]}
    assert queues["compute"] == 4  # RENDER_GRAPH_QUEUE_ASYNC_COMPUTE_BIT
    streams = {p["name"]: p["stream"] for p in g["passes"]}
    assert streams["compute"] == "async" and streams["final"] == "generic"


def test_unreferenced_pass_is_culled(cases):
    g = cases["culling"]["graph"]
    assert [p["name"] for p in g["passes"]] == ["a", "b"]
    assert "dead-out" not in {r["name"] for r in g["resources"]}


def test_rmw_aliases_and_input_relative_sizes(cases):
    g = cases["rmw"]["graph"]
    phys = {}
    for p in g["passes"]:
        for r in p["writes"] + p["reads"]:
            phys[r["name"]] = r["phys"]
    assert phys["base"] == phys["sum"] and phys["small"] != phys["sum"]
    res = {r["phys"]: r for r in g["resources"]}
    small = res[phys["small"]]
    assert (small["width"], small["height"]) == (251, 84)  # ceil(1001 * 0.25), ceil(333 * 0.25)
    # two writers share physical image 0: it is not a hand-over resource, ordering is by events only
    assert res[phys["sum"]]["double_buffered"] is False


def test_disjoint_lifetime_attachments_share_an_allocation(cases):
    """build_aliases (render_graph.cpp:1548-1746): same geometry + disjoint lifetimes => one allocation; storage images,
    history images and the swapchain never alias; nothing joins an allocation it overlaps any member of."""
    g = cases["aliasing"]["graph"]
    res = {r["name"]: r for r in g["resources"]}
    assert res["a"]["alias_of"] == -1 and res["b"]["alias_of"] == -1
    assert res["c"]["alias_of"] == res["a"]["phys"]
    assert res["d"]["alias_of"] == res["b"]["phys"]      # not a's: c lives there and p3 reads c while writing d
    assert res["s"]["alias_of"] == -1 and res["screen"]["alias_of"] == -1
    # With the frame front pipelined on other streams (p0 on the async stream, p1..p4 on the front stream), "a" crosses
    # streams as a hand-over ring and stays out of it; b/d live on the front stream alone and still share.
    gp = cases["aliasing-pipelined"]["graph"]
    rp = {r["name"]: r for r in gp["resources"]}
    assert {p["name"]: p["stream"] for p in gp["passes"]}["p0"] == "async" and rp["a"]["double_buffered"]
    assert rp["c"]["alias_of"] == -1 and rp["d"]["alias_of"] == rp["b"]["phys"]
    off = {r["name"]: r["alias_of"] for r in cases["aliasing-off"]["graph"]["resources"]}
    assert set(off.values()) == {-1}


def test_validation_errors_are_logic_errors_with_the_reference_messages(cases):
    assert cases["no-writer"]["error"] == "No pass exists which writes to resource."
    assert cases["missing-backbuffer"]["error"] == "Backbuffer source does not exist."
    assert cases["cycle"]["error"] == "Cycle detected."
    assert cases["texture-as-buffer"]["error"] == "Resource is not a buffer: x"
    # a colour input of different size is not an error: it becomes a scaled input (render_graph.cpp:575-583)
    assert cases["rmw-size-mismatch"]["error"] is None
