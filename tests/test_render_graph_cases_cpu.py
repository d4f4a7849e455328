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


# ---- random graphs: invariants of bake() ------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def random_graphs(cases, tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("graph_cases_random") / "graph_cases")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           os.path.join(ROOT, "tests", "cpp", "graph_cases.cpp"), "-o", exe, "-L" + LIB, "-lgranite_host",
                           "-lgranite_hip", "-Wl,-rpath," + LIB, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64"])
    out = subprocess.check_output([exe, "--random", "60"], text=True)
    return list(map(json.loads, out.strip().splitlines()))


def _live_passes(declared):
    """Passes that contribute to "back", by walking the declared edges backwards (what bake() must keep, render_graph.cpp
    traverse_dependencies)."""
    by_name = {p["name"]: p for p in declared}
    writer = {w: p["name"] for p in declared for w in p["writes"]}
    live, stack = set(), ["final"]
    while stack:
        n = stack.pop()
        if n in live:
            continue
        live.add(n)
        stack.extend(writer[r] for r in by_name[n]["reads"])
    return live


def test_random_graphs_keep_exactly_the_live_passes_in_dependency_order(random_graphs):
    for case in random_graphs:
        declared, g = case["declared"], case["graph"]
        order = [p["name"] for p in g["passes"]]
        assert set(order) == _live_passes(declared), case["case"]
        assert len(order) == len(set(order)) and order[-1] == "final"
        position = {n: i for i, n in enumerate(order)}
        writer = {w: p["name"] for p in declared for w in p["writes"]}
        for p in declared:
            if p["name"] in position:
                for r in p["reads"]:
                    assert position[writer[r]] < position[p["name"]], (case["case"], r)
        # what a kept pass declares is what the baked pass reads and writes
        baked = {p["name"]: p for p in g["passes"]}
        for p in declared:
            if p["name"] in baked:
                assert sorted(r["name"] for r in baked[p["name"]]["reads"]) == sorted(p["reads"])
                assert sorted(w["name"] for w in baked[p["name"]]["writes"]) == sorted(p["writes"])
                assert baked[p["name"]]["queue"] == p["queue"]


def test_random_graphs_alias_only_disjoint_single_stream_images_of_equal_geometry(random_graphs):
    shared_any = 0
    for case in random_graphs:
        g = case["graph"]
        res = {r["phys"]: r for r in g["resources"]}
        use = {}      # phys -> [first position, last position, set of streams]
        for i, p in enumerate(g["passes"]):
            for r in p["reads"] + p["writes"]:
                u = use.setdefault(r["phys"], [i, i, set()])
                u[1] = i
                u[2].add(p["stream"])
        groups = {}
        for phys, r in res.items():
            root = phys
            while res[root]["alias_of"] >= 0:
                root = res[root]["alias_of"]
            groups.setdefault(root, []).append(phys)
        if case["case"].endswith("-0"):
            assert all(len(v) == 1 for v in groups.values()), case["case"]   # set_alias_disjoint_images(false)
            continue
        for root, members in groups.items():
            if len(members) < 2:
                continue
            shared_any += 1
            members.sort(key=lambda m: use[m][0])
            for a, b in zip(members, members[1:]):
                assert use[a][1] < use[b][0], (case["case"], res[a]["name"], res[b]["name"])   # lifetimes do not touch
            first = res[members[0]]
            for m in members:
                assert (res[m]["width"], res[m]["height"], res[m]["format"]) == (first["width"], first["height"], first["format"])
                assert len(use[m][2]) == 1 and not res[m]["double_buffered"] and not res[m]["history"]
            assert len({next(iter(use[m][2])) for m in members}) == 1      # one in-order stream orders the reuse
        assert g["swapchain_phys"] == next(r["phys"] for r in g["resources"] if r["name"] == "back")
    assert shared_any >= 5   # the generator does produce aliasing opportunities (most images cross streams and become rings)


def test_random_graphs_physical_passes_are_runs_of_graphics_passes(random_graphs):
    for case in random_graphs:
        g = case["graph"]
        groups = {}
        for i, p in enumerate(g["passes"]):
            groups.setdefault(p["physical_pass"], []).append((i, p))
        previous_end = -1
        for index in sorted(groups):
            members = groups[index]
            positions = [i for i, _ in members]
            assert positions == list(range(positions[0], positions[0] + len(positions))) and positions[0] == previous_end + 1
            previous_end = positions[-1]
            if len(members) > 1:
                assert all(p["queue"] == 1 for _, p in members), case["case"]    # only graphics passes merge
                sizes = set()
                res = {r["phys"]: r for r in g["resources"]}
                for _, p in members:
                    for w in p["writes"]:
                        sizes.add((res[w["phys"]]["width"], res[w["phys"]]["height"]))
                assert len(sizes) == 1, case["case"]                             # subpasses share the render area
