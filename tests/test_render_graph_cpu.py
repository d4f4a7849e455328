"""CPU: the restated RenderGraph bakes the reference's graphs to the expected order / sizes / aliasing without a GPU."""
import pytest

from granite_amd import app as gapp, capi


def graph_of(w, h, **kw):
    a = gapp.Application(w, h, device=-1, **kw)
    g = a.graph()
    a.close()
    return g


def sizes(g):
    return {r["name"]: (r["width"], r["height"]) for r in g["resources"]}


def test_compute_graph_order_and_level_sizes_4k():
    g = graph_of(3840, 2160)
    order = [p["name"] for p in g["passes"]]
    assert order == ["clustering-bindless", "gbuffer-main", "lighting-main", "bloom-compute", "tonemap"]
    s = sizes(g)
    # ceil(input * scale) relative to the full-res input (render_graph.cpp:3158-3170)
    assert s["threshold"] == (1920, 1080) and s["downsample-0"] == (960, 540) and s["downsample-1"] == (480, 270)
    assert s["downsample-2"] == (240, 135) and s["downsample-3"] == (120, 68)
    assert s["upsample-0"] == (960, 540) and s["tonemapped"] == (3840, 2160)
    hist = {r["name"] for r in g["resources"] if r["history"]}
    assert hist == {"downsample-3"}
    bufs = {r["name"]: r["buffer_size"] for r in g["resources"] if r["buffer_size"]}
    assert bufs["cluster-bitmask"] == 128 * 64 * 512 and bufs["cluster-range"] == 4096 * 8
    assert bufs["cluster-transforms"] == 852480 and bufs["average-luminance"] == 12
    # tonemapped has the swapchain's geometry + format => it is the swapchain image
    tm = next(r["phys"] for r in g["resources"] if r["name"] == "tonemapped")
    assert g["swapchain_phys"] == tm


def test_level_sizes_1080p_odd_levels():
    s = sizes(graph_of(1920, 1080))
    assert s["downsample-2"] == (120, 68) and s["downsample-3"] == (60, 34)


def test_graphics_variant_orders_threshold_before_luminance_update():
    g = graph_of(256, 256, lighting=False, compute_post=False)
    order = [p["name"] for p in g["passes"]]
    assert order[0] == "hdr-input-main" and order[-1] == "tonemap"
    # WAR: threshold reads last frame's exposure before adapt-luminance rewrites the aliased buffer
    assert order.index("bloom-threshold") < order.index("adapt-luminance")
    assert order.index("bloom-downsample-3") < order.index("adapt-luminance") < order.index("tonemap")
    phys = {}
    for p in g["passes"]:
        for r in p["writes"] + p["reads"]:
            phys[r["name"]] = r["phys"]
    assert phys["average-luminance"] == phys["average-luminance-updated"]


def test_rmw_declaration_aliases_emissive_and_hdr():
    g = graph_of(640, 360, rmw_emissive=True)
    phys = {w["name"]: w["phys"] for p in g["passes"] for w in p["writes"]}
    assert phys["HDR-main"] == phys["emissive-main"]
    g2 = graph_of(640, 360, rmw_emissive=False)
    phys2 = {w["name"]: w["phys"] for p in g2["passes"] for w in p["writes"]}
    assert phys2["HDR-main"] != phys2["emissive-main"]


def test_dry_application_refuses_to_render():
    a = gapp.Application(64, 64, device=-1)
    with pytest.raises(capi.GraniteHipError):
        a.render_frames(1)
    a.close()


def test_independent_compute_pass_is_hoisted_to_async_stream():
    """Executor policy: a COMPUTE pass that reads nothing produced in the graph (the cluster build) runs on the async
    stream; passes that consume graph resources stay on the generic stream."""
    g = graph_of(3840, 2160)
    streams = {p["name"]: p["stream"] for p in g["passes"]}
    assert streams["clustering-bindless"] == "async"
    assert streams["lighting-main"] == "generic" and streams["bloom-compute"] == "generic" and streams["tonemap"] == "generic"
    # what the hoisted pass writes alternates between two copies, so frame N+1's build never waits for frame N's lighting
    dbl = {r["name"] for r in g["resources"] if r["double_buffered"]}
    assert {"cluster-bitmask", "cluster-range", "cluster-transforms", "cluster-cull-setup", "cluster-transformed-spot"} <= dbl
    assert "average-luminance" not in dbl and "HDR-main" not in dbl
    # no lighting => no cluster pass => single stream
    g1 = graph_of(256, 256, lighting=False)
    assert {p["stream"] for p in g1["passes"]} == {"generic"}
