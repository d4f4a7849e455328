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


def test_frame_front_is_pipelined_on_the_second_stream():
    """Executor policy: passes that depend on nothing carried over from the previous frame (cluster build, G-buffer,
    lighting) form the front of the frame and run on the second stream; passes with cross-frame feedback (the bloom
    pyramid's history + exposure) and everything after them stay on the first.  What crosses from front to back
    (HDR-main) is double-buffered, so frame N+1's front never waits for frame N's back."""
    g = graph_of(3840, 2160)
    streams = {p["name"]: p["stream"] for p in g["passes"]}
    assert streams == {"clustering-bindless": "async", "gbuffer-main": "async", "lighting-main": "front",
                       "bloom-compute": "generic", "tonemap": "generic"}
    dbl = {r["name"] for r in g["resources"] if r["double_buffered"]}
    # cluster build -> lighting and lighting -> bloom / tonemap cross streams
    assert {"HDR-main", "cluster-bitmask", "cluster-range", "cluster-transforms"} <= dbl
    assert "average-luminance" not in dbl and "downsample-3" not in dbl
    # TAA reads the depth and motion vectors the front produces: they cross to the back as well
    g2 = graph_of(1280, 720, pre_aa=gapp.POST_AA_TAA_HIGH)
    s2 = {p["name"]: p["stream"] for p in g2["passes"]}
    assert s2["lighting-main"] == "front" and s2["taa-resolve"] == "generic" and s2["tonemap"] == "generic"
    dbl2 = {r["name"] for r in g2["resources"] if r["double_buffered"]}
    assert "HDR-main" in dbl2 and "HDR-resolved" not in dbl2
    # no lighting: the frame starts at the HDR upload; nothing but the upload pass is independent of feedback
    g1 = graph_of(256, 256, lighting=False)
    assert {p["name"]: p["stream"] for p in g1["passes"]}["bloom-compute"] == "generic"


def test_graphics_bloom_chain_reuses_downsample_levels_for_the_upsample_chain():
    """The 10-pass graphics form of the HDR chain (hdr.cpp:402-561) writes colour attachments: once downsample-(k+1) exists,
    downsample-k is dead, so the upsample level of the same size moves into its allocation (build_aliases).  The compute
    form writes storage images, which are implicitly preserved and never alias (render_graph.cpp:1671-1673)."""
    g = graph_of(1280, 720, lighting=False, compute_post=False)
    res = {r["name"]: r for r in g["resources"]}
    for up, down in (("bloom-upsample-0", "bloom-downsample-2"), ("bloom-upsample-1", "bloom-downsample-1"),
                     ("bloom-upsample-2", "bloom-downsample-0")):
        assert res[up]["alias_of"] == res[down]["phys"], up
    assert res["bloom-downsample-3"]["alias_of"] == -1 and res["bloom-downsample-3"]["history"]
    assert res["tonemapped"]["alias_of"] == -1 and res["HDR-main"]["alias_of"] == -1
    g2 = graph_of(1280, 720)
    assert {r["alias_of"] for r in g2["resources"]} == {-1}
    g3 = graph_of(1280, 720, lighting=False, compute_post=False, alias_images=False)
    assert {r["alias_of"] for r in g3["resources"]} == {-1}


def test_gbuffer_and_lighting_form_one_physical_pass_like_the_reference():
    """build_physical_passes (render_graph.cpp:1221-1392): lighting consumes the G-buffer as input attachments of the same
    pixel, so the reference folds both into one VkRenderPass (two subpasses); everything that samples a texture another pass
    wrote, and every compute pass, starts a new physical pass."""
    g = graph_of(1280, 720, compute_post=False, post_aa=gapp.POST_AA_FXAA)
    pp = {p["name"]: p["physical_pass"] for p in g["passes"]}
    assert pp["gbuffer-main"] == pp["lighting-main"]
    others = [v for k, v in pp.items() if k not in ("gbuffer-main", "lighting-main")]
    assert len(set(others)) == len(others) and pp["gbuffer-main"] not in others
    assert [p["physical_pass"] for p in g["passes"]] == sorted(p["physical_pass"] for p in g["passes"])


def test_depth_hierarchy_pass_is_kept_alive_by_a_proxy_and_sized_in_whole_tiles():
    """setup_depth_hierarchy_pass (spd.cpp:196-232): chain = input rounded up to multiples of 64 (halved with
    output_downsample), plus a 4-byte counter; a proxy resource (render_graph.hpp:513-514) ties the otherwise unconsumed
    pass to the end of the frame.  It reads "depth-main", which the lighting pass publishes, and joins the frame front."""
    g = graph_of(1280, 720, depth_hierarchy=1)
    order = [p["name"] for p in g["passes"]]
    assert order == ["clustering-bindless", "gbuffer-main", "lighting-main", "depth-hiz", "bloom-compute", "tonemap"]
    res = {r["name"]: r for r in g["resources"]}
    assert (res["depth-hiz"]["width"], res["depth-hiz"]["height"], res["depth-hiz"]["format"]) == (1280, 768, 100)
    assert res["depth-hiz-counter"]["buffer_size"] == 4
    assert res["depth-hiz-ready"]["width"] == 0 and res["depth-hiz-ready"]["buffer_size"] == 0  # proxy: no memory
    assert {p["name"]: p["stream"] for p in g["passes"]}["depth-hiz"] == "front"
    g2 = graph_of(1280, 720, depth_hierarchy=2)
    res2 = {r["name"]: r for r in g2["resources"]}
    assert (res2["depth-hiz"]["width"], res2["depth-hiz"]["height"]) == (640, 384)
    assert "depth-hiz" not in {p["name"] for p in graph_of(1280, 720)["passes"]}
    with pytest.raises(capi.GraniteHipError):
        graph_of(256, 256, lighting=False, depth_hierarchy=1)


def test_resolution_scale_adds_the_upscale_passes_at_backbuffer_size():
    """scene_viewer_application.cpp:758-761,1263-1268 + aa.cpp:75-174: everything up to the AA output runs at
    ceil(size * scale); "<out>-scale" (EASU, R8G8B8A8_UNORM) and "<out>-sharpen" (RCAS, swapchain format) are swapchain
    sized, and the sharpened output is the swapchain image itself."""
    g = graph_of(1920, 1080, resolution_scale=2.0 / 3.0, post_aa=gapp.POST_AA_FXAA)
    order = [p["name"] for p in g["passes"]]
    assert order[-3:] == ["fxaa", "post-scale-output-scale", "post-scale-output-sharpen"]
    res = {r["name"]: r for r in g["resources"]}
    for name in ("HDR-main", "tonemapped", "post-aa-output", "albedo-main"):
        assert (res[name]["width"], res[name]["height"]) == (1280, 720), name
    assert (res["threshold"]["width"], res["threshold"]["height"]) == (640, 360)
    assert (res["post-scale-output-scale"]["width"], res["post-scale-output-scale"]["height"], res["post-scale-output-scale"]["format"]) == (1920, 1080, 37)
    assert g["swapchain_phys"] == res["post-scale-output"]["phys"]
    # without the sharpener the EASU output is UNORM: not the swapchain's format, so it is blitted
    g2 = graph_of(1920, 1080, resolution_scale=0.5, resolution_scale_sharpen=False)
    assert [p["name"] for p in g2["passes"]][-1] == "post-scale-output-scale" and g2["swapchain_phys"] == -1
    assert {r["name"]: (r["width"], r["height"]) for r in g2["resources"]}["HDR-main"] == (960, 540)
    with pytest.raises(capi.GraniteHipError):
        graph_of(1920, 1080, resolution_scale=1.5)


def test_ambient_occlusion_pass_sits_between_gbuffer_and_lighting():
    """scene_viewer_application.cpp:950-980: the SSAO pass (compute, R8_UNORM the size of the depth input) reads depth and
    normals; lighting takes its output as a texture.  A compute pass between them ends the G-buffer + lighting subpass merge."""
    g = graph_of(1280, 720, ambient_occlusion=True)
    order = [p["name"] for p in g["passes"]]
    assert order.index("gbuffer-main") < order.index("ssao-main") < order.index("lighting-main")
    res = {r["name"]: r for r in g["resources"]}
    assert (res["ssao-output-main"]["width"], res["ssao-output-main"]["height"], res["ssao-output-main"]["format"]) == (1280, 720, 9)
    lighting = next(p for p in g["passes"] if p["name"] == "lighting-main")
    assert "ssao-output-main" in {r["name"] for r in lighting["reads"]}
    pp = {p["name"]: p["physical_pass"] for p in g["passes"]}
    assert len({pp["gbuffer-main"], pp["ssao-main"], pp["lighting-main"]}) == 3


def test_hdr10_graph_ends_in_the_pq_encoder_on_a_10_bit_backbuffer():
    g = graph_of(1280, 720, hdr10=True, hdr_bloom=False)
    assert [p["name"] for p in g["passes"]][-2:] == ["lighting-main", "pq10"] and "ui" in {p["name"] for p in g["passes"]}
    res = {r["name"]: r for r in g["resources"]}
    assert res["ui-output"]["format"] == 64 and res["ui-temporary"]["format"] == 43
    assert g["swapchain_phys"] == res["ui-output"]["phys"]
    with pytest.raises(capi.GraniteHipError):
        graph_of(1280, 720, hdr10=True)  # bloom + tonemap and the PQ encoder are alternatives
