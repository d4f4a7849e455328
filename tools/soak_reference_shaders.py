"""Soak: the oracle against the executed reference shaders (oracle/_ref/libref_shaders.so) over many seeds and sizes -- lighting, post
chain, AA, FSR.  Prints the number of mismatching values per pass (all zero expected).  Needs `make -C oracle/ref_build`."""
import ctypes as C, numpy as np, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc
from granite_amd import synth
ref = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libref_shaders.so"))
t0 = time.time(); total = 0; bad = 0
for seed in range(1, 25):
    w, h = 128 + 7 * seed, 72 + 5 * seed
    cam = synth.Camera(w, h, fovy_deg=40 + seed, eye=(0.3 * seed - 3, 2.0 + 0.05 * seed, 8.0 - 0.1 * seed))
    gbuf = synth.make_gbuffer(cam, seed)
    rp = cam.render_params()
    try:
        descs = synth.make_lights(cam, 100 + 37 * seed, seed=seed)
    except TypeError:
        descs = synth.make_lights(cam, 100 + 37 * seed)
    n, lights, model, tmask, _ = orc.pack_lights(descs, rp[99:102])
    prm = orc.cluster_params(rp, *synth.CLUSTER_RESOLUTION, n)
    cb = orc.cluster_build(rp, prm, lights, model, tmask, n, synth.CLUSTER_RESOLUTION[2])
    args = (gbuf, rp, prm, lights, tmask, cb["bitmask"], cb["range"], synth.DIRECTIONAL_COLOR, synth.DIRECTIONAL_DIRECTION)
    a = orc.lighting(*args); b = orc.lighting(*args, entry=ref.ref_lighting)
    total += a.size; bad += int((a != b).sum())
print("lighting soak: mismatching halves", bad, "of", total, f"{time.time()-t0:.0f}s")
def p(a): return None if a is None else a.ctypes.data
from granite_amd.data import load_smaa_luts
P = C.c_void_p
def p(a): return None if a is None else a.ctypes.data
ref.ref_bloom_threshold.argtypes = [P, C.c_int, C.c_int, P, C.c_int, C.c_int, P]
ref.ref_bloom_downsample.argtypes = [P, C.c_int, C.c_int, P, C.c_int, C.c_int, P, C.c_float]
ref.ref_bloom_upsample.argtypes = [P, C.c_int, C.c_int, P, C.c_int, C.c_int]
ref.ref_luminance.argtypes = [P, C.c_int, C.c_int, P, C.c_float, C.c_float, C.c_float]
ref.ref_tonemap.argtypes = [P, C.c_int, C.c_int, P, C.c_int, C.c_int, P, C.c_float, P]
ref.ref_fxaa.argtypes = [P, C.c_int, C.c_int, P, C.c_int]
ref.ref_taa_resolve.argtypes = [P, P, P, P, C.c_int, C.c_int, P, C.c_int, P, P]
ref.ref_fsr_easu.argtypes = [P, C.c_int, C.c_int, P, C.c_int, C.c_int]
ref.ref_fsr_easu_fp16.argtypes = [P, C.c_int, C.c_int, P, C.c_int, C.c_int]
ref.ref_fsr_rcas.argtypes = [P, C.c_int, C.c_int, P, C.c_float, C.c_int]
area, search = load_smaa_luts()
bad = {}
def chk(name, a, b):
    bad[name] = bad.get(name, 0) + int((np.asarray(a) != np.asarray(b)).sum())
t0 = time.time()
for seed in range(1, 13):
    w, h = 96 + 11 * seed, 64 + 7 * seed
    hdr = synth.make_hdr(w, h, seed)
    lum = np.array([0.1 * seed - 0.5, 2.0 ** (0.1 * seed - 0.5), 2.0 ** -(0.1 * seed - 0.5)], np.float32)
    sz = [orc.level_size(w, h, s) for s in (0.5, 0.25, 0.125, 0.0625, 0.03125)]
    t = orc.bloom_threshold(hdr, *sz[0], lum3=lum); g = np.zeros_like(t)
    ref.ref_bloom_threshold(p(hdr), w, h, p(g), sz[0][0], sz[0][1], p(lum)); chk("threshold", g, t)
    lv = [t]
    for i in range(1, 5):
        hist = orc.bloom_downsample(lv[-1], *sz[i]) if i == 4 else None
        d = orc.bloom_downsample(lv[-1], *sz[i], history=hist, lerp=0.07 if hist is not None else 0.0); g = np.zeros_like(d)
        ref.ref_bloom_downsample(p(lv[-1]), lv[-1].shape[1], lv[-1].shape[0], p(g), sz[i][0], sz[i][1], p(hist), 0.07); chk("down", g, d); lv.append(d)
    want = orc.luminance(lv[-1], lum.copy(), 0.0069); got = lum.copy()
    ref.ref_luminance(p(lv[-1]), lv[-1].shape[1], lv[-1].shape[0], p(got), 0.0069, -3.0, 2.0); chk("lum", got.view(np.uint32), want.view(np.uint32))
    up = lv[-1]
    for i in (3, 2, 1):
        u = orc.bloom_upsample(up, *sz[i]); g = np.zeros_like(u)
        ref.ref_bloom_upsample(p(up), up.shape[1], up.shape[0], p(g), sz[i][0], sz[i][1]); chk("up", g, u); up = u
    tm = orc.tonemap(hdr, up, want, 1.0 + 0.1 * seed); g = np.zeros_like(tm)
    ref.ref_tonemap(p(hdr), w, h, p(up), up.shape[1], up.shape[0], p(want), 1.0 + 0.1 * seed, p(g)); chk("tonemap", g, tm)
    for srgb in (0, 1):
        f = orc.fxaa(tm, bool(srgb)); g = np.zeros_like(f); ref.ref_fxaa(p(tm), w, h, p(g), srgb); chk("fxaa", g, f)
    for q in range(4):
        e = orc.smaa_edges(tm, q); ge = np.zeros_like(e); fe = getattr(ref, f"ref_smaa_edges_q{q}"); fe.argtypes = [P, C.c_int, C.c_int, P]; fe(p(tm), w, h, p(ge)); chk("smaa_edges", ge, e)
        wt = orc.smaa_weights(e, area, search, q); gw = np.zeros_like(wt); fw = getattr(ref, f"ref_smaa_weights_q{q}"); fw.argtypes = [P, C.c_int, C.c_int, P, P, P]; fw(p(e), w, h, p(area), p(search), p(gw)); chk("smaa_weights", gw, wt)
    ow, oh = int(w * 1.5), int(h * 1.5)
    for fp16, fn in ((False, ref.ref_fsr_easu), (True, ref.ref_fsr_easu_fp16)):
        e = orc.fsr_easu(tm, ow, oh, fp16); g = np.zeros_like(e); fn(p(tm), w, h, p(g), ow, oh); chk("easu" + ("16" if fp16 else "32"), g, e)
    for srgb in (0, 1):
        r = orc.fsr_rcas(e, orc.fsr_rcas_sharpness(0.5), bool(srgb)); g = np.zeros_like(r); ref.ref_fsr_rcas(p(e), ow, oh, p(g), 0.5, srgb); chk("rcas", g, r)
    cam = synth.Camera(w, h); depth = np.ascontiguousarray(synth.make_gbuffer(cam, seed)["depth"], np.float32); mv = np.ascontiguousarray(synth.make_motion_vectors(w, h), np.uint16)
    V2 = synth.look_at((0.01 * seed, 2.0, 8.0), (0.01, 1.0, 0.0)); T = np.eye(4); T[0, 0] = T[1, 1] = 0.5; T[0, 3] = T[1, 3] = 0.5
    reproj = np.ascontiguousarray((T @ (cam.P @ V2) @ cam.invVP).T, np.float32).reshape(16)
    for q in range(3):
        hist = None
        for fr in range(2):
            cur = synth.make_hdr(w, h, seed * 10 + fr)
            wc, wh = orc.taa_resolve(cur, depth, mv, hist, reproj, q); gc, gh = np.zeros_like(wc), np.zeros_like(wh)
            ref.ref_taa_resolve(p(cur), p(depth), p(mv), p(hist), w, h, p(reproj), q, p(gc), p(gh)); chk("taa", gc, wc); chk("taa", gh, wh); hist = wh
print(bad, f"{time.time()-t0:.0f}s")
# ---- single-pass downsampler and HDR10 encode --------------------------------------------------------------------------------------
ref.ref_spd.restype = C.c_int
bad = {}
for seed in range(1, 13):
    iw, ih = 96 + 23 * seed, 64 + 17 * seed
    src = synth.make_hdr(iw, ih, seed)
    w0, h0 = max(iw // 2 - (seed % 3), 1), max(ih // 2 - (seed % 2), 1)
    mips = min(12, max(w0, h0).bit_length())
    for comp, depth, fm in ((4, False, None), (3, False, np.random.default_rng(seed).uniform(0.0, 2.0, (mips, 4)).astype(np.float32)), (1, True, None)):
        a = orc.spd(src, w0, h0, mips, comp, depth, fm, fill=0x3c00)
        b = orc.spd(src, w0, h0, mips, comp, depth, fm, entry=ref.ref_spd, fill=0x3c00)
        bad["spd"] = bad.get("spd", 0) + sum(int((x != y).sum()) for x, y in zip(a, b))
    ref.ref_pq10_encode.argtypes = [P, P, C.c_int, C.c_int, P, C.c_float, C.c_float, C.c_float, P]
    ui = np.random.default_rng(seed).integers(0, 256, (ih, iw, 4), dtype=np.uint8)
    conv = orc.rec709_to_display()
    want = orc.pq10_encode(src, ui, conv, 500.0, 400.0, 250.0 * seed)
    got = np.zeros_like(want)
    ref.ref_pq10_encode(p(src), p(ui), iw, ih, p(conv), 500.0, 400.0, 250.0 * seed, p(got))
    bad["pq10"] = bad.get("pq10", 0) + int((want != got).sum())
print("spd / pq10 soak:", bad, f"{time.time()-t0:.0f}s")
