/* granite_app.h — C ABI of the headless harness that composes Granite's image-space graph on the HIP executor.
 *
 * Mirrors what application/scene_viewer_application.cpp:876-991,1167-1318 (graph composition) and
 * application/platforms/application_headless.cpp:581-671 (warm-up, timed loop, readback) do for the reference, with the
 * scene/mesh renderer replaced by synthetic G-buffer uploads.  Used by bench.py and the parity tests (ctypes); a
 * Granite application would call the C++ classes in granite_amd/csrc/host directly.
 */
#ifndef GRANITE_APP_H_
#define GRANITE_APP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gra_app gra_app;

/* PostAAType (renderer/post/aa.hpp:33-47) subset that is live in the reference. */
typedef enum gra_post_aa
{
	GRA_POST_AA_NONE = 0,
	GRA_POST_AA_FXAA = 1,
	GRA_POST_AA_SMAA_LOW = 2,
	GRA_POST_AA_SMAA_MEDIUM = 3,
	GRA_POST_AA_SMAA_HIGH = 4,
	GRA_POST_AA_SMAA_ULTRA = 5,
	GRA_POST_AA_TAA_LOW = 6,
	GRA_POST_AA_TAA_MEDIUM = 7,
	GRA_POST_AA_TAA_HIGH = 8
} gra_post_aa;

typedef struct gra_config
{
	int32_t device;            /* HIP device index */
	uint32_t width, height;    /* backbuffer = R8G8B8A8_SRGB width x height (application_headless.cpp:207-229) */
	int32_t enable_lighting;   /* 0: config-1 style graph (HDR input -> bloom -> tonemap); 1: gbuffer + clustering + lighting */
	int32_t hdr_bloom;         /* viewer_config "hdrBloom" */
	int32_t dynamic_exposure;  /* "hdrBloomDynamicExposure" */
	int32_t compute_post;      /* ImplementationQuirks::use_async_compute_post: 1 = setup_hdr_postprocess_compute */
	int32_t post_aa;           /* gra_post_aa applied after the post chain (fxaa / smaa) */
	int32_t pre_aa;            /* gra_post_aa applied before the post chain (taa) */
	int32_t rmw_emissive;      /* 1: lighting declared add_color_output("HDR", info, "emissive") exactly like the reference
	                              (G-buffer emissive restored every frame); 0: emissive is a 5th attachment input */
	uint32_t cluster_res[3];   /* LightClusterer::set_resolution; viewer default 128 x 64 x 4096 */
	float frame_time;          /* FrameParameters::frame_time fed to the lerps; headless default 0.01 */
	float directional_color[3];
	float directional_direction[3];
	int32_t enable_timestamps; /* RenderGraph::enable_timestamps */
	/* Row-band tiling of the frame across executors: this instance is rank strip_index of strip_count.
	 * 0 / 1 = the whole frame here.  Needs enable_lighting, hdr_bloom, compute_post and no AA; the band exchanges go
	 * through gra_set_exchange_callback. */
	uint32_t strip_index, strip_count;
	/* 0 (reference behaviour): attachment images of identical geometry with disjoint lifetimes inside the frame share
	 * one allocation (RenderGraph::build_aliases, render_graph.cpp:1548-1746); 1: every image gets its own. */
	int32_t disable_image_aliasing;
	/* Depth hierarchy of the frame's depth attachment (setup_depth_hierarchy_pass, renderer/post/spd.cpp:196-232), resource
	 * "depth-hiz": 0 = none, 1 = full chain, 2 = output_downsample (no full-resolution level).  Needs enable_lighting.
	 * The pass is tied to the end of the frame through a proxy resource, as its consumers (occlusion culling, SSR) live
	 * outside this build. */
	int32_t depth_hierarchy;
	/* viewer_config "resolutionScale" / "resolutionScaleSharpen" (scene_viewer_application.cpp:248-250,758-761,1263-1268):
	 * with 0 < resolution_scale < 1 the G-buffer, lighting, post chain and AA run at ceil(size * scale) and
	 * setup_after_post_chain_upscaling (FSR 1.0 EASU, + RCAS when resolution_scale_sharpen) brings the result to the
	 * backbuffer size; uploads are then render-sized.  0 or 1 = off.  fsr_fp32 = 1 forces the FP16 = 0 shader variant. */
	float resolution_scale;
	int32_t resolution_scale_sharpen;
	int32_t fsr_fp32;
	/* viewer_config "ssao" on the deferred path (scene_viewer_application.cpp:950-980,1571): the lighting pass takes the
	 * R8_UNORM texture "ssao-output-main" as LightingParameters::ambient_occlusion.  Its producer in Granite is FFX CACAO,
	 * shipped as SPIR-V blobs only; the harness fills the texture from gra_upload_ambient_occlusion like the G-buffer. */
	int32_t ambient_occlusion;
	/* HDR10 output (setup_hdr10_pq_encoding, renderer/post/hdr.cpp:595-658): the backbuffer is A2B10G10R10_UNORM_PACK32 and the
	 * frame ends lighting -> "ui" (an R8G8B8A8_SRGB layer cleared to transparent: (0, 0, 0, 1) = scene fully visible) -> "pq10"
	 * with ST.2020 primaries, maxContentLightLevel 1000, hdr / ui pre-exposure 500 / 400 (scene_viewer_application.cpp:1283-1285).
	 * Replaces bloom + tonemap (hdr_bloom must be 0); no AA, no resolution scaling, no row bands.  Note: the viewer itself
	 * declares the UI layer as a colour read-modify-write of the HDR target with another format, which its own validation
	 * rejects (scene_viewer_application.cpp:1279-1287); the layer is its own attachment here. */
	int32_t hdr10;
	/* viewer_config "ssr" on the deferred path (scene_viewer_application.cpp:1206-1212): setup_ssr_pass between lighting and
	 * the post chain -- depth hierarchy "depth-transient-main-hier", "SSR-trace" (classify + trace), "SSR" (apply, blended into
	 * the lit target).  Needs enable_lighting and gra_install_ssr_tables before the first frame. */
	int32_t ssr;
	/* The frame graph of Granite's AA benchmark (tools/aa_bench.cpp:65-161) instead of the viewer's: "main" blits one of two
	 * input images (alternating per frame, gra_upload_aa_bench_images) into "HDR-main" with LinearClamp and clears "depth-main",
	 * pre_aa (TAA) resolves it, "tonemap" blits the result into a swapchain-sized target with NearestClamp, post_aa (FXAA /
	 * SMAA) follows, and with 0 < resolution_scale < 1 FSR 1.0 (+ sharpen) ends the frame.  Identity camera matrices feed the
	 * temporal jitter (aa_bench.cpp:52), timestamps are on.  enable_lighting and hdr_bloom must be 0. */
	int32_t aa_bench;
	/* Row bands: the finished frame's bands meet in every rank's output image as RGB888 (packed before, unpacked after the
	 * all-gather; the alpha byte of a tonemapped / FXAA'd frame is 255) -- 3/4 of the bytes per xGMI link.  1 = send the RGBA8 rows
	 * as they are (A/B).  An SMAA output always travels as RGBA8. */
	int32_t output_gather_rgba;
	/* viewer_config "renderTargetFp16" = false, the reference's shipped default (viewer/viewer_config.json:15,
	 * scene_viewer_application.cpp:881-883): emissive / HDR-main are B10G11R11_UFLOAT_PACK32 instead of R16G16B16A16_SFLOAT
	 * (4 bytes per texel; both lighting blends round to the packed floats), and a TAA resolve in front of the post chain writes
	 * its colour in the same format (temporal.cpp:211-213; its history stays RGBA16F).  The emissive upload is then one
	 * 32-bit word per texel.  0 = RGBA16F everywhere (SURVEY.md 8d fixes the headline configuration on it).  Not with ssr / hdr10
	 * (those passes read-modify-write the lit target as RGBA16F here). */
	int32_t hdr_packed_float;
	/* Row bands with a TAA resolve (no reference analogue): > 0 = how many rows a resolved pixel's history fetch may
	 * lie away from the pixel (vertical motion in rows + 3 for the Catmull-Rom footprint).  The history bands then only exchange
	 * their boundary rows with their neighbours (gra_get_strip_plan_taa_history says how many) instead of meeting whole in every
	 * rank.  A frame in which some pixel reaches further is NOT the single-device frame: the resolve notices (gr_taa_resolve_band),
	 * and the next gra_render_frame / gra_sync / read-back fails with an error that says so.  0 = whole bands are all-gathered
	 * (any motion).  Every rank must be given the same value. */
	uint32_t taa_history_reach_rows;
} gra_config;

/* Scene-level light description (one PositionalLight + its node transform). */
typedef struct gra_light_desc
{
	int32_t type;            /* 0 = spot, 1 = point */
	float color[3];
	float inner_cone, outer_cone;
	float cutoff_range;      /* PositionalLight::set_maximum_range */
	float pad;
	float transform[12];     /* mat_affine rows */
} gra_light_desc;

/* RenderParameters subset, tightly packed: projection, view, view_projection, inv_projection, inv_view,
 * inv_view_projection (column-major mat4 each), camera_position[3], camera_front[3], z_near, z_far = 104 floats. */
#define GRA_RENDER_PARAMS_FLOATS 104

gra_app *gra_create(const gra_config *config, char *error, size_t error_size);
void gra_destroy(gra_app *app);
const char *gra_last_error(gra_app *app);

/* RenderContext::set_camera(projection, view): derives inverses etc. on the host like the reference. */
int gra_set_camera(gra_app *app, const float *projection16, const float *view16);
/* The eye moves by `translation` (world units) every frame, applied to the view matrix of gra_set_camera before each frame
 * (BASELINE config 4: "camera translates 0.01 units/frame" -- the TAA reprojection and the cluster build then really change from
 * frame to frame).  (0, 0, 0) = a static camera.  Needs gra_set_camera. */
int gra_set_camera_motion(gra_app *app, const float translation[3]);
/* Installs precomputed parameters verbatim (so that the CPU oracle and the device see bit-identical inputs). */
int gra_set_render_parameters(gra_app *app, const float *params104);
int gra_get_render_parameters(gra_app *app, float *params104);
/* Size the G-buffer is rendered (and uploaded) at: the backbuffer size unless resolution_scale is in use. */
int gra_get_render_size(gra_app *app, uint32_t *width, uint32_t *height);

int gra_set_lights(gra_app *app, const gra_light_desc *lights, uint32_t count);

/* Synthetic G-buffer (host pointers, tightly packed rows, width x height of the config).  Any pointer may be NULL to
 * leave that attachment unchanged.  Graphs without lighting take `emissive` as the HDR input. */
int gra_upload_gbuffer(gra_app *app, const void *emissive_rgba16f, const void *albedo_rgba8, const void *normal_a2b10g10r10,
                       const void *pbr_rg8, const void *depth_d32f, const void *motion_vectors_rg16f);

/* Render-sized R8_UNORM ambient-occlusion image (host pointer, tightly packed); needs config.ambient_occlusion. */
/* The two input images of the aa_bench graph: R8G8B8A8_SRGB, tightly packed, any size (both the same). */
int gra_upload_aa_bench_images(gra_app *app, const void *rgba8_first, const void *rgba8_second, uint32_t width, uint32_t height);
int gra_upload_ambient_occlusion(gra_app *app, const void *ao_r8);

/* ---- GTX ("GRANITE TEXFMT1", vulkan/texture/memory_mapped_texture.cpp:29-44): the container Granite keeps textures and
 * image dumps in -- the wire format on either side of this path.  Header fields as stored; payload = mip levels in order,
 * each at a 16-byte aligned offset (vulkan/texture/texture_format.cpp:349-387). */
typedef struct gra_gtx_info
{
	uint32_t type, format, width, height, depth, layers, levels, flags;
	uint64_t payload_size;
} gra_gtx_info;
int gra_gtx_probe(const char *path, gra_gtx_info *info, char *error, size_t error_size);
int gra_gtx_read(const char *path, void *payload, uint64_t payload_capacity, char *error, size_t error_size);
int gra_gtx_write(const char *path, const gra_gtx_info *info, const void *payload, char *error, size_t error_size);
/* gra_upload_gbuffer from .gtx files (2-D, one layer, level 0 is used).  Formats must be the attachment's own:
 * emissive / HDR R16G16B16A16_SFLOAT, albedo R8G8B8A8_SRGB or _UNORM, normal A2B10G10R10_UNORM_PACK32, pbr R8G8_UNORM,
 * depth D32_SFLOAT or R32_SFLOAT, motion vectors R16G16_SFLOAT; sizes must equal the configured frame.  NULL = unchanged. */
int gra_upload_gbuffer_gtx(gra_app *app, const char *emissive, const char *albedo, const char *normal, const char *pbr,
                           const char *depth, const char *motion_vectors);
/* Writes a graph texture (all its mip levels) or, with name == NULL, the last rendered backbuffer as .gtx. */
int gra_save_resource_gtx(gra_app *app, const char *name, const char *path);

/* Application::run_frame x count; asynchronous unless sync != 0. */
int gra_render_frames(gra_app *app, uint32_t count, int32_t sync);
int gra_sync(gra_app *app);

/* Resource access by render-graph name ("HDR-main", "tonemapped", "upsample-0", "cluster-bitmask", ...). */
typedef struct gra_resource_info
{
	void *device_ptr;
	uint32_t width, height, format; /* 0 x 0 for buffers */
	uint64_t size_bytes;            /* whole mip chain when levels > 1 (gr_mip_chain_size) */
	int32_t physical_index;
	uint32_t levels;                /* 1 unless the attachment was declared with a mip chain; 0 for buffers */
} gra_resource_info;
int gra_get_resource(gra_app *app, const char *name, gra_resource_info *info);
int gra_read_resource(gra_app *app, const char *name, void *dst_host, uint64_t size_bytes);
/* The write-side twin of gra_read_resource: replaces the contents of a graph resource with `size_bytes` of host data (the whole
 * resource).  Meant for the state a frame inherits from the one before it -- "average-luminance" (exposure adaptation),
 * "downsample-3" (bloom feedback), "<output>-history" of a TAA resolve: what gra_read_resource returned after frame N, written into
 * a fresh application before its first frame (or into a running one), is what frame N + 1 reads as "previous" (persistent
 * buffers in place; an image with a history copy becomes the history at the next frame's swap, render_graph.cpp:2704-2708).
 * Together with gra_get / set_frame_state a run can be resumed bit for bit (checkpoint / replay). */
int gra_write_resource(gra_app *app, const char *name, const void *src_host, uint64_t size_bytes);
/* Host-side state a frame inherits: elapsed time, swapchain ring position, the (moving) camera's view matrix and the temporal
 * jitter's phase with its saved view-projection ring (temporal.cpp:40-196). */
typedef struct gra_frame_state
{
	uint64_t frames;
	double elapsed;
	uint32_t swapchain_index;
	uint32_t jitter_phase;
	float base_view[16];
	float jittered_projection[16];
	float view_proj[16][16], inv_view_proj[16][16], jittered_view_proj[16][16]; /* first jitter_count entries */
} gra_frame_state;
int gra_get_frame_state(gra_app *app, gra_frame_state *state);
int gra_set_frame_state(gra_app *app, const gra_frame_state *state);
/* The swapchain image the last frame was rendered into (external to the graph, 4-image ring). */
int gra_get_backbuffer(gra_app *app, gra_resource_info *info);
int gra_read_backbuffer(gra_app *app, void *dst_host, uint64_t size_bytes);

/* Clusterer CPU-side state of the last refresh, for parity tests.  Returns the number of lights. */
int gra_get_cluster_state(gra_app *app, void *lights48, void *models48, uint32_t *type_mask128, void *params176,
                          uint32_t *light_ranges_uvec2);

/* JSON dump of the baked graph (pass order, physical resources).  Returns the length needed. */
size_t gra_dump_graph(gra_app *app, char *buffer, size_t size);

/* GPU time per physical pass accumulated since creation or the last gra_reset_timestamps (enable_timestamps): tags are the
 * reference's (render_graph.cpp:2274-2289, "a + b" for passes it runs as subpasses of one render pass), one accumulation per
 * physical pass and frame.  Fills up to max entries, returns the count. */
typedef struct gra_timestamp
{
	char tag[64];
	uint64_t count;
	double total_ms;
} gra_timestamp;
int gra_collect_timestamps(gra_app *app, gra_timestamp *entries, int max_entries);
/* emit_single_pass_downsample (renderer/post/spd.cpp:56-102) the way the reference uses it (renderer/ocean.cpp:579-601): an
 * RGBA16F image of `levels` mip levels whose level 0 is the source and whose levels 1.. are written by the single-pass
 * downsampler (`components` channels, optional per-level filter_mods = (levels - 1) x vec4).  level0: width x height texels in;
 * chain: all levels out, tightly packed (gr_mip_chain_offset).  Runs on the application's device. */
int gra_generate_mipmaps(gra_app *app, const void *level0_rgba16f, uint32_t width, uint32_t height, uint32_t levels,
                         uint32_t components, const float *filter_mods, void *chain_rgba16f);

/* Device::timestamp_log_reset (application_headless.cpp:591): forget what was accumulated, e.g. the warm-up frame. */
int gra_reset_timestamps(gra_app *app);
/* The scene's DirectionalLightComponent (read_lights, scene_viewer_application.cpp:58-77): replaces the values of
 * gra_config for the frames that follow; direction is normalised here. */
int gra_set_directional_light(gra_app *app, const float direction[3], const float color[3]);
/* LightingParameters::fog (renderer/lights/lights.hpp; the scene's "fog" entry in Granite): with falloff > 0 render_light ends with
 * the fog quad (renderer.cpp:1179-1196).  falloff = 0 switches it off again. */
int gra_set_fog(gra_app *app, const float color[3], float falloff);
/* T(.5,.5,0) S(.5,.5,1) VP_prev inv(VP_cur) as pushed to the last taa-resolve (temporal.cpp:239-243). */
int gra_get_taa_reprojection(gra_app *app, float *reproj16);
/* SMAA AreaTex (160x560 RG8) / SearchTex (64x16 R8) payloads, host pointers; needed before a frame with an SMAA pass. */
int gra_set_smaa_luts(gra_app *app, const void *area_rg8, const void *search_r8);
/* Row-band tiling.  The executor calls `fn` where the bands of all ranks must meet (after the 1/8 bloom level, after
 * tonemap): an in-place all-gather of `rank_count` chunks of `chunk_bytes` bytes laid out back to back from `device_ptr`
 * (this rank's chunk already sits at device_ptr + strip_index * chunk_bytes), to be enqueued on `stream` (hipStream_t).
 * bench.py implements it with torch.distributed (RCCL); tests with local copies. */
typedef void (*gra_exchange_fn)(void *user, const char *tag, void *device_ptr, uint64_t chunk_bytes, uint32_t rank_count, void *stream);
int gra_set_exchange_callback(gra_app *app, gra_exchange_fn fn, void *user);
/* RCCL transport for the band exchanges: rank 0 creates a 128-byte id (gra_comm_create_unique_id), the launcher ships it
 * to every rank (bench.py: torch.distributed broadcast), every rank calls gra_comm_init; from then on the exchange points
 * are in-place ncclAllGather calls on the executor's stream (xGMI between the GPUs of one node).  ranks must equal the
 * config's strip_count and rank its strip_index. */
int gra_comm_create_unique_id(uint8_t *id128);
int gra_comm_init(gra_app *app, const uint8_t *id128, int32_t rank, int32_t ranks);
/* Optional second communicator (its own id from gra_comm_create_unique_id, after gra_comm_init): the all-gather of the
 * tonemapped bands then runs on a stream of its own behind each frame's tonemap and overlaps the following frames instead of
 * sitting on the back-of-frame stream.  gra_sync / readbacks wait for it. */
/* What the in-frame communicator reports about itself, for run records: nranks = ncclCommCount, version = ncclGetVersion (each -1
 * when the loaded library has no such entry point), stand_in = 1 when GRANITE_RCCL_LIBRARY replaced librccl.so.1. */
int gra_comm_info(gra_app *app, int32_t *nranks, int32_t *version, int32_t *stand_in);
int gra_comm_init_output(gra_app *app, const uint8_t *id128, int32_t rank, int32_t ranks);
/* The band plan of this instance: out[0..3] = index, count, width, height; then {whole, first, count} for lighting,
 * threshold, downsample-0, downsample-1, upsample-0, tonemap; then d1_chunk_rows, out_chunk_rows (24 values). */
int gra_get_strip_plan(gra_app *app, uint32_t *out24);
/* The anti-aliasing part of the plan (config.pre_aa / post_aa under row bands): {whole, first, count} for the TAA resolve,
 * smaa-edge, smaa-weights and the post-AA output band (12 values).  The tonemap and lighting ranges of gra_get_strip_plan
 * already include the rows these passes read around the band. */
int gra_get_strip_plan_aa(gra_app *app, uint32_t *out12);
/* The TAA history under row bands: out[0] = config.taa_history_reach_rows, out[1] = rows every rank hands to each neighbour per
 * frame (0 = the bands are all-gathered whole: reach 0, or bands thinner than the exchange), out[2..4] = {whole, first, count} of
 * the history rows this rank holds after the exchange. */
int gra_get_strip_plan_taa_history(gra_app *app, uint32_t *out5);

/* Host-side frame-loop cost since creation: out[0] = frames, out[1] = seconds spent inside the frame loop (light
 * refresh + graph execution = launches), out[2] = seconds of those spent blocked on GPU back-pressure. */
/* The constant tables of the SSR pass (process-wide): the blue-noise sampler's 128 x 128 x 2 integer values and the
 * R16G16_SFLOAT split-sum BRDF table (width x height texels); see csrc/host/post/ssr.hpp. */
int gra_install_ssr_tables(const uint8_t *blue_noise_128x128_rg8, const uint16_t *brdf_lut_rg16f, uint32_t brdf_width, uint32_t brdf_height);
int gra_get_host_stats(gra_app *app, double *out3);
/* Row bands with the output gather beside the frame (gra_comm_init_output): out[0] = times a pass was about to overwrite an output
 * image, out[1] = of those, how often the all-gather that last read it was still in flight (the pass waits: the gather was not hidden
 * behind the following frames).  The collectives' device times are gr_timing_* names "inframe_gather" / "output_gather". */
int gra_get_output_gather_stats(gra_app *app, uint64_t *out2);
/* Frames whose light sort + pack (LightClusterer::refresh) had already been done by the clusterer's helper thread while the
 * previous frame was being enqueued (the reference runs its refreshes as TaskComposer tasks beside command recording). */
int gra_get_prefetched_refreshes(gra_app *app, uint64_t *out);
/* Pre-recorded launch sequences replayed so far (HIP::CommandBuffer::replayable, opt-in with GRANITE_LAUNCH_GRAPHS=1: the bloom
 * pass's six launches and the cluster build's four go out as one hipGraph launch each once their arguments repeat; less host
 * time, but a slower frame on this runtime -- see hip_device.hpp -- so the default launches every kernel directly). */
int gra_get_launch_graph_replays(gra_app *app, uint64_t *out);
/* compute_rec709_to_st2020 (hdr.cpp:580-593) for display primaries r, g, b, white (CIE xy, 8 floats): column-major 3 x 3. */
int gra_compute_rec709_to_display(const float *primaries8, float *out9);
/* Bytes of HBM currently held by the executor's images and buffers (graph attachments, hand-over rings, uploads). */
int gra_get_allocated_bytes(gra_app *app, uint64_t *out);
/* Per-kernel timing lives in the kernel library: gr_timing_* on this context. */
void *gra_get_kernel_context(gra_app *app); /* gr_ctx* */
void *gra_get_stream(gra_app *app);         /* hipStream_t of the generic queue */

#ifdef __cplusplus
}
#endif
#endif
