/* granite_hip.h — C ABI of the MI355X (gfx950) executor for Granite's image-space chain.
 *
 * Granite has no C plugin ABI: its operator interface for this path is the C++ RenderGraph pass API
 * (renderer/render_graph.hpp:488-716,779-893) whose callbacks record GLSL dispatches / full-screen quads on a
 * Vulkan::CommandBuffer.  This header is what a HIP backend for those callbacks binds instead: one entry point per
 * shader dispatch on the hot path, taking plain pointers + sizes (no C++/torch types), with push-constant structs that
 * are byte-identical to the reference's.  Each entry point cites the reference call site it replaces.
 *
 * Conventions
 *   - All image/buffer pointers are DEVICE pointers into linear, row-major HBM buffers (origin top-left).
 *   - Every launcher is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream).
 *   - Return value: 0 on success, negative gr_status on failure; gr_last_error() holds the message.
 *     No exceptions cross this boundary.  Thread-safe per (ctx, stream).
 */
#ifndef GRANITE_HIP_H_
#define GRANITE_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GR_ABI_VERSION 1

typedef struct gr_ctx gr_ctx;
typedef void *gr_stream; /* hipStream_t */

typedef enum gr_status
{
	GR_OK = 0,
	GR_ERR_INVALID_ARGUMENT = -1,
	GR_ERR_HIP = -2,
	GR_ERR_UNSUPPORTED_FORMAT = -3,
	GR_ERR_OUT_OF_MEMORY = -4
} gr_status;

/* Subset of VkFormat used on the path; numeric values are VkFormat's so reference call sites read the same. */
typedef enum gr_format
{
	GR_FORMAT_UNDEFINED = 0,
	GR_FORMAT_R8_UNORM = 9,
	GR_FORMAT_R8G8_UNORM = 16,
	GR_FORMAT_R8G8B8A8_UNORM = 37,
	GR_FORMAT_R8G8B8A8_SRGB = 43,
	GR_FORMAT_A2B10G10R10_UNORM_PACK32 = 64,
	GR_FORMAT_R16_SFLOAT = 76,
	GR_FORMAT_R16G16_SFLOAT = 83,
	GR_FORMAT_R16G16B16A16_SFLOAT = 97,
	GR_FORMAT_R32_SFLOAT = 100,
	GR_FORMAT_B10G11R11_UFLOAT_PACK32 = 122, /* HDR targets with renderTargetFp16 = false (scene_viewer_application.cpp:881-883), TAA colour (temporal.cpp:211-213) */
	GR_FORMAT_D16_UNORM = 124,
	GR_FORMAT_D32_SFLOAT = 126
} gr_format;

/* A 2-D attachment as the executor sees it: what Vulkan::ImageView is to the reference's callbacks. */
typedef struct gr_image
{
	void *ptr;            /* device pointer */
	uint32_t width;
	uint32_t height;
	uint32_t pitch_bytes; /* row pitch; rows are tightly packed unless stated */
	uint32_t format;      /* gr_format */
} gr_image;

/* ---- context ----------------------------------------------------------------------------------------------------- */
gr_ctx *gr_create(int device);          /* replaces Vulkan::Context/Device creation for this path */
void gr_destroy(gr_ctx *ctx);
const char *gr_last_error(gr_ctx *ctx);
int gr_abi_version(void);
int gr_sync(gr_ctx *ctx, gr_stream stream); /* hipStreamSynchronize; Device::wait_idle analogue */

/* Physical-resource allocation used by RenderGraph::setup_physical_{image,buffer} (render_graph.cpp:2577-2684).
 * Memory is zero-initialised like the reference's (render_graph.cpp:2586-2587). */
int gr_alloc(gr_ctx *ctx, size_t bytes, void **dptr);
int gr_free(gr_ctx *ctx, void *dptr);
int gr_upload(gr_ctx *ctx, gr_stream stream, void *dst, const void *src_host, size_t bytes);   /* cmd.update_buffer */
int gr_download(gr_ctx *ctx, gr_stream stream, void *dst_host, const void *src, size_t bytes); /* readback */
/* Several cmd.update_buffer calls of one pass (clusterer.cpp:1178-1207,1302) as ONE kernel that reads pinned host memory
 * (hipHostMalloc'd, device-mapped) and writes HBM: no copy engine, no cross-engine signalling, ordered like any other
 * kernel on `stream`.  Sizes and offsets must be multiples of 4 bytes; at most GR_MAX_UPLOAD_RANGES ranges. */
#define GR_MAX_UPLOAD_RANGES 8
typedef struct gr_upload_range
{
	void *dst;             /* device */
	const void *src_pinned; /* pinned host */
	size_t bytes;
} gr_upload_range;
int gr_upload_batch(gr_ctx *ctx, gr_stream stream, const gr_upload_range *ranges, uint32_t count);
/* Pinned, device-mapped host memory for gr_upload_batch sources (hipHostMalloc / hipHostFree). */
int gr_alloc_host(gr_ctx *ctx, size_t bytes, void **hptr);
int gr_free_host(gr_ctx *ctx, void *hptr);
int gr_copy(gr_ctx *ctx, gr_stream stream, void *dst, const void *src, size_t bytes);
int gr_fill_zero(gr_ctx *ctx, gr_stream stream, void *dst, size_t bytes);
int gr_fill_byte(gr_ctx *ctx, gr_stream stream, void *dst, int value, size_t bytes); /* clear to a UNORM8 constant */

/* Host-only: the linear -> sRGB8 staircase table the *_SRGB stores use (count must be 1665 {threshold bits, byte} pairs;
 * csrc/device_common.hpp).  Exposed so that the table can be checked against the transfer function without a GPU. */
int gr_srgb_encode_table(uint32_t *entries_xy, uint32_t count);
/* Host-only: the exposed colour -> tonemapped sRGB8 staircase of gr_tonemap's *_SRGB path (1025 pairs). */
int gr_tonemap_srgb8_table(uint32_t *entries_xy, uint32_t count);

/* Measured HBM ceiling for the roofline ("a hipMemcpyDtoD / stream-triad probe on the box, in the same
 * run"): a float4 copy b = a and a triad a = b + s * c over `bytes`-sized arrays that do not fit the 256 MiB Infinity
 * Cache, best of `repeats` launches timed with hipEvents.  GB/s counts bytes read + written (copy 2 x, triad 3 x bytes). */
int gr_bandwidth_probe(gr_ctx *ctx, size_t bytes, int repeats, double *copy_GBps, double *triad_GBps);

/* Per-kernel GPU timing (RenderGraph::enable_timestamps analogue, render_graph.cpp:2196-2310): when enabled, every
 * launcher brackets its kernel with hipEvents on the launch stream; gr_timing_query drains {name, count, total_ms}. */
typedef struct gr_timing_entry
{
	const char *name;
	uint64_t count;
	double total_ms;
} gr_timing_entry;
int gr_timing_enable(gr_ctx *ctx, int enable);
/* Restrict bracketing to launchers whose name equals `name` (NULL = all): keeps event overhead out of a timed loop. */
int gr_timing_set_filter(gr_ctx *ctx, const char *name);
/* 1 when a launch of the kernel timed under `name` would currently get a hipEvent bracket (timing enabled and the filter, if any,
 * names it): a pre-recorded launch sequence containing it must be launched directly instead. */
int gr_timing_brackets(gr_ctx *ctx, const char *name);
/* Bracket only every n-th matching launch (1 = all).  An event pair around a kernel keeps the command processor from
 * overlapping that launch with its neighbours on the stream; sampling keeps a timed loop close to its unbracketed speed
 * while the launch duration is still measured live inside it. */
int gr_timing_set_sampling(gr_ctx *ctx, uint32_t every_nth);
/* Restrict bracketing to several launchers: gr_timing_set_filter also takes a comma-separated list ("lighting,output_gather").
 * gr_timing_span_begin / _end bracket work that is not one of this library's launches (a collective on its stream) under the caller's
 * name, into the same accumulator; *span is NULL when the name is not being timed (then _end is a no-op).  gr_timing_max_ms: the
 * longest single bracket under a name since the last reset. */
int gr_timing_span_begin(gr_ctx *ctx, gr_stream stream, const char *name, void **span);
int gr_timing_span_end(gr_ctx *ctx, gr_stream stream, void *span);
int gr_timing_max_ms(gr_ctx *ctx, const char *name, double *max_ms);
int gr_timing_reset(gr_ctx *ctx);
int gr_timing_query(gr_ctx *ctx, gr_timing_entry *entries, int max_entries); /* returns number of entries; syncs */

/* Render area of one launch: output rows [first, first + count) only.  A NULL pointer = the whole image; count == 0 = no rows
 * (the launcher returns GR_OK without launching: a rank whose band is empty must not touch the image).  The one embedded use,
 * gr_lighting_args.rows, keeps {0, 0} = whole image, as a zero-initialised argument struct has it.
 * The reference restricts draws with VkRect2D render areas / scissors (vulkan/command_buffer.cpp set_scissor); the
 * executor uses row bands to tile one frame across GPUs.  Coordinates stay those of the full image, so a
 * band computes bit-identical values to the same rows of a whole-image launch. */
typedef struct gr_rows
{
	uint32_t first;
	uint32_t count;
} gr_rows;

/* ---- HDR post chain (renderer/post/hdr.cpp) ---------------------------------------------------------------------- */

/* LuminanceData, 12 B (assets/shaders/post/luminance.comp:4-9): {log2 avg, avg, 1/avg}. */
typedef struct gr_luminance_data
{
	float average_log_luminance;
	float average_linear_luminance;
	float average_inv_linear_luminance;
} gr_luminance_data;

/* bloom_threshold_build_compute (hdr.cpp:115-144) + bloom_threshold.comp.  lum = device LuminanceData or NULL
 * (DYNAMIC_EXPOSURE=0).  hdr, out: R16G16B16A16_SFLOAT. */
typedef struct gr_push_bloom_threshold
{
	uint32_t threads[2];
	float inv_output_size[2];
} gr_push_bloom_threshold;
int gr_bloom_threshold(gr_ctx *ctx, gr_stream stream, const gr_image *hdr, const gr_image *out,
                       const gr_luminance_data *lum, const gr_push_bloom_threshold *push);
int gr_bloom_threshold_rows(gr_ctx *ctx, gr_stream stream, const gr_image *hdr, const gr_image *out,
                            const gr_luminance_data *lum, const gr_push_bloom_threshold *push, const gr_rows *rows);

/* bloom_downsample_build_compute (hdr.cpp:146-187) + bloom_downsample.comp.  history = previous frame's output
 * (FEEDBACK=1, NearestClamp) or NULL. */
typedef struct gr_push_bloom_downsample
{
	uint32_t threads[2];
	float inv_output_size[2];
	float inv_input_size[2];
	float lerp;
} gr_push_bloom_downsample;
int gr_bloom_downsample(gr_ctx *ctx, gr_stream stream, const gr_image *in, const gr_image *out,
                        const gr_image *history, const gr_push_bloom_downsample *push);
int gr_bloom_downsample_rows(gr_ctx *ctx, gr_stream stream, const gr_image *in, const gr_image *out,
                             const gr_image *history, const gr_push_bloom_downsample *push, const gr_rows *rows);

/* bloom_upsample_build_compute (hdr.cpp:189-216) + bloom_upsample.comp. */
typedef struct gr_push_bloom_upsample
{
	uint32_t threads[2];
	float inv_output_size[2];
	float inv_input_size[2];
} gr_push_bloom_upsample;
int gr_bloom_upsample(gr_ctx *ctx, gr_stream stream, const gr_image *in, const gr_image *out,
                      const gr_push_bloom_upsample *push);
int gr_bloom_upsample_rows(gr_ctx *ctx, gr_stream stream, const gr_image *in, const gr_image *out,
                           const gr_push_bloom_upsample *push, const gr_rows *rows);

/* luminance_build_compute (hdr.cpp:68-98) + luminance.comp: mean log-luminance of `in`.a over a size.x x size.y
 * bilinear grid, clamp, temporal lerp, read-modify-write of *lum. */
typedef struct gr_push_luminance
{
	uint32_t size[2];
	float lerp;
	float min_loglum;
	float max_loglum;
} gr_push_luminance;
int gr_luminance(gr_ctx *ctx, gr_stream stream, const gr_image *in, gr_luminance_data *lum,
                 const gr_push_luminance *push);

/* The coarse end of the pyramid as recorded by hdr.cpp:364-377 -- downsample-2, downsample-3 (+ feedback), luminance,
 * upsample-2, upsample-1 -- in two launches instead of five: every texel is computed by the same code as the separate entry
 * points above (values and fp16 roundings between levels are identical), the intermediate levels are still written.
 * gr_bloom_tail_supported() says whether a pyramid qualifies (whole levels, each level ceil(half) of the one above as
 * render_graph.cpp's InputRelative sizes are: 4K, 1080p, odd sizes alike); otherwise the five separate calls are the path. */
int gr_bloom_tail_supported(const gr_image *d1, const gr_image *d2, const gr_image *d3, const gr_image *u2, const gr_image *u1,
                            const gr_push_bloom_downsample *push_d2, const gr_push_bloom_downsample *push_d3,
                            const gr_push_bloom_upsample *push_u2, const gr_push_bloom_upsample *push_u1);
int gr_bloom_down_tail(gr_ctx *ctx, gr_stream stream, const gr_image *d1, const gr_image *d2, const gr_image *d3,
                       const gr_image *history, const gr_push_bloom_downsample *push_d2,
                       const gr_push_bloom_downsample *push_d3);
/* downsample-0 and downsample-1 (hdr.cpp:358-362) in one launch, the same way: a workgroup makes an 8 x 8 tile of downsample-1 from
 * the patch of downsample-0 under its taps, which it first makes from the threshold level and stores.  rows_d1 restricts
 * downsample-1 (row bands; downsample-0 is written under those rows' taps).  Values in both levels are those of two
 * gr_bloom_downsample calls, byte for byte.  gr_bloom_down_mid_supported(): whole levels whose patch fits (any pyramid of
 * InputRelative sizes) and a downsample-1 of at most 65536 texels -- frames up to 1440p, where the chain of dependent launches sets
 * the pace; above, the recomputed overlap costs more than the launch it saves (measured at 4K). */
int gr_bloom_down_mid_supported(const gr_image *threshold, const gr_image *d0, const gr_image *d1, const gr_push_bloom_downsample *push_d0,
                                const gr_push_bloom_downsample *push_d1);
int gr_bloom_down_mid(gr_ctx *ctx, gr_stream stream, const gr_image *threshold, const gr_image *d0, const gr_image *d1,
                      const gr_push_bloom_downsample *push_d0, const gr_push_bloom_downsample *push_d1, const gr_rows *rows_d1);
/* The threshold dispatch with downsample-0 and downsample-1 (hdr.cpp:354-362) in one launch: a workgroup makes an 8 x 8 tile of
 * downsample-1, under it the patch of downsample-0, under that the patch of the threshold level from the HDR target, every texel by the
 * code of the separate entry points (all three levels are written, byte for byte what gr_bloom_threshold + gr_bloom_down_mid leave).
 * gr_bloom_down_head_supported(): what gr_bloom_down_mid_supported() asks for, and every level exactly half of its input (the 2:1 forms
 * of gr_bloom_threshold and gr_bloom_downsample).  lum NULL: no dynamic exposure.  Whole images (row bands use the separate calls). */
int gr_bloom_down_head_supported(const gr_image *hdr, const gr_image *threshold, const gr_image *d0, const gr_image *d1,
                                 const gr_push_bloom_threshold *push_t, const gr_push_bloom_downsample *push_d0,
                                 const gr_push_bloom_downsample *push_d1);
int gr_bloom_down_head(gr_ctx *ctx, gr_stream stream, const gr_image *hdr, const gr_image *threshold, const gr_image *d0, const gr_image *d1,
                       const gr_luminance_data *lum, const gr_push_bloom_threshold *push_t, const gr_push_bloom_downsample *push_d0,
                       const gr_push_bloom_downsample *push_d1);
/* lum / push_lum both NULL: no dynamic exposure, no luminance reduction. */
int gr_bloom_up_tail(gr_ctx *ctx, gr_stream stream, const gr_image *d3, const gr_image *u2, const gr_image *u1,
                     gr_luminance_data *lum, const gr_push_bloom_upsample *push_u2, const gr_push_bloom_upsample *push_u1,
                     const gr_push_luminance *push_lum);

/* The whole upsample chain -- luminance, upsample-2, upsample-1, upsample-0 (hdr.cpp:368-379) -- in one launch: a workgroup makes a 32 x 32
 * tile of upsample-0 from the patch of upsample-1 under it, that from the patch of upsample-2, that from downsample-3; all three levels are
 * written, byte for byte what gr_bloom_up_tail + gr_bloom_upsample leave.  gr_bloom_up_all_supported(): whole levels of a pyramid of
 * InputRelative sizes, upsample-0 exactly twice upsample-1 and at most 960 x 540 (a 4K frame: above, the tiles' recomputed patches cost more than the launch saved).  lum / push_lum both NULL: no
 * dynamic exposure, no luminance reduction. */
int gr_bloom_up_all_supported(const gr_image *d3, const gr_image *u2, const gr_image *u1, const gr_image *u0, const gr_push_bloom_upsample *push_u2,
                              const gr_push_bloom_upsample *push_u1, const gr_push_bloom_upsample *push_u0);
/* flags: GR_BLOOM_BUSY_FRAME_BIT -- scheduling hint, no effect on any value: the rest of the frame's back (a temporal resolve, SMAA) fills the
 * machine beside the lighting launch.  The launch then runs 256-thread workgroups (one wave per SIMD: a workgroup starts wherever one lighting
 * workgroup has retired) instead of 1024-thread ones (which wait for four retirements on one CU and so end up in the idle time between two lighting
 * launches: best for a frame whose back is short).  Measured (profiles/r06_back_chain_beside_lighting.txt): the launch inside the 4K frame 23
 * against 64-68 us; the TAA + SMAA frame 0.548 against 0.558-0.573 ms, the lighting + bloom + tonemap frame 0.204 against 0.199 ms. */
#define GR_BLOOM_BUSY_FRAME_BIT 1u
int gr_bloom_up_all(gr_ctx *ctx, gr_stream stream, const gr_image *d3, const gr_image *u2, const gr_image *u1, const gr_image *u0,
                    gr_luminance_data *lum, const gr_push_bloom_upsample *push_u2, const gr_push_bloom_upsample *push_u1,
                    const gr_push_bloom_upsample *push_u0, const gr_push_luminance *push_lum, uint32_t flags);

/* The WHOLE bloom pass of a small frame -- every dispatch of bloom_build_compute (hdr.cpp:354-379): threshold, downsample-0 .. -3 (+ feedback),
 * luminance, upsample-2 .. -0 -- in ONE launch: the work of gr_bloom_down_head, gr_bloom_down_tail and gr_bloom_up_all as three block ranges of one
 * grid, a range starting when the one before it has published its levels (device-side counters in the context; workgroups wait for lower
 * workgroup ids only).  Every level is written, byte for byte what the separate launches leave.  gr_bloom_pyramid_supported(): what the three
 * launches require (whole levels of a pyramid of InputRelative sizes, threshold / downsample-0 / -1 exactly half of their inputs, upsample-0
 * exactly twice upsample-1, a feedback history) and a frame of at most 640 x 384 pixels -- where the host's runtime calls, not the device, set
 * the frame's pace; above, the device-side hand-overs between the ranges cost more than the launches they replace (gr_bloom_pyramid itself
 * accepts any size the three launches accept).  lum NULL: static exposure, no luminance reduction.  Launches of this entry point on one context must not
 * overlap in time (issue them on one stream, as the executor does); gr_debug_pyramid_giveups() synchronises the device and returns how many
 * workgroups ever gave up waiting for a predecessor (0 unless that rule was broken). */
typedef struct gr_bloom_pyramid_args
{
	gr_image hdr, threshold, d0, d1, d2, d3, history, u2, u1, u0;
	gr_luminance_data *lum;
	gr_push_bloom_threshold push_threshold;
	gr_push_bloom_downsample push_d0, push_d1, push_d2, push_d3;
	gr_push_bloom_upsample push_u2, push_u1, push_u0;
	gr_push_luminance push_luminance;
} gr_bloom_pyramid_args;
int gr_bloom_pyramid_supported(const gr_bloom_pyramid_args *args);
int gr_bloom_pyramid(gr_ctx *ctx, gr_stream stream, const gr_bloom_pyramid_args *args);
int gr_debug_pyramid_giveups(gr_ctx *ctx, uint32_t *count);

/* tonemap_build_render_pass (hdr.cpp:283-306) + tonemap.frag (full-screen quad).  out: R8G8B8A8_SRGB (linear value
 * is sRGB-encoded on store, as the attachment hardware does) or R8G8B8A8_UNORM.  lum NULL => DYNAMIC_EXPOSURE=0. */
typedef struct gr_push_tonemap
{
	float dynamic_exposure;
} gr_push_tonemap;
int gr_tonemap(gr_ctx *ctx, gr_stream stream, const gr_image *hdr, const gr_image *bloom, const gr_image *out,
               const gr_luminance_data *lum, const gr_push_tonemap *push);
int gr_tonemap_rows(gr_ctx *ctx, gr_stream stream, const gr_image *hdr, const gr_image *bloom, const gr_image *out,
                    const gr_luminance_data *lum, const gr_push_tonemap *push, const gr_rows *rows);

/* ---- clustered lighting (renderer/lights/clusterer.cpp, renderer/renderer.cpp) ------------------------------------- */

/* PositionalFragmentInfo (renderer/lights/light_info.hpp:35-44), 48 B. */
typedef struct gr_light_info
{
	float color[3];
	uint32_t spot_scale_bias; /* 2 x fp16: scale | bias << 16 */
	float position[3];
	uint32_t offset_radius;   /* 2 x fp16: offset | radius << 16 */
	float direction[3];
	float inv_radius;
} gr_light_info;

/* mat_affine (math/muglm/muglm.hpp:899-958): 3 rows of vec4. */
typedef struct gr_mat_affine
{
	float rows[3][4];
} gr_mat_affine;

/* ClustererParametersBindless, std140, 176 B (assets/shaders/lights/clusterer_data.h:20-39; math/render_parameters.hpp:90-108). */
typedef struct gr_cluster_params
{
	float transform[16];
	float clip_scale[4];
	float camera_base[3];
	float pad0;
	float camera_front[3];
	float pad1;
	float xy_scale[2];
	int32_t resolution_xy[2];
	float inv_resolution_xy[2];
	int32_t num_lights;
	int32_t num_lights_32;
	int32_t num_decals;
	int32_t num_decals_32;
	int32_t decals_texture_offset;
	int32_t z_max_index;
	float z_scale;
	float pad2[3];
} gr_cluster_params;

#define GR_MAX_LIGHTS_BINDLESS 4096
#define GR_CULL_SETUP_BYTES_PER_LIGHT 512        /* CullSetup {vec4 data[32]}, clusterer.cpp:1601 */
#define GR_TRANSFORMED_SPOT_BYTES_PER_LIGHT 96   /* TransformedSpot {vec4 clip[5]; vec4 z}, clusterer.cpp:1606 */

/* ClustererBindlessTransforms byte offsets (clusterer_data.h:46-53), total 852480 B (clusterer.cpp:1596). */
#define GR_TRANSFORMS_OFFSET_LIGHTS 0u
#define GR_TRANSFORMS_OFFSET_SHADOW 196608u
#define GR_TRANSFORMS_OFFSET_MODEL 458752u
#define GR_TRANSFORMS_OFFSET_TYPE_MASK 655360u
#define GR_TRANSFORMS_OFFSET_DECALS 655872u
#define GR_TRANSFORMS_SIZE 852480u

/* clusterer_bindless_spot_transform.comp; push = clusterer.cpp:1477-1493 (100 B). */
typedef struct gr_push_spot_transform
{
	float vp[16];
	float camera_pos[3];
	uint32_t num_lights;
	float camera_front[3];
	float z_near;
	float z_far;
} gr_push_spot_transform;
int gr_cluster_spot_transform(gr_ctx *ctx, gr_stream stream, const void *transforms, void *transformed_spots,
                              const gr_push_spot_transform *push);

/* clusterer_bindless_setup.comp; push = clusterer.cpp:1502-1509 (68 B).  params is passed by value (UBO). */
typedef struct gr_push_cluster_setup
{
	float view[16];
	uint32_t num_lights;
} gr_push_cluster_setup;
int gr_cluster_setup(gr_ctx *ctx, gr_stream stream, const void *transforms, const void *transformed_spots,
                     void *cull_setup, const gr_cluster_params *params, const gr_push_cluster_setup *push);

/* clusterer_bindless_binning.comp, SUBGROUPS variant at wave64 (8x8 cell tile per wave; clusterer.cpp:1516-1561).
 * bitmask[(y*res_x + x) * num_lights_32 + chunk]. */
int gr_cluster_binning(gr_ctx *ctx, gr_stream stream, const void *transforms, const void *cull_setup, uint32_t *bitmask,
                       const gr_cluster_params *params);

/* clusterer_bindless_z_range{,_opt}.comp; push = clusterer.cpp:1291-1300.  light_ranges: uvec2[num_volumes] slice
 * intervals (compute_uint_range, clusterer.cpp:1265-1275); out: uvec2[num_ranges] = [first, last] light index. */
typedef struct gr_push_z_range
{
	uint32_t num_volumes;
	uint32_t num_volumes_128;
	uint32_t num_ranges;
} gr_push_z_range;
int gr_cluster_z_range(gr_ctx *ctx, gr_stream stream, const uint32_t *light_ranges, uint32_t *out,
                       const gr_push_z_range *push);

/* The front of the cluster build as ONE launch: the four cmd.update_buffer of clusterer.cpp:1178-1207,1302, spot_transform.comp +
 * setup.comp (clusterer.cpp:1463-1510) and z_range.comp (:1277-1346) all depend on the frame's CPU-packed light data only.
 * src_* are the packed arrays in pinned host memory (gr_alloc_host) -- the launch copies them into `transforms` / `light_ranges`
 * as it reads them -- or NULL when the buffers already hold them.  Results are bit-identical to the separate launches above;
 * only gr_cluster_binning remains to be launched after it. */
typedef struct gr_cluster_front_args
{
	void *transforms;               /* device: GR_TRANSFORMS_SIZE bytes */
	const void *src_lights;         /* pinned: num_lights gr_light_info, or NULL */
	const void *src_models;         /* pinned: num_lights gr_mat_affine, or NULL */
	const void *src_type_mask;      /* pinned: num_lights_32 words, or NULL */
	void *transformed_spots;        /* device, out */
	void *cull_setup;               /* device, out */
	const gr_cluster_params *params;
	const gr_push_spot_transform *spot_push;
	const gr_push_cluster_setup *setup_push;
	const void *src_ranges;         /* pinned: z_push->num_volumes uvec2 slice intervals, or NULL */
	uint32_t *light_ranges;         /* device copy of the intervals */
	uint32_t *range_out;            /* device, out: uvec2[num_ranges] */
	const gr_push_z_range *z_push;
} gr_cluster_front_args;
int gr_cluster_front(gr_ctx *ctx, gr_stream stream, const gr_cluster_front_args *args);

/* DeferredLightRenderer::render_light (renderer.cpp:1004-1156): directional quad (directional.frag) then clustered quad
 * (clustering.frag), both additively blended into the RGBA16F HDR target with depth test NOT_EQUAL against z = 0.
 * One fused kernel; the intermediate blend result is rounded to fp16 exactly where the two reference draws round it:
 * hdr = rne16(rne16(emissive + directional) + clustered); alpha and far-plane pixels are copied through. */
typedef struct gr_push_directional /* renderer.cpp:1073-1103, 88 B used */
{
	float inv_view_proj_col2[4];
	float color[3];
	float environment_intensity;
	float camera_pos[3];
	float environment_mipscale;
	float direction[3];
	float cascade_log_bias;
	float camera_front[3];
	float pad0;
	float inv_resolution[2];
	float pad1[2];
} gr_push_directional;

typedef struct gr_push_clustering /* renderer.cpp:1110-1121 */
{
	float inv_view_proj_col2[4];
	float camera_pos[3];
	float pad0;
	float inv_resolution[2];
	float pad1[2];
} gr_push_clustering;

#define GR_LIGHTING_DIRECTIONAL_BIT 1u        /* draw the directional quad */
#define GR_LIGHTING_CLUSTERED_BIT 2u          /* draw the clustered quad */
#define GR_LIGHTING_AMBIENT_FALLBACK_BIT 4u   /* VOLUMETRIC_DIFFUSE_FALLBACK, renderer.cpp:1049-1055 */
#define GR_LIGHTING_AMBIENT_OCCLUSION_BIT 8u  /* AMBIENT_OCCLUSION (renderer.cpp:1050-1051): `ambient_occlusion` scales the
                                                 fallback ambient term, directional.frag:52-64.  Needs the fallback bit. */
/* Scheduling hint, no effect on any value: register-heavy passes of other streams run beside this launch (the temporal resolve and the
 * SMAA passes of the frame before: 512-thread workgroups of 64 registers, i.e. 128 per SIMD).  The launch then leaves them a quarter of each SIMD's register
 * file -- four resident workgroups per CU instead of five.  Measured (profiles/r05_lighting_instruction_diet.txt): five are 3 % faster
 * for a lighting + bloom + tonemap frame, 5-10 % slower for the TAA High + SMAA Ultra frame. */
#define GR_LIGHTING_SHARE_REGISTERS_BIT 16u

typedef struct gr_lighting_args
{
	gr_image albedo; /* R8G8B8A8_SRGB, a = ambient */
	gr_image normal; /* A2B10G10R10_UNORM_PACK32 */
	gr_image pbr;    /* R8G8_UNORM (metallic, roughness) */
	gr_image depth;  /* D32_SFLOAT, reverse-Z */
	gr_image emissive; /* R16G16B16A16_SFLOAT blend destination contents before the draws (G-buffer emissive) */
	gr_image hdr;    /* R16G16B16A16_SFLOAT result.  May alias `emissive` (same ptr): that is the reference's
	                    add_color_output("HDR", info, "emissive") read-modify-write; distinct buffers give the same
	                    values and bytes without clobbering the G-buffer. */
	float inv_view_projection[16]; /* DirectionalLightUBO / clustering.vert UBO */
	gr_push_directional directional;
	gr_push_clustering clustering;
	gr_cluster_params cluster;     /* UBO, by value */
	const void *transforms;        /* cluster-transforms buffer */
	const uint32_t *bitmask;       /* cluster-bitmask */
	const uint32_t *range;         /* cluster-range, uvec2[res_z] */
	uint32_t flags;
	gr_rows rows;                  /* render area; {0, 0} = whole target */
	gr_image ambient_occlusion;    /* R8_UNORM, any size, sampled LinearClamp at the pixel centre (LightingParameters::
	                                  ambient_occlusion, renderer.cpp:611-612); read only with the AMBIENT_OCCLUSION bit */
	/* The fog quad render_light draws last when LightingParameters::fog.falloff > 0 (renderer.cpp:1179-1196, lights/fog.{vert,frag},
	 * fog.h): src = (fog_color, exp2(-|world_pos - camera_pos|^2 * falloff)) blended ONE_MINUS_SRC_ALPHA / SRC_ALPHA into the target
	 * (colour and alpha), under the lighting quads' depth test.  falloff <= 0 = no fog (a zeroed struct). */
	float fog_color[3];
	float fog_falloff;
} gr_lighting_args;
int gr_lighting(gr_ctx *ctx, gr_stream stream, const gr_lighting_args *args);

/* 24-bit transport form of an RGBA8 target whose alpha byte is 255 everywhere (a tonemapped frame): the row-band executor
 * all-gathers its finished bands in this form (3/4 of the bytes over xGMI) and restores the RGBA8 rows on arrival.  `packed` is a
 * width x height x 3 byte image (row y at y * width * 3); *_rows restrict the rows as everywhere else (16-byte accesses where pitch, base and
 * width allow, bytes otherwise).  No reference counterpart (Granite renders a frame on one device). */
int gr_pack_rgb8_rows(gr_ctx *ctx, gr_stream stream, const gr_image *image, const gr_rows *rows, void *packed);
int gr_unpack_rgb8_rows(gr_ctx *ctx, gr_stream stream, const void *packed, const gr_image *image, const gr_rows *rows);

/* builtin://shaders/blit.frag over a full-screen quad (FragColor = textureLod(uTex, vUV, 0)): the copy between targets of
 * different size / format that Granite's tools record (tools/aa_bench.cpp:97-105 into the HDR target with LinearClamp,
 * :138-147 into the swapchain with NearestClamp).  in / out: R16G16B16A16_SFLOAT, R8G8B8A8_UNORM or R8G8B8A8_SRGB (decoded on
 * the fetch / encoded by the store, as the image views would); linear != 0 = StockSampler::LinearClamp, else NearestClamp. */
int gr_blit(gr_ctx *ctx, gr_stream stream, const gr_image *in, const gr_image *out, int linear);

/* ---- anti-aliasing (renderer/post/{fxaa,smaa,temporal}.cpp) ------------------------------------------------------------ */

/* setup_fxaa_postprocess (fxaa.cpp:28-55) + fxaa.frag.  `in` is read through its UNORM alias (cmd.set_unorm_texture):
 * the stored bytes of the tonemapped R8G8B8A8_SRGB image.  out: R8G8B8A8_SRGB or _UNORM (same bytes either way: the
 * shader's decode_srgb and the attachment's encode cancel). */
typedef struct gr_push_fxaa
{
	float inv_resolution[2];
} gr_push_fxaa;
int gr_fxaa(gr_ctx *ctx, gr_stream stream, const gr_image *in, const gr_image *out, const gr_push_fxaa *push);
/* The anti-aliasing passes over a render area (row bands): *_rows = the same pass restricted to output
 * rows [first, first + count); what a pass reads above and below its band is the executor's business (StripPlan). */
int gr_fxaa_rows(gr_ctx *ctx, gr_stream stream, const gr_image *in, const gr_image *out, const gr_push_fxaa *push, const gr_rows *rows);

/* SMAA 1x (smaa.cpp:32-208 + SMAA.hlsl).  quality 0..3 = SMAA_PRESET_LOW..ULTRA (smaa_common.h).
 * rt_metrics = (1/w, 1/h, w, h) (smaa.cpp:129-133). */
typedef struct gr_push_smaa
{
	float rt_metrics[4];
} gr_push_smaa;
/* AreaTex 160x560 R8G8_UNORM and SearchTex 64x16 R8_UNORM payloads (assets/textures/smaa/{area,search}.gtx), host pointers. */
int gr_smaa_set_luts(gr_ctx *ctx, const void *area_rg8, const void *search_r8);
/* smaa-edge pass: SMAALumaEdgeDetectionPS.  edges: R8G8_UNORM; every pixel is written (0 where the shader discards). */
int gr_smaa_edge_detection(gr_ctx *ctx, gr_stream stream, const gr_image *color, const gr_image *edges, const gr_push_smaa *push, int quality);
/* smaa-weights pass: SMAABlendingWeightCalculationPS, SMAA_SUBPIXEL_MODE 0.  weights: R8G8B8A8_UNORM.  The reference's
 * D16 "smaa-mask" depth-EQUAL test is equivalent to "edge texel != 0", which is what the kernel tests. */
int gr_smaa_blend_weight(gr_ctx *ctx, gr_stream stream, const gr_image *edges, const gr_image *weights, const gr_push_smaa *push, int quality);
/* smaa-blend pass: SMAANeighborhoodBlendingPS. */
int gr_smaa_neighbor_blend(gr_ctx *ctx, gr_stream stream, const gr_image *color, const gr_image *weights, const gr_image *out,
                           const gr_push_smaa *push);
int gr_smaa_edge_detection_rows(gr_ctx *ctx, gr_stream stream, const gr_image *color, const gr_image *edges, const gr_push_smaa *push,
                                int quality, const gr_rows *rows);
int gr_smaa_blend_weight_rows(gr_ctx *ctx, gr_stream stream, const gr_image *edges, const gr_image *weights, const gr_push_smaa *push,
                              int quality, const gr_rows *rows);
int gr_smaa_neighbor_blend_rows(gr_ctx *ctx, gr_stream stream, const gr_image *color, const gr_image *weights, const gr_image *out,
                                const gr_push_smaa *push, const gr_rows *rows);
/* setup_taa_resolve (temporal.cpp:199-266) + taa_resolve.frag.  quality 0..2 = TAAQuality Low/Medium/High.
 * history NULL => REPROJECTION_HISTORY = 0 (first frame).  current/out_color/history: R16G16B16A16_SFLOAT,
 * depth D32_SFLOAT, mv R16G16_SFLOAT.  reproj = T(.5,.5,0) S(.5,.5,1) VP_prev inv(VP_cur) (temporal.cpp:239-243). */
typedef struct gr_push_taa
{
	float reproj[16];
	float rt_metrics[4];
} gr_push_taa;
int gr_taa_resolve(gr_ctx *ctx, gr_stream stream, const gr_image *current, const gr_image *depth, const gr_image *mv,
                   const gr_image *history, const gr_image *out_color, const gr_image *out_history, const gr_push_taa *push, int quality);
int gr_taa_resolve_rows(gr_ctx *ctx, gr_stream stream, const gr_image *current, const gr_image *depth, const gr_image *mv,
                        const gr_image *history, const gr_image *out_color, const gr_image *out_history, const gr_push_taa *push, int quality,
                        const gr_rows *rows);
/* Row bands whose history only holds the rows around the band (no reference analogue): history_rows = the rows of
 * `history` that carry last frame's values on this device.  A resolved pixel whose reprojection (motion vector, or depth + reproj)
 * fetches a history row outside them stores 1 to *reach_flag, which must be memory the device can write and the host can read
 * (gr_alloc_host): such a frame is not the single-device frame, and the caller has to say so.  Both NULL = gr_taa_resolve_rows. */
int gr_taa_resolve_band(gr_ctx *ctx, gr_stream stream, const gr_image *current, const gr_image *depth, const gr_image *mv,
                        const gr_image *history, const gr_image *out_color, const gr_image *out_history, const gr_push_taa *push, int quality,
                        const gr_rows *rows, const gr_rows *history_rows, uint32_t *reach_flag);

/* ---- depth hierarchy ---------------------------------------------------------------------------------------------------
 * HiZPassState::build_render_pass (renderer/post/spd.cpp:141-194) + assets/shaders/post/hiz.comp: turns the depth
 * attachment into a full max-reduction mip chain of linearised depth.  Workgroups reduce 64 x 64 tiles down to one texel
 * (mips 0..6); one more workgroup reduces what is left, folding in the odd row / column of non-power-of-two levels.  The
 * shader hands over inside one dispatch through an atomic counter (hiz.comp:355-365); with eight XCD-private L2s that
 * hand-over is cheaper as a second, one-workgroup launch on the same stream, so the counter is accepted for interface
 * parity but not touched.
 *
 * Mip chains live in one allocation, level after level, each level tightly packed; level l of a w x h chain is
 * max(w >> l, 1) x max(h >> l, 1) texels (gr_mip_chain_offset / gr_mip_chain_size, in bytes). */
size_t gr_mip_chain_offset(uint32_t width, uint32_t height, uint32_t bytes_per_texel, uint32_t level);
size_t gr_mip_chain_size(uint32_t width, uint32_t height, uint32_t bytes_per_texel, uint32_t levels);

typedef struct gr_hiz_args
{
	gr_image depth;            /* D32_SFLOAT / R32_SFLOAT; texels past its edge repeat the edge (NearestClamp) */
	void *chain;               /* R32_SFLOAT mip chain */
	uint32_t chain_width;      /* level 0 of the chain: the input size rounded up to multiples of 64 (spd.cpp:214-215), */
	uint32_t chain_height;     /* halved when output_downsample */
	uint32_t chain_levels;     /* spd.cpp:212: max(1, floor_log2(max(w, h)) - output_downsample) */
	uint32_t output_downsample; /* 1: the transformed full-resolution level is not stored; chain level 0 is mip 1 */
	float z_transform[4];      /* column-major mat2, spd.cpp:164-165: depth -> (num, den), stored value min(num / den, 1e30) */
	uint32_t *counter;         /* the pass's "-counter" buffer (one uint32); never written, stays zero */
} gr_hiz_args;
int gr_hiz(gr_ctx *ctx, gr_stream stream, const gr_hiz_args *args);

/* ---- screen-space reflections ------------------------------------------------------------------------------------------------
 * SSRState::build_render_pass (renderer/post/ssr.cpp:84-173) with ffx-sssr/{classify,build_indirect,trace_primary}.comp, and
 * the apply pass (ssr.cpp:286-322, apply.frag).  gr_ssr_trace = classify (clears `output` / `ray_confidence`, lists one ray per
 * glossy (roughness < 0.2), reflective (hierarchy level 0 < 1) pixel -- one per 2 x 2 quad phase for non-mirror surfaces, with
 * copy flags), build_indirect (the counter buffer: indirect.xyzw, atomic_count = 0, copied_count) and trace_primary (GGX VNDF
 * reflection direction from the blue-noise dither layer `frame`, hierarchical traversal of the depth chain from mip 1,
 * hit validation, radiance from `light`) in one call.  The ray list is in tile order (8 x 8 tiles row-major, Z-order inside a
 * tile) -- one of the orders the shader's atomic append can produce -- so the wave-level early exit of the traversal is
 * reproducible.  texelFetch outside a level reads 0.  scratch: gr_ssr_scratch_bytes(width, height) bytes. */
typedef struct gr_ssr_args
{
	const void *depth_chain;      /* R32_SFLOAT mip chain as gr_hiz writes it ("<depth>-hier") */
	uint32_t chain_width, chain_height, chain_levels;
	gr_image pbr;                 /* R8G8_UNORM */
	gr_image normal;              /* A2B10G10R10_UNORM_PACK32 */
	gr_image light;               /* R16G16B16A16_SFLOAT: the lit target rays fetch radiance from */
	const void *dither_lut;       /* R8G8_UNORM 128 x 128 x 64 layers (ssr.cpp:178-205) */
	uint32_t frame;               /* dither layer, 0..63 (ssr.cpp:161) */
	float view_projection[16];    /* column-major */
	float inv_view_projection[16];
	float camera_position[3];
	gr_image output;              /* R16G16B16A16_SFLOAT "<output>-sssr" */
	gr_image ray_length;          /* R16_SFLOAT */
	gr_image ray_confidence;      /* R8_UNORM */
	uint32_t *ray_list;           /* width * height dwords */
	uint32_t *ray_counter;        /* >= 6 dwords */
	void *scratch;
} gr_ssr_args;
size_t gr_ssr_scratch_bytes(uint32_t width, uint32_t height);
int gr_ssr_trace(gr_ctx *ctx, gr_stream stream, const gr_ssr_args *args);

typedef struct gr_ssr_apply_args
{
	gr_image hdr;       /* R16G16B16A16_SFLOAT, read-modify-write: blend ONE / ONE, fp16 rounding of the sum */
	gr_image reflected; /* "<output>-sssr", NearestClamp at the pixel centre */
	gr_image albedo;    /* R8G8B8A8_SRGB */
	gr_image normal;
	gr_image pbr;
	gr_image depth;     /* D32_SFLOAT: pixels at exactly 1.0 are skipped (depth test NOT_EQUAL against the quad at z = 1) */
	gr_image brdf_lut;  /* R16G16_SFLOAT split-sum table (builtin://textures/ibl_brdf_lut.gtx), LinearClamp at (NoV, roughness) */
	float inv_view_projection[16];
	float camera_position[3];
} gr_ssr_apply_args;
int gr_ssr_apply(gr_ctx *ctx, gr_stream stream, const gr_ssr_apply_args *args);

/* ---- single-pass downsampler ----------------------------------------------------------------------------------------------
 * emit_single_pass_downsample (renderer/post/spd.cpp:56-102) + assets/shaders/post/ffx-spd/spd.comp (AMD FidelityFX SPD):
 * fills an RGBA16F mip chain from an RGBA16F image in one call -- SPDInfo's input, output_mips[0..num_mips), num_components,
 * filter_mod and mode.  Level 0 of the chain is width x height (the size of output_mips[0]: half the input when the input
 * is the level above it, as in the reference's use), written from one LinearClamp tap per texel at the centre of its 2 x 2
 * source footprint (NearestClamp of the co-located texel in depth mode); levels 1.. are 2 x 2 averages (minima of .x in
 * depth mode) with SPD's rounding points: fp32 inside a 64 x 64 source tile down to level 5, level 6 from level 5 as
 * stored.  Every store is multiplied by filter_mods[level] (mips x vec4, host memory, NULL = none) and cut to `components`
 * channels (the rest written as 0).  Texels of a level that lie outside max(size >> level, 1) do not exist; the chain is
 * laid out as gr_mip_chain_offset(width, height, 8, level) says.  The shader's atomic counter buffer has no counterpart:
 * the last-workgroup stage is a second launch on the same stream. */
#define GR_SPD_REDUCTION_COLOR 0u
#define GR_SPD_REDUCTION_DEPTH 1u
typedef struct gr_spd_args
{
	gr_image input;            /* R16G16B16A16_SFLOAT */
	void *chain;               /* R16G16B16A16_SFLOAT mip chain, `mips` levels */
	uint32_t width, height;    /* level 0 of the chain = push.base_image_resolution (spd.cpp:85-86) */
	uint32_t mips;             /* 1..12 */
	uint32_t components;       /* 1..4: COMPONENTS */
	uint32_t reduction_mode;   /* GR_SPD_REDUCTION_* */
	const float *filter_mods;  /* FILTER_MOD: mips x 4 floats in host memory, or NULL */
} gr_spd_args;
int gr_spd_downsample(gr_ctx *ctx, gr_stream stream, const gr_spd_args *args);

/* ---- spatial upscaling after the post chain ---------------------------------------------------------------------------
 * setup_after_post_chain_upscaling (renderer/post/aa.cpp:75-174).  Both images R8G8B8A8_{UNORM,SRGB}.
 * gr_fsr_upscale: the "-scale" pass, upscale.{vert,frag} + FsrEasu{F,H} with the constants of FsrEasuCon (viewport = input
 * size): reads the stored (gamma-space) bytes of `in`, writes out->width x out->height.  fp16 != 0 selects the FP16 shader
 * variant (the reference's choice on hardware with fp16 arithmetic, aa.cpp:118-119).  The value written is the gamma-space
 * result for either output format (TARGET_SRGB decodes and the *_SRGB store re-encodes).
 * gr_fsr_sharpen: the "-sharpen" pass, sharpen.{vert,frag} + FsrRcasF; sharpness = exp2(-stops), FsrRcasCon (aa.cpp:64-74,
 * 0.5 stops at :157).  With an *_SRGB output the input is read through its sRGB view (linear) and the store encodes
 * (aa.cpp:147-151); with a UNORM output bytes go straight through. */
int gr_fsr_upscale(gr_ctx *ctx, gr_stream stream, const gr_image *in, const gr_image *out, int fp16);
int gr_fsr_sharpen(gr_ctx *ctx, gr_stream stream, const gr_image *in, const gr_image *out, float sharpness);

/* ---- HDR10 output ---------------------------------------------------------------------------------------------------------
 * setup_hdr10_pq_encoding (renderer/post/hdr.cpp:595-658) + pq10_encode.frag: out = PQ(soft-clip(primary_conversion *
 * (hdr * hdr_pre_exposure * ui.a + ui.rgb * ui_pre_exposure) / max) * max) in the display's primaries.  hdr R16G16B16A16_SFLOAT
 * (texel fetch), ui R8G8B8A8 read through its sRGB view, out A2B10G10R10_UNORM_PACK32 (alpha 1); all the same size.  The
 * push block is byte-identical to the shader's UBO (hdr.cpp:626-633). */
typedef struct gr_push_pq10
{
	float primary_conversion[16]; /* mat4(compute_rec709_to_st2020(metadata)), column major; the 3 x 3 part is used */
	float hdr_pre_exposure;
	float ui_pre_exposure;
	float max_light_level;        /* VkHdrMetadataEXT::maxContentLightLevel */
	float inv_max_light_level;
} gr_push_pq10;
int gr_pq10_encode(gr_ctx *ctx, gr_stream stream, const gr_image *hdr, const gr_image *ui, const gr_image *out, const gr_push_pq10 *push);
/* Fill with a 32-bit pattern (count dwords): attachment clears to a colour. */
int gr_fill_u32(gr_ctx *ctx, gr_stream stream, void *dst, uint32_t value, size_t count);
/* Executor self-test operation (no counterpart in the reference): out[i] = hash(i, salt, one dword of each of up to four
 * inputs).  Used by tests/cpp/graph_cases.cpp --execute to run random frame graphs on the executor's streams and compare the
 * swapchain image with a serial run. */
/* fp32 (r, g, b) triples -> B10G11R11_UFLOAT_PACK32 words: the attachment store conversion the lighting and TAA kernels apply
 * when their target has that format (closest finite packed value, ties to even; negative -> 0; +inf -> +inf; NaN -> NaN). */
int gr_pack_b10g11r11(gr_ctx *ctx, gr_stream stream, const float *rgb, uint32_t *out, uint32_t texels);
int gr_debug_mix(gr_ctx *ctx, gr_stream stream, void *out, size_t out_dwords, const void *const *inputs, const size_t *input_dwords,
                 uint32_t input_count, uint32_t salt);
/* VkPhysicalDeviceProperties::deviceName / driverVersion as the headless runner reports them in its --stat file
 * (application_headless.cpp:634-635): HIP device name and hipDriverGetVersion. */
int gr_get_device_info(gr_ctx *ctx, char *name, size_t name_capacity, uint32_t *driver_version);

/* ---- layout contracts, checked where the structs are declared: the push-constant blocks and the
 * cluster UBO are byte-identical to what the reference's host code pushes and its shaders declare. ------------------------------------ */
#include <stddef.h>
#ifdef __cplusplus
#define GR_STATIC_ASSERT(cond, what) static_assert(cond, what)
#else
#define GR_STATIC_ASSERT(cond, what) _Static_assert(cond, what)
#endif
#define GR_ASSERT_SIZE(type, bytes) GR_STATIC_ASSERT(sizeof(type) == (bytes), #type ": size differs from the reference's block")
#define GR_ASSERT_OFFSET(type, member, bytes) GR_STATIC_ASSERT(offsetof(type, member) == (bytes), #type "." #member ": offset differs from the reference's block")
GR_ASSERT_SIZE(gr_luminance_data, 12);        /* luminance.comp:4-9 */
GR_ASSERT_SIZE(gr_push_luminance, 20);        /* hdr.cpp:85-96 */
GR_ASSERT_OFFSET(gr_push_luminance, lerp, 8);
GR_ASSERT_OFFSET(gr_push_luminance, min_loglum, 12);
GR_ASSERT_OFFSET(gr_push_luminance, max_loglum, 16);
GR_ASSERT_SIZE(gr_push_bloom_threshold, 16);  /* hdr.cpp:133-142 */
GR_ASSERT_OFFSET(gr_push_bloom_threshold, inv_output_size, 8);
GR_ASSERT_SIZE(gr_push_bloom_downsample, 28); /* hdr.cpp:169-185 */
GR_ASSERT_OFFSET(gr_push_bloom_downsample, inv_output_size, 8);
GR_ASSERT_OFFSET(gr_push_bloom_downsample, inv_input_size, 16);
GR_ASSERT_OFFSET(gr_push_bloom_downsample, lerp, 24);
GR_ASSERT_SIZE(gr_push_bloom_upsample, 24);   /* hdr.cpp:202-214 */
GR_ASSERT_OFFSET(gr_push_bloom_upsample, inv_input_size, 16);
GR_ASSERT_SIZE(gr_push_tonemap, 4);           /* hdr.cpp:297-302 */
GR_ASSERT_SIZE(gr_push_fxaa, 8);              /* fxaa.cpp:45-47 */
GR_ASSERT_SIZE(gr_push_smaa, 16);             /* smaa.cpp:129-133 */
GR_ASSERT_SIZE(gr_push_taa, 80);              /* temporal.cpp:232-250 */
GR_ASSERT_OFFSET(gr_push_taa, rt_metrics, 64);
GR_ASSERT_SIZE(gr_push_clustering, 48);       /* renderer.cpp:1110-1121: 40 B used, 48 in C++ */
GR_ASSERT_OFFSET(gr_push_clustering, camera_pos, 16);
GR_ASSERT_OFFSET(gr_push_clustering, inv_resolution, 32);
GR_ASSERT_SIZE(gr_push_directional, 96);      /* renderer.cpp:1073-1103: 88 B used */
GR_ASSERT_OFFSET(gr_push_directional, color, 16);
GR_ASSERT_OFFSET(gr_push_directional, environment_intensity, 28);
GR_ASSERT_OFFSET(gr_push_directional, camera_pos, 32);
GR_ASSERT_OFFSET(gr_push_directional, environment_mipscale, 44);
GR_ASSERT_OFFSET(gr_push_directional, direction, 48);
GR_ASSERT_OFFSET(gr_push_directional, cascade_log_bias, 60);
GR_ASSERT_OFFSET(gr_push_directional, camera_front, 64);
GR_ASSERT_OFFSET(gr_push_directional, inv_resolution, 80);
GR_ASSERT_SIZE(gr_push_spot_transform, 100);  /* clusterer.cpp:1477-1493 */
GR_ASSERT_OFFSET(gr_push_spot_transform, camera_pos, 64);
GR_ASSERT_OFFSET(gr_push_spot_transform, num_lights, 76);
GR_ASSERT_OFFSET(gr_push_spot_transform, camera_front, 80);
GR_ASSERT_OFFSET(gr_push_spot_transform, z_near, 92);
GR_ASSERT_OFFSET(gr_push_spot_transform, z_far, 96);
GR_ASSERT_SIZE(gr_push_cluster_setup, 68);    /* clusterer.cpp:1502-1509 */
GR_ASSERT_OFFSET(gr_push_cluster_setup, num_lights, 64);
GR_ASSERT_SIZE(gr_push_z_range, 12);          /* clusterer.cpp:1291-1300 */
GR_ASSERT_SIZE(gr_light_info, 48);            /* light_info.hpp:35-44 */
GR_ASSERT_OFFSET(gr_light_info, spot_scale_bias, 12);
GR_ASSERT_OFFSET(gr_light_info, position, 16);
GR_ASSERT_OFFSET(gr_light_info, offset_radius, 28);
GR_ASSERT_OFFSET(gr_light_info, direction, 32);
GR_ASSERT_OFFSET(gr_light_info, inv_radius, 44);
GR_ASSERT_SIZE(gr_mat_affine, 48);            /* muglm.hpp:899-958 */
GR_ASSERT_SIZE(gr_cluster_params, 176);       /* clusterer_data.h:20-39, std140 */
GR_ASSERT_OFFSET(gr_cluster_params, clip_scale, 64);
GR_ASSERT_OFFSET(gr_cluster_params, camera_base, 80);
GR_ASSERT_OFFSET(gr_cluster_params, camera_front, 96);
GR_ASSERT_OFFSET(gr_cluster_params, xy_scale, 112);
GR_ASSERT_OFFSET(gr_cluster_params, resolution_xy, 120);
GR_ASSERT_OFFSET(gr_cluster_params, inv_resolution_xy, 128);
GR_ASSERT_OFFSET(gr_cluster_params, num_lights, 136);
GR_ASSERT_OFFSET(gr_cluster_params, num_lights_32, 140);
GR_ASSERT_OFFSET(gr_cluster_params, num_decals, 144);
GR_ASSERT_OFFSET(gr_cluster_params, decals_texture_offset, 152);
GR_ASSERT_OFFSET(gr_cluster_params, z_max_index, 156);
GR_ASSERT_OFFSET(gr_cluster_params, z_scale, 160);
GR_ASSERT_SIZE(gr_push_pq10, 80);             /* hdr.cpp:626-633 */
GR_STATIC_ASSERT(GR_TRANSFORMS_OFFSET_SHADOW == GR_MAX_LIGHTS_BINDLESS * 48u, "ClustererBindlessTransforms: lights[4096] of 48 B");
GR_STATIC_ASSERT(GR_TRANSFORMS_OFFSET_MODEL + GR_MAX_LIGHTS_BINDLESS * 48u == GR_TRANSFORMS_OFFSET_TYPE_MASK, "ClustererBindlessTransforms: model[4096] of 48 B");
GR_STATIC_ASSERT(GR_TRANSFORMS_OFFSET_TYPE_MASK + GR_MAX_LIGHTS_BINDLESS / 8u == GR_TRANSFORMS_OFFSET_DECALS, "ClustererBindlessTransforms: type_mask[128]");
#undef GR_ASSERT_SIZE
#undef GR_ASSERT_OFFSET

#ifdef __cplusplus
}
#endif
#endif
