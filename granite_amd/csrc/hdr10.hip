// HDR10 output encode for gfx950: PQEncoder::build_render_pass (renderer/post/hdr.cpp:619-641) +
// assets/shaders/post/pq10_encode.frag.  16 B of traffic per pixel (8 B HDR + 4 B UI in, 4 B out) against six pow() per
// pixel, evaluated as v_exp_f32(m * v_log_f32(x)); the 78.84 exponent amplifies the relative error of the inner quotient to
// ~1e-5, two orders below one step of the 10-bit output.
#include "ctx.hpp"
#include "device_common.hpp"

namespace
{
struct Pq10Args
{
	DevImage hdr, ui;
	uint8_t *out;
	uint32_t out_pitch;
	float m[9]; // column-major mat3
	float hdr_pre_exposure, ui_pre_exposure, max_light_level, inv_max_light_level;
	const float *srgb_lut;
};

__device__ __forceinline__ float pow_fast(float x, float e) { return __builtin_amdgcn_exp2f(e * __builtin_amdgcn_logf(x)); }

__device__ __forceinline__ float encode_pq(float nits)
{
	const float y = nits * (1.0f / 10000.0f);
	const float c1 = 0.8359375f, c2 = 18.8515625f, c3 = 18.6875f, m1 = 0.1593017578125f, m2 = 78.84375f;
	const float p = y > 0.0f ? pow_fast(y, m1) : 0.0f;
	const float q = fmaf(c2, p, c1) * __builtin_amdgcn_rcpf(fmaf(c3, p, 1.0f));
	return pow_fast(q, m2);
}

__device__ __forceinline__ uint32_t unorm10(float v)
{
	v = fminf(fmaxf(v, 0.0f), 1.0f); // NaN -> 0
	return uint32_t(v * 1023.0f + 0.5f);
}

__device__ __forceinline__ float soft_clip(float c)
{
	const float k = c * 4.0f;
	return c > 0.75f ? k * __builtin_amdgcn_rcpf(1.0f + k) : c;
}

__global__ __launch_bounds__(256) void k_pq10_encode(Pq10Args a)
{
	const int x = blockIdx.x * 64 + (threadIdx.x & 63);
	const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
	if (x >= a.hdr.w || y >= a.hdr.h)
		return;
	const f16x4 h = *reinterpret_cast<const f16x4 *>(a.hdr.ptr + size_t(y) * a.hdr.pitch + size_t(x) * 8u);
	const uint32_t u = *reinterpret_cast<const uint32_t *>(a.ui.ptr + size_t(y) * a.ui.pitch + size_t(x) * 4u);
	const float hdr_scale = a.hdr_pre_exposure * unorm8_to_float(u >> 24);
	const float r = fmaf(float(h.x), hdr_scale, a.srgb_lut[u & 255u] * a.ui_pre_exposure);
	const float g = fmaf(float(h.y), hdr_scale, a.srgb_lut[(u >> 8) & 255u] * a.ui_pre_exposure);
	const float b = fmaf(float(h.z), hdr_scale, a.srgb_lut[(u >> 16) & 255u] * a.ui_pre_exposure);
	float c[3];
#pragma unroll
	for (int i = 0; i < 3; i++)
		c[i] = soft_clip(fmaf(a.m[6 + i], b, fmaf(a.m[3 + i], g, a.m[i] * r)) * a.inv_max_light_level) * a.max_light_level;
	const uint32_t packed = unorm10(encode_pq(c[0])) | (unorm10(encode_pq(c[1])) << 10) | (unorm10(encode_pq(c[2])) << 20) | (3u << 30);
	*reinterpret_cast<uint32_t *>(a.out + size_t(y) * a.out_pitch + size_t(x) * 4u) = packed;
}
} // namespace

extern "C" int gr_pq10_encode(gr_ctx *ctx, gr_stream stream, const gr_image *hdr, const gr_image *ui, const gr_image *out, const gr_push_pq10 *push)
{
	if (!ctx)
		return GR_ERR_INVALID_ARGUMENT;
	GR_CHECK_ARG(ctx, hdr && ui && out && push && hdr->ptr && ui->ptr && out->ptr && hdr->width && hdr->height);
	GR_CHECK_ARG(ctx, hdr->format == GR_FORMAT_R16G16B16A16_SFLOAT && hdr->pitch_bytes >= hdr->width * 8u);
	GR_CHECK_ARG(ctx, (ui->format == GR_FORMAT_R8G8B8A8_SRGB || ui->format == GR_FORMAT_R8G8B8A8_UNORM) && ui->width == hdr->width &&
	                      ui->height == hdr->height && ui->pitch_bytes >= ui->width * 4u);
	GR_CHECK_ARG(ctx, out->format == GR_FORMAT_A2B10G10R10_UNORM_PACK32 && out->width == hdr->width && out->height == hdr->height &&
	                      out->pitch_bytes >= out->width * 4u);
	GR_CHECK_ARG(ctx, push->max_light_level > 0.0f);
	Pq10Args a;
	a.hdr = DevImage{static_cast<const uint8_t *>(hdr->ptr), int(hdr->width), int(hdr->height), hdr->pitch_bytes};
	a.ui = DevImage{static_cast<const uint8_t *>(ui->ptr), int(ui->width), int(ui->height), ui->pitch_bytes};
	a.out = static_cast<uint8_t *>(out->ptr);
	a.out_pitch = out->pitch_bytes;
	for (int col = 0; col < 3; col++) // mat3(config.primary_conversion): the upper-left 3 x 3 of the column-major mat4
		for (int row = 0; row < 3; row++)
			a.m[3 * col + row] = push->primary_conversion[4 * col + row];
	a.hdr_pre_exposure = push->hdr_pre_exposure;
	a.ui_pre_exposure = push->ui_pre_exposure;
	a.max_light_level = push->max_light_level;
	a.inv_max_light_level = push->inv_max_light_level;
	a.srgb_lut = ctx->srgb_decode_lut;
	gr_scoped_timing timing{ctx, gr_to_stream(stream), "pq10_encode"};
	hipLaunchKernelGGL(k_pq10_encode, dim3(gr_div_up(hdr->width, 64), gr_div_up(hdr->height, 4)), dim3(256), 0, gr_to_stream(stream), a);
	GR_CHECK_LAUNCH(ctx);
	return GR_OK;
}
