// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// C entry points over the REFERENCE's own math library (math/muglm), compiled from the sources where they lie under
// /root/reference by oracle/ref_build/Makefile into oracle/_ref/libref_muglm.so.  Nothing from the reference is copied
// into this repository: this file only includes its headers at build time.  The library validates the restatements of
// the host-side math on the path (half packing of the light records, camera inverses, the TAA reprojection matrix) against
// the real reference code; it never ships and is absent on the GPU box unless built here first.
#include "muglm/muglm_impl.hpp"
#include "muglm/matrix_helper.hpp"
#include "transforms.hpp"
#include <cmath>
#include <cstdint>
#include <cstring>

// The FidelityFX headers the reference vendors, in their CPU mode: only the constant set-up functions exist there
// (FsrEasuCon / FsrRcasCon); the filters themselves are GPU-only code.
#define A_CPU 1
#include "ffx-a/ffx_a.h"
#include "ffx-fsr/ffx_fsr1.h"

using namespace muglm;

static mat4 load(const float *m)
{
	mat4 r;
	memcpy(&r, m, sizeof(float) * 16);
	return r;
}

static void store(float *out, const mat4 &m)
{
	memcpy(out, &m, sizeof(float) * 16);
}

extern "C" {

// muglm::floatToHalf / halfToFloat (math/muglm/muglm_impl.hpp:794-907): the CPU-side packing used by
// LightClusterer::refresh_bindless_prepare for spot scale / bias and offset / radius.
uint16_t ref_float_to_half(float v) { return floatToHalf(v); }
float ref_half_to_float(uint16_t v) { return halfToFloat(v); }
void ref_float_to_half_array(const float *in, uint16_t *out, uint64_t count)
{
	for (uint64_t i = 0; i < count; i++)
		out[i] = floatToHalf(in[i]);
}

// Column-major 4x4, as stored by muglm::mat4.
void ref_mat4_inverse(const float *m, float *out) { store(out, inverse(load(m))); }
void ref_mat4_mul(const float *a, const float *b, float *out) { store(out, load(a) * load(b)); }
void ref_perspective(float fovy, float aspect, float z_near, float z_far, float *out) { store(out, perspective(fovy, aspect, z_near, z_far)); }
void ref_translate(const float *v3, float *out) { store(out, translate(vec3(v3[0], v3[1], v3[2]))); }
void ref_scale(const float *v3, float *out) { store(out, scale(vec3(v3[0], v3[1], v3[2]))); }

// RenderContext::set_camera's derived quantities (renderer/render_context.cpp:53-86) computed with the reference's
// operators: out = inv_projection, inv_view, view_projection, inv_view_projection (16 floats each), then camera
// position (3), front (3), z_near, z_far.
void ref_camera_parameters(const float *projection, const float *view, float *out72)
{
	const mat4 p = load(projection), v = load(view);
	const mat4 vp = p * v;
	const mat4 inv_p = inverse(p), inv_v = inverse(v), inv_vp = inverse(vp);
	store(out72, inv_p);
	store(out72 + 16, inv_v);
	store(out72 + 32, vp);
	store(out72 + 48, inv_vp);
	const vec3 pos = inv_v[3].xyz();
	const vec3 front = -inv_v[2].xyz();
	out72[64] = pos.x, out72[65] = pos.y, out72[66] = pos.z;
	out72[67] = front.x, out72[68] = front.y, out72[69] = front.z;
	mat2 inv_zw(inv_p[2].zw(), inv_p[3].zw());
	auto project = [](const vec2 &zw) { return -zw.x / zw.y; };
	const bool infinite_z = inv_vp[3][3] == 0.0f;
	out72[70] = project(inv_zw * vec2(1.0f, 1.0f));
	out72[71] = project(inv_zw * vec2(infinite_z ? 1e-10f : 0.0f, 1.0f));
}

// setup_taa_resolve's reprojection matrix (renderer/post/temporal.cpp:239-243):
// translate(0.5, 0.5, 0) * scale(0.5, 0.5, 1) * view_proj_prev * inv_view_proj_current.
void ref_taa_reprojection(const float *prev_view_proj, const float *inv_view_proj, float *out)
{
	store(out, translate(vec3(0.5f, 0.5f, 0.0f)) * scale(vec3(0.5f, 0.5f, 1.0f)) * load(prev_view_proj) * load(inv_view_proj));
}

// FsrEasuCon as setup_after_post_chain_upscaling calls it (renderer/post/aa.cpp:106-108: viewport = input size) and
// FsrRcasCon (aa.cpp:157): the 16 + 4 constant words.
void ref_fsr_easu_constants(float iw, float ih, float ow, float oh, uint32_t *out16)
{
	FsrEasuCon(out16, out16 + 4, out16 + 8, out16 + 12, iw, ih, iw, ih, ow, oh);
}
void ref_fsr_rcas_constants(float stops, uint32_t *out4) { FsrRcasCon(out4, stops); }

// compute_rec709_to_st2020 (renderer/post/hdr.cpp:580-593) -- a static function there, so its three statements are repeated
// here over the reference's own compute_xyz_matrix (math/transforms.cpp:353-370), inverse and mat3 product.
// primaries8 = display red, green, blue, white point (CIE xy); out9 column major.
void ref_rec709_to_display(const float *primaries8, float *out9)
{
	using namespace Granite;
	const Primaries rec709 = {vec2(0.640f, 0.330f), vec2(0.3f, 0.6f), vec2(0.150f, 0.060f), vec2(0.3127f, 0.3290f)};
	const Primaries display = {vec2(primaries8[0], primaries8[1]), vec2(primaries8[2], primaries8[3]), vec2(primaries8[4], primaries8[5]),
	                           vec2(primaries8[6], primaries8[7])};
	const mat3 srgb_to_xyz = compute_xyz_matrix(rec709);
	const mat3 xyz_to_display = inverse(compute_xyz_matrix(display));
	const mat3 m = xyz_to_display * srgb_to_xyz;
	memcpy(out9, &m, sizeof(float) * 9);
}

} // extern "C"
