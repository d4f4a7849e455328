// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// Runs the REFERENCE's own spatial-upscaling shaders on the CPU: post/ffx-fsr/upscale.frag (FP16 = 0: FsrEasuF) and
// post/ffx-fsr/sharpen.frag (FsrRcasF) together with the FidelityFX headers the reference vendors (ffx-a/ffx_a.h,
// ffx-fsr/ffx_fsr1.h in their GLSL GPU spelling), re-spelled into gen/ at build time.  Bindings and constants as
// setup_after_post_chain_upscaling makes them (renderer/post/aa.cpp:75-174): NearestClamp, the UNORM alias of the input for
// EASU, FsrEasuCon(viewport = input size), FsrRcasCon(0.5 stops) -- both constant functions are the header's own.
// The FP16 = 1 variant (the reference's choice on fp16-capable hardware) runs FsrEasuH from the same headers on emulated
// float16 types (glsl_cpu.hpp: every operation rounds to half); its three gather callbacks are restated in
// fsr_easu_h_glue.glsl because upscale.frag relies on an implicit vec3 <- f16vec3 out-parameter conversion.
#include "glsl_cpu.hpp"

namespace glsl
{
#define FP16 0
#define TARGET_SRGB 0
namespace easu
{
#include "gen/fsr_upscale.inc"
}
#undef FSR_EASU_F // a macro of upscale.frag; sharpen.frag only asks for FSR_RCAS_F
namespace rcas
{
#include "gen/fsr_sharpen.inc"
}
#undef TARGET_SRGB
#undef FP16
#undef FSR_RCAS_F
namespace easu_h
{
#include "gen/fsr_easu_h.inc"
}
} // namespace glsl

using namespace glsl;

extern "C" {

void ref_fsr_easu(const uint8_t *in, int iw, int ih, uint8_t *out, int ow, int oh)
{
	namespace s = glsl::easu;
	s::uTex.data = in;
	s::uTex.w = iw;
	s::uTex.h = ih;
	s::uTex.format = Format::RGBA8_UNORM;
	s::uTex.filter = Filter::Nearest;
	s::FsrEasuCon(s::param0, s::param1, s::param2, s::param3, float(iw), float(ih), float(iw), float(ih), float(ow), float(oh));
	Image target;
	target.data = out;
	target.w = ow;
	target.h = oh;
	target.format = Format::RGBA8_UNORM;
	for (int y = 0; y < oh; y++)
		for (int x = 0; x < ow; x++)
		{
			s::vUV = vec2(float(x) + 0.5f, float(y) + 0.5f); // upscale.vert: (0.5 * Position + 0.5) * out_resolution
			s::main();
			imageStore(target, ivec2(x, y), s::FragColor);
		}
}

void ref_fsr_easu_fp16(const uint8_t *in, int iw, int ih, uint8_t *out, int ow, int oh)
{
	namespace s = glsl::easu_h;
	s::uTex.data = in;
	s::uTex.w = iw;
	s::uTex.h = ih;
	s::uTex.format = Format::RGBA8_UNORM;
	s::uTex.filter = Filter::Nearest;
	uvec4 con0, con1, con2, con3;
	s::FsrEasuCon(con0, con1, con2, con3, float(iw), float(ih), float(iw), float(ih), float(ow), float(oh));
	Image target;
	target.data = out;
	target.w = ow;
	target.h = oh;
	target.format = Format::RGBA8_UNORM;
	for (int y = 0; y < oh; y++)
		for (int x = 0; x < ow; x++)
		{
			f16vec3 pix;
			s::FsrEasuH(pix, uvec2(uint(x), uint(y)), con0, con1, con2, con3);
			imageStore(target, ivec2(x, y), vec4(float(pix.x), float(pix.y), float(pix.z), 1.0f));
		}
}

void ref_fsr_rcas(const uint8_t *in, int w, int h, uint8_t *out, float stops, int srgb)
{
	namespace s = glsl::rcas;
	s::uTex.data = in;
	s::uTex.w = w;
	s::uTex.h = h;
	s::uTex.format = srgb ? Format::RGBA8_SRGB : Format::RGBA8_UNORM; // cmd.set_srgb_texture / set_unorm_texture, aa.cpp:147-151
	s::uTex.filter = Filter::Nearest;
	s::FsrRcasCon(s::param0, stops);
	s::range = ivec4(0, 0, w - 1, h - 1);
	Image target;
	target.data = out;
	target.w = w;
	target.h = h;
	target.format = srgb ? Format::RGBA8_SRGB : Format::RGBA8_UNORM;
	for (int y = 0; y < h; y++)
		for (int x = 0; x < w; x++)
		{
			s::vUV = vec2(float(x) + 0.5f, float(y) + 0.5f);
			s::main();
			imageStore(target, ivec2(x, y), s::FragColor);
		}
}

} // extern "C"
