// ORACLE — TEST INFRASTRUCTURE ONLY.  Shader-side glue of post/ffx-fsr/upscale.frag for its FP16 = 1 variant (upscale.frag:
// 8-14,25-26): the defines and the three gather callbacks, restated, followed by the reference's own headers.  upscale.frag
// itself is not used for this variant because it passes a vec3 to FsrEasuH's f16vec3 out parameter (an implicit GLSL
// conversion with no C++ counterpart); the runner calls FsrEasuH directly.  @REF@ is filled in by the Makefile.
#define A_GLSL 1
#define A_GPU 1
#define FSR_EASU_H 1
#define A_HALF 1
uniform sampler2D uTex;
f16vec4 FsrEasuRH(vec2 p) { return f16vec4(textureGather(uTex, p, 0)); }
f16vec4 FsrEasuGH(vec2 p) { return f16vec4(textureGather(uTex, p, 1)); }
f16vec4 FsrEasuBH(vec2 p) { return f16vec4(textureGather(uTex, p, 2)); }
#include "@REF@/assets/shaders/post/ffx-a/ffx_a.h"
#include "@REF@/assets/shaders/post/ffx-fsr/ffx_fsr1.h"
