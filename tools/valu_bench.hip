// VALU issue-rate microbenchmark for gfx950: cycles per wave64 instruction per SIMD for a few opcodes.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define ITER 2048
template <int OP> __global__ __launch_bounds__(256) void k(float *out, float a, float b)
{
	float r[8]; f32x2 p[8];
	for (int i = 0; i < 8; i++) { r[i] = a + i + threadIdx.x; p[i] = f32x2{a + i, b + threadIdx.x}; }
	f32x2 pa = {a, b}, pb = {b, a};
	for (int it = 0; it < ITER; it++)
	{
#pragma unroll
		for (int i = 0; i < 8; i++)
		{
			if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
			if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pa), "v"(pb));
			if (OP == 2) asm volatile("v_max_f32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
			if (OP == 3) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
			if (OP == 4) asm volatile("v_rsq_f32 %0, %0" : "+v"(r[i]));
			if (OP == 5) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
			if (OP == 6) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pa));
			if (OP == 7) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
			if (OP == 8) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(r[i]), "v"(a) : "vcc");
			if (OP == 9) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i]));
			if (OP == 10) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "s"(a), "v"(b));
			if (OP == 11) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(p[i]) : "v"(pa), "v"(pb));
			if (OP == 12) asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(r[i]));
			if (OP == 13) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pa));
			if (OP == 14) asm volatile("v_fmaak_f32 %0, %0, %1, 0x41200000" : "+v"(r[i]) : "v"(a));
			if (OP == 15) asm volatile("v_fma_f32 %0, %0, %1, 2.0" : "+v"(r[i]) : "v"(a));
			if (OP == 16) asm volatile("v_fma_f32 %0, %0, %1, %2 clamp" : "+v"(r[i]) : "v"(a), "v"(b));
			if (OP == 17) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(r[i]) : "s"(a));
			if (OP == 18) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(r[i]) : "s"(a));
			if (OP == 19) asm volatile("v_max_f32 %0, 1.0, %0" : "+v"(r[i]));
			if (OP == 20) asm volatile("v_min_f32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
			if (OP == 21) asm volatile("v_mov_b32 %0, %1" : "=v"(r[i]) : "s"(a));
			if (OP == 22) asm volatile("v_fma_f32 %0, -%0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
			if (OP == 23) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
			if (OP == 24) asm volatile("v_fma_f32 %0, %0, %1, %2 mul:2" : "+v"(r[i]) : "v"(a), "v"(b));
			if (OP == 25) asm volatile("v_mul_f32 %0, 0.5, %0" : "+v"(r[i]));
			if (OP == 26) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(r[i]));
			if (OP == 27) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
			if (OP == 28) asm volatile("v_fma_mix_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
			if (OP == 29) asm volatile("v_mul_f32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(r[i]) : "v"(a));
		}
	}
	float s = 0; for (int i = 0; i < 8; i++) s += r[i] + p[i].x + p[i].y;
	if (s == 12345.678f) out[0] = s;
}
template <int OP> void run(const char *name, float *d)
{
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	const int blocks = 256 * 8;
	hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0f, 0.5f);
	hipEventRecord(e0);
	hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0f, 0.5f);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	// waves per SIMD = blocks*4/1024 ; instr per wave = ITER*8
	double instr_per_simd = double(blocks) * 4 / 1024 * ITER * 8;
	printf("%-28s %8.3f ms  -> %.2f cycles/instr/SIMD @2.4GHz (%.2f @2.0GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / instr_per_simd, ms * 1e-3 * 2.0e9 / instr_per_simd);
}
int main()
{
	float *d; hipMalloc(&d, 4);
	run<0>("v_fma_f32", d); run<1>("v_pk_fma_f32", d); run<11>("v_pk_fma_f32 op_sel bcast", d); run<2>("v_max_f32", d); run<3>("v_med3_f32", d);
	run<4>("v_rsq_f32", d); run<9>("v_rcp_f32", d); run<5>("v_mul_f32", d); run<6>("v_pk_mul_f32", d); run<13>("v_pk_add_f32", d); run<7>("v_add_f32", d);
	run<8>("v_cmp_lt_f32", d); run<10>("v_fma_f32 sgpr src", d); run<12>("v_cvt_f16_f32", d);
	run<14>("v_fmaak_f32 literal", d); run<15>("v_fma_f32 inline const", d); run<16>("v_fma_f32 clamp", d); run<17>("v_mul_f32 sgpr", d);
	run<18>("v_sub_f32 sgpr", d); run<19>("v_max_f32 inline", d); run<20>("v_min_f32", d); run<21>("v_mov_b32 sgpr", d);
	run<22>("v_fma_f32 neg mod", d); run<23>("v_fmac_f32", d); run<24>("v_fma_f32 omod", d); run<25>("v_mul_f32 inline 0.5", d);
	run<26>("v_cvt_f32_f16", d); run<27>("v_and_b32", d); run<28>("v_fma_mix_f32", d); run<29>("v_mul_f32_dpp", d);
	return 0;
}
