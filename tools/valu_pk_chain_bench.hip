// Dependent v_pk_fma_f32 chains on gfx950 (round 6): does a wave that issues packed fp32 instructions whose operands come from the packed
// instruction before it pay more than the ~8-cycle issue interval every wave pays (tools/valu_bank_bench.hip), and how many independent chains
// hide it?  Decides whether a lighting walk on float2 values over TWO lights at a time (two chains) can work where round 4's single-chain
// form lost 30 %.  Explicit registers; cycles per instruction per SIMD at an assumed clock (compare rows, not absolute numbers).
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_pk_chain_bench.hip -o /tmp/valu_pk_chain_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 4096
template <int MODE> __global__ __launch_bounds__(256) void k(float *out, float a)
{
	float s = 0;
	asm volatile(
	    "v_mov_b32 v0, %1\n v_mov_b32 v1, %1\n v_mov_b32 v2, %1\n v_mov_b32 v3, %1\n v_mov_b32 v4, %1\n v_mov_b32 v5, %1\n v_mov_b32 v6, %1\n v_mov_b32 v7, %1\n"
	    "v_mov_b32 v16, 0\n v_mov_b32 v17, 0\n v_mov_b32 v18, 0\n v_mov_b32 v19, 0\n v_mov_b32 v20, 0\n v_mov_b32 v21, 0\n v_mov_b32 v22, 0\n v_mov_b32 v23, 0\n"
	    "v_mov_b32 v24, 0\n v_mov_b32 v25, 0\n v_mov_b32 v26, 0\n v_mov_b32 v27, 0\n v_mov_b32 v28, 0\n v_mov_b32 v29, 0\n v_mov_b32 v30, 0\n v_mov_b32 v31, 0\n"
	    "s_mov_b32 s20, %2\n"
	    "1:\n"
	    ".if %3 == 0\n" // scalar fma, one dependent chain (reference)
	    "v_fma_f32 v16, v0, v1, v16\n v_fma_f32 v16, v0, v1, v16\n v_fma_f32 v16, v0, v1, v16\n v_fma_f32 v16, v0, v1, v16\n"
	    "v_fma_f32 v16, v0, v1, v16\n v_fma_f32 v16, v0, v1, v16\n v_fma_f32 v16, v0, v1, v16\n v_fma_f32 v16, v0, v1, v16\n"
	    ".endif\n"
	    ".if %3 == 1\n" // pk fma, one dependent chain through the accumulator
	    "v_pk_fma_f32 v[16:17], v[0:1], v[2:3], v[16:17]\n v_pk_fma_f32 v[16:17], v[0:1], v[2:3], v[16:17]\n v_pk_fma_f32 v[16:17], v[0:1], v[2:3], v[16:17]\n v_pk_fma_f32 v[16:17], v[0:1], v[2:3], v[16:17]\n"
	    "v_pk_fma_f32 v[16:17], v[0:1], v[2:3], v[16:17]\n v_pk_fma_f32 v[16:17], v[0:1], v[2:3], v[16:17]\n v_pk_fma_f32 v[16:17], v[0:1], v[2:3], v[16:17]\n v_pk_fma_f32 v[16:17], v[0:1], v[2:3], v[16:17]\n"
	    ".endif\n"
	    ".if %3 == 2\n" // pk fma, two chains interleaved
	    "v_pk_fma_f32 v[16:17], v[0:1], v[2:3], v[16:17]\n v_pk_fma_f32 v[18:19], v[4:5], v[6:7], v[18:19]\n v_pk_fma_f32 v[16:17], v[0:1], v[2:3], v[16:17]\n v_pk_fma_f32 v[18:19], v[4:5], v[6:7], v[18:19]\n"
	    "v_pk_fma_f32 v[16:17], v[0:1], v[2:3], v[16:17]\n v_pk_fma_f32 v[18:19], v[4:5], v[6:7], v[18:19]\n v_pk_fma_f32 v[16:17], v[0:1], v[2:3], v[16:17]\n v_pk_fma_f32 v[18:19], v[4:5], v[6:7], v[18:19]\n"
	    ".endif\n"
	    ".if %3 == 3\n" // pk fma, four chains
	    "v_pk_fma_f32 v[16:17], v[0:1], v[2:3], v[16:17]\n v_pk_fma_f32 v[18:19], v[4:5], v[6:7], v[18:19]\n v_pk_fma_f32 v[20:21], v[0:1], v[6:7], v[20:21]\n v_pk_fma_f32 v[22:23], v[4:5], v[2:3], v[22:23]\n"
	    "v_pk_fma_f32 v[16:17], v[0:1], v[2:3], v[16:17]\n v_pk_fma_f32 v[18:19], v[4:5], v[6:7], v[18:19]\n v_pk_fma_f32 v[20:21], v[0:1], v[6:7], v[20:21]\n v_pk_fma_f32 v[22:23], v[4:5], v[2:3], v[22:23]\n"
	    ".endif\n"
	    ".if %3 == 4\n" // pk chain whose multiplicand is the previous result (a * a form): dependency through a source, not the accumulator
	    "v_pk_mul_f32 v[16:17], v[16:17], v[0:1]\n v_pk_mul_f32 v[16:17], v[16:17], v[0:1]\n v_pk_mul_f32 v[16:17], v[16:17], v[0:1]\n v_pk_mul_f32 v[16:17], v[16:17], v[0:1]\n"
	    "v_pk_mul_f32 v[16:17], v[16:17], v[0:1]\n v_pk_mul_f32 v[16:17], v[16:17], v[0:1]\n v_pk_mul_f32 v[16:17], v[16:17], v[0:1]\n v_pk_mul_f32 v[16:17], v[16:17], v[0:1]\n"
	    ".endif\n"
	    ".if %3 == 5\n" // the walk's mix, one light: pk -> scalar rsq on each half -> pk (two dependent chains of the lane's two pixels joined in packed ops)
	    "v_pk_fma_f32 v[16:17], v[0:1], v[2:3], v[16:17]\n v_rsq_f32 v18, v16\n v_rsq_f32 v19, v17\n v_pk_mul_f32 v[16:17], v[18:19], v[16:17]\n"
	    "v_pk_fma_f32 v[16:17], v[0:1], v[2:3], v[16:17]\n v_med3_f32 v18, v16, 0, 1.0\n v_med3_f32 v19, v17, 0, 1.0\n v_pk_mul_f32 v[16:17], v[18:19], v[16:17]\n"
	    ".endif\n"
	    ".if %3 == 6\n" // the same mix, two lights interleaved
	    "v_pk_fma_f32 v[16:17], v[0:1], v[2:3], v[16:17]\n v_pk_fma_f32 v[24:25], v[4:5], v[6:7], v[24:25]\n v_rsq_f32 v18, v16\n v_rsq_f32 v26, v24\n v_rsq_f32 v19, v17\n v_rsq_f32 v27, v25\n"
	    "v_pk_mul_f32 v[16:17], v[18:19], v[16:17]\n v_pk_mul_f32 v[24:25], v[26:27], v[24:25]\n"
	    "v_pk_fma_f32 v[16:17], v[0:1], v[2:3], v[16:17]\n v_pk_fma_f32 v[24:25], v[4:5], v[6:7], v[24:25]\n v_med3_f32 v18, v16, 0, 1.0\n v_med3_f32 v26, v24, 0, 1.0\n v_med3_f32 v19, v17, 0, 1.0\n v_med3_f32 v27, v25, 0, 1.0\n"
	    "v_pk_mul_f32 v[16:17], v[18:19], v[16:17]\n v_pk_mul_f32 v[24:25], v[26:27], v[24:25]\n"
	    ".endif\n"
	    ".if %3 == 7\n" // the scalar form of mode 5's work (what the compiled walk mostly is): 2 x (fma, rsq, mul, fma, med3, mul)
	    "v_fma_f32 v16, v0, v2, v16\n v_fma_f32 v17, v1, v3, v17\n v_rsq_f32 v18, v16\n v_rsq_f32 v19, v17\n v_mul_f32 v16, v18, v16\n v_mul_f32 v17, v19, v17\n"
	    "v_fma_f32 v16, v0, v2, v16\n v_fma_f32 v17, v1, v3, v17\n v_med3_f32 v18, v16, 0, 1.0\n v_med3_f32 v19, v17, 0, 1.0\n v_mul_f32 v16, v18, v16\n v_mul_f32 v17, v19, v17\n"
	    ".endif\n"
	    "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n"
	    "v_add_f32 %0, v16, v17\n v_add_f32 %0, %0, v18\n v_add_f32 %0, %0, v20\n v_add_f32 %0, %0, v24\n v_add_f32 %0, %0, v28\n"
	    : "=v"(s)
	    : "v"(a + threadIdx.x * 1e-9f), "s"(ITER), "n"(MODE)
	    : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31",
	      "s20", "scc");
	if (s == 12345.678f)
		out[0] = s;
}
template <int MODE> void run(const char *name, float *d, int per_iter)
{
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	printf("%-86s", name);
	for (int waves = 1; waves <= 8; waves = waves == 4 ? 5 : waves == 5 ? 8 : waves * 2)
	{
		const int blocks = 256 * waves;
		hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
		hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
		hipEventRecord(e0);
		hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
		hipEventRecord(e1);
		hipEventSynchronize(e1);
		float ms;
		hipEventElapsedTime(&ms, e0, e1);
		printf("  %dw %6.2f", waves, ms * 1e-3 * 2.0e9 / (double(waves) * ITER)); // cycles per ITERATION per SIMD at 2.0 GHz
	}
	printf("   (%d instructions per iteration)\n", per_iter);
}
int main()
{
	float *d;
	hipMalloc(&d, 4);
	printf("cycles per loop iteration per SIMD at an assumed 2.0 GHz, by waves per SIMD\n");
	run<0>("scalar fma, 1 dependent chain", d, 8);
	run<1>("pk fma, 1 dependent chain (through the accumulator)", d, 8);
	run<2>("pk fma, 2 chains interleaved", d, 8);
	run<3>("pk fma, 4 chains interleaved", d, 8);
	run<4>("pk mul, 1 dependent chain (through a multiplicand)", d, 8);
	run<5>("walk mix, one light: pk, 2 rsq, pk, pk, 2 med3, pk (dependent)", d, 8);
	run<6>("walk mix, two lights interleaved (twice the work of the row above)", d, 16);
	run<7>("walk mix, scalar form of one light (the work of the one-light row)", d, 12);
	return 0;
}
