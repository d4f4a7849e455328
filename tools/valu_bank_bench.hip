// VALU operand / dependency microbenchmark for gfx950 (round 4): why a compiled fp32 stream issues at ~4 cycles per wave64 instruction when
// the per-opcode peak (tools/valu_bench.hip) is ~2.3.  Measures v_fma_f32 with explicit registers:
//   * source registers in distinct VGPR banks (index mod 4) against all in one bank, constant sources against sources that change,
//   * ILP (independent chains per wave) x waves per SIMD, i.e. the dependent-issue latency a wave sees.
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_bank_bench.hip -o /tmp/valu_bank_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 4096
#define REP8(X) X X X X X X X X

// eight fma per iteration; CHAINS of them independent (destination registers v16.., each its own accumulator), sources per MODE
template <int MODE> __global__ __launch_bounds__(256) void k(float *out, float a)
{
	float s = 0;
	asm volatile(
	    "v_mov_b32 v0, %1\n v_mov_b32 v1, %1\n v_mov_b32 v2, %1\n v_mov_b32 v3, %1\n v_mov_b32 v4, %1\n v_mov_b32 v5, %1\n v_mov_b32 v6, %1\n v_mov_b32 v7, %1\n"
	    "v_mov_b32 v8, %1\n v_mov_b32 v9, %1\n v_mov_b32 v10, %1\n v_mov_b32 v11, %1\n v_mov_b32 v12, %1\n v_mov_b32 v13, %1\n v_mov_b32 v14, %1\n v_mov_b32 v15, %1\n"
	    "v_mov_b32 v16, 0\n v_mov_b32 v17, 0\n v_mov_b32 v18, 0\n v_mov_b32 v19, 0\n v_mov_b32 v20, 0\n v_mov_b32 v21, 0\n v_mov_b32 v22, 0\n v_mov_b32 v23, 0\n"
	    "s_mov_b32 s20, %2\n"
	    "1:\n"
	    ".if %3 == 0\n" // 8 chains, sources v0 v1 (banks 0, 1) + accumulator
	    "v_fma_f32 v16, v0, v1, v16\n v_fma_f32 v17, v0, v1, v17\n v_fma_f32 v18, v0, v1, v18\n v_fma_f32 v19, v0, v1, v19\n"
	    "v_fma_f32 v20, v0, v1, v20\n v_fma_f32 v21, v0, v1, v21\n v_fma_f32 v22, v0, v1, v22\n v_fma_f32 v23, v0, v1, v23\n"
	    ".endif\n"
	    ".if %3 == 1\n" // 8 chains, sources v0 v4 (both bank 0)
	    "v_fma_f32 v16, v0, v4, v16\n v_fma_f32 v17, v0, v4, v17\n v_fma_f32 v18, v0, v4, v18\n v_fma_f32 v19, v0, v4, v19\n"
	    "v_fma_f32 v20, v0, v4, v20\n v_fma_f32 v21, v0, v4, v21\n v_fma_f32 v22, v0, v4, v22\n v_fma_f32 v23, v0, v4, v23\n"
	    ".endif\n"
	    ".if %3 == 2\n" // 8 chains, three sources of one bank: v0 v4 and accumulators v16 v20 v24... -> use acc = bank 0 only: v16, v20 alternate (4 chains deep dependency 2)
	    "v_fma_f32 v16, v0, v4, v16\n v_fma_f32 v20, v8, v12, v20\n v_fma_f32 v24, v0, v4, v24\n v_fma_f32 v28, v8, v12, v28\n"
	    "v_fma_f32 v32, v0, v4, v32\n v_fma_f32 v36, v8, v12, v36\n v_fma_f32 v40, v0, v4, v40\n v_fma_f32 v44, v8, v12, v44\n"
	    ".endif\n"
	    ".if %3 == 3\n" // 8 chains, sources change every instruction, all banks distinct per instruction (a: bank 0/1.., b: +1, acc: +2)
	    "v_fma_f32 v18, v0, v1, v18\n v_fma_f32 v19, v5, v6, v19\n v_fma_f32 v24, v10, v11, v24\n v_fma_f32 v25, v15, v12, v25\n"
	    "v_fma_f32 v22, v4, v9, v22\n v_fma_f32 v23, v13, v2, v23\n v_fma_f32 v28, v14, v3, v28\n v_fma_f32 v29, v7, v8, v29\n"
	    ".endif\n"
	    ".if %3 == 4\n" // sources change every instruction, all three of one bank
	    "v_fma_f32 v16, v0, v4, v16\n v_fma_f32 v17, v5, v9, v17\n v_fma_f32 v18, v10, v14, v18\n v_fma_f32 v19, v15, v3, v19\n"
	    "v_fma_f32 v20, v8, v12, v20\n v_fma_f32 v21, v13, v1, v21\n v_fma_f32 v22, v2, v6, v22\n v_fma_f32 v23, v7, v11, v23\n"
	    ".endif\n"
	    ".if %3 == 5\n" // 4 chains (each twice per iteration)
	    "v_fma_f32 v16, v0, v1, v16\n v_fma_f32 v17, v0, v1, v17\n v_fma_f32 v18, v0, v1, v18\n v_fma_f32 v19, v0, v1, v19\n"
	    "v_fma_f32 v16, v0, v1, v16\n v_fma_f32 v17, v0, v1, v17\n v_fma_f32 v18, v0, v1, v18\n v_fma_f32 v19, v0, v1, v19\n"
	    ".endif\n"
	    ".if %3 == 6\n" // 2 chains
	    "v_fma_f32 v16, v0, v1, v16\n v_fma_f32 v17, v0, v1, v17\n v_fma_f32 v16, v0, v1, v16\n v_fma_f32 v17, v0, v1, v17\n"
	    "v_fma_f32 v16, v0, v1, v16\n v_fma_f32 v17, v0, v1, v17\n v_fma_f32 v16, v0, v1, v16\n v_fma_f32 v17, v0, v1, v17\n"
	    ".endif\n"
	    ".if %3 == 7\n" // 1 chain
	    "v_fma_f32 v16, v0, v1, v16\n v_fma_f32 v16, v0, v1, v16\n v_fma_f32 v16, v0, v1, v16\n v_fma_f32 v16, v0, v1, v16\n"
	    "v_fma_f32 v16, v0, v1, v16\n v_fma_f32 v16, v0, v1, v16\n v_fma_f32 v16, v0, v1, v16\n v_fma_f32 v16, v0, v1, v16\n"
	    ".endif\n"
	    ".if %3 == 8\n" // 2 chains of rsq -> mul (transcendental latency)
	    "v_rsq_f32 v16, v16\n v_rsq_f32 v17, v17\n v_mul_f32 v16, v16, v0\n v_mul_f32 v17, v17, v0\n"
	    "v_rsq_f32 v16, v16\n v_rsq_f32 v17, v17\n v_mul_f32 v16, v16, v0\n v_mul_f32 v17, v17, v0\n"
	    ".endif\n"
	    ".if %3 == 9\n" // the same work, 8 independent
	    "v_rsq_f32 v16, v16\n v_rsq_f32 v17, v17\n v_mul_f32 v18, v18, v0\n v_mul_f32 v19, v19, v0\n"
	    "v_rsq_f32 v20, v20\n v_rsq_f32 v21, v21\n v_mul_f32 v22, v22, v0\n v_mul_f32 v23, v23, v0\n"
	    ".endif\n"
	    ".if %3 == 10\n" // fma mixed with v_pk_fma (independent)
	    "v_fma_f32 v16, v0, v1, v16\n v_pk_fma_f32 v[18:19], v[2:3], v[4:5], v[18:19]\n v_fma_f32 v17, v0, v1, v17\n v_pk_fma_f32 v[20:21], v[2:3], v[4:5], v[20:21]\n"
	    "v_fma_f32 v22, v0, v1, v22\n v_pk_fma_f32 v[24:25], v[2:3], v[4:5], v[24:25]\n v_fma_f32 v23, v0, v1, v23\n v_pk_fma_f32 v[26:27], v[2:3], v[4:5], v[26:27]\n"
	    ".endif\n"
	    ".if %3 == 11\n" // v_pk_fma whose sources are unaligned-bank pairs changing every instruction
	    "v_pk_fma_f32 v[16:17], v[0:1], v[4:5], v[16:17]\n v_pk_fma_f32 v[18:19], v[2:3], v[6:7], v[18:19]\n v_pk_fma_f32 v[20:21], v[8:9], v[12:13], v[20:21]\n v_pk_fma_f32 v[22:23], v[10:11], v[14:15], v[22:23]\n"
	    "v_pk_fma_f32 v[24:25], v[0:1], v[6:7], v[24:25]\n v_pk_fma_f32 v[26:27], v[2:3], v[4:5], v[26:27]\n v_pk_fma_f32 v[28:29], v[8:9], v[14:15], v[28:29]\n v_pk_fma_f32 v[30:31], v[10:11], v[12:13], v[30:31]\n"
	    ".endif\n"
	    "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n"
	    "v_add_f32 %0, v16, v17\n v_add_f32 %0, %0, v18\n v_add_f32 %0, %0, v20\n v_add_f32 %0, %0, v24\n v_add_f32 %0, %0, v28\n"
	    : "=v"(s)
	    : "v"(a + threadIdx.x * 1e-9f), "s"(ITER), "n"(MODE)
	    : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22",
	      "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v36", "v40", "v44", "s20", "scc");
	if (s == 12345.678f)
		out[0] = s;
}
template <int MODE> void run(const char *name, float *d)
{
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	printf("%-58s", name);
	for (int waves = 1; waves <= 8; waves *= 2)
	{
		const int blocks = 256 * waves; // x 4 waves / 1024 SIMDs = `waves` per SIMD
		hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
		hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
		hipEventRecord(e0);
		hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
		hipEventRecord(e1);
		hipEventSynchronize(e1);
		float ms;
		hipEventElapsedTime(&ms, e0, e1);
		printf("  %dw %5.2f", waves, ms * 1e-3 * 2.0e9 / (double(waves) * ITER * 8)); // cycles per instruction per SIMD at 2.0 GHz
	}
	printf("\n");
}
int main()
{
	float *d;
	hipMalloc(&d, 4);
	printf("cycles per wave64 instruction per SIMD at an assumed 2.0 GHz, by waves per SIMD\n");
	run<0>("fma, 8 chains, 2 constant sources in banks 0,1", d);
	run<1>("fma, 8 chains, 2 constant sources both in bank 0", d);
	run<2>("fma, 8 chains, all three sources in bank 0", d);
	run<3>("fma, 8 chains, sources change, three banks", d);
	run<4>("fma, 8 chains, sources change, one bank", d);
	run<5>("fma, 4 chains", d);
	run<6>("fma, 2 chains", d);
	run<7>("fma, 1 chain", d);
	run<8>("rsq -> mul, 2 chains (4 rsq + 4 mul per iteration)", d);
	run<9>("rsq, mul independent (4 + 4)", d);
	run<10>("fma + pk_fma alternating, independent", d);
	run<11>("pk_fma, sources change", d);
	return 0;
}
