/* What does a wave64 VALU instruction cost on gfx950, by opcode?  (round 6; measurement aid)
 *
 * The fused kernels are bound by VALU issue (scope_trio_kernel: 25 instructions per texel, the VALU busy two thirds of the launch
 * at four waves per SIMD).  Their instruction counts have been priced at "four cycles each" since round 2 -- but the chip's fp32 rate
 * says a plain v_fma_f32 goes through a SIMD faster than that, so the price may depend on the opcode.  This program times each
 * opcode the kernels' slots are made of: eight independent dependency chains per lane, 128 instructions per loop body, 1 / 2 / 4
 * waves per SIMD, one workgroup per CU; it prints cycles per wave instruction per SIMD (at the clock s_memtime / wall time gives).
 *
 * build: hipcc --offload-arch=gfx950 -O3 -o valu_rate_bench valu_rate_bench.hip */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef uint32_t u32;

#define CHECK(x)                                                                                                                  \
	do {                                                                                                                      \
		hipError_t e_ = (x);                                                                                              \
		if (e_ != hipSuccess) {                                                                                           \
			fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));                                   \
			exit(1);                                                                                                  \
		}                                                                                                                 \
	} while (0)

#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define R16(X) R8(X) R8(X)
#define R128(X) R16(X) R16(X) R16(X) R16(X) R16(X) R16(X) R16(X) R16(X)

/* one kernel per opcode: ASM uses %0 (the chain register, read and written), %1 and %2 (lane constants) */
#define OPKERNEL(name, ASM)                                                                                                       \
	__global__ void __launch_bounds__(1024) k_##name(u32 iters, u32 *out, unsigned long long *clk)                           \
	{                                                                                                                         \
		u32 c[8];                                                                                                         \
		for (int j = 0; j < 8; j++)                                                                                       \
			c[j] = threadIdx.x * 2654435761u + j * 40503u + blockIdx.x;                                               \
		u32 a = threadIdx.x | 0x01010101u, b = (threadIdx.x << 3) ^ 0x00ff00ffu;                                          \
		asm volatile("" : "+v"(a), "+v"(b));                                                                              \
		const unsigned long long t0 = __builtin_readcyclecounter();                                                       \
		for (u32 it = 0; it < iters; it++) {                                                                              \
			R128(STEP_##name)                                                                                         \
		}                                                                                                                 \
		const unsigned long long t1 = __builtin_readcyclecounter();                                                       \
		u32 s = 0;                                                                                                        \
		for (int j = 0; j < 8; j++)                                                                                       \
			s ^= c[j];                                                                                                \
		out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                                   \
		if (threadIdx.x == 0)                                                                                             \
			clk[blockIdx.x] = t1 - t0;                                                                                \
	}

#define STEP_add(j) asm volatile("v_add_u32 %0, %0, %1" : "+v"(c[j]) : "v"(a), "v"(b));
OPKERNEL(add, )
#define STEP_and(j) asm volatile("v_and_b32 %0, %0, %1" : "+v"(c[j]) : "v"(a), "v"(b));
OPKERNEL(and, )
#define STEP_lshr(j) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(c[j]) : "v"(a), "v"(b));
OPKERNEL(lshr, )
#define STEP_fma(j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(c[j]) : "v"(a), "v"(b));
OPKERNEL(fma, )
#define STEP_mulf(j) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(c[j]) : "v"(a), "v"(b));
OPKERNEL(mulf, )
#define STEP_mad24(j) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(c[j]) : "v"(a), "v"(b));
OPKERNEL(mad24, )
#define STEP_mul24(j) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(c[j]) : "v"(a), "v"(b));
OPKERNEL(mul24, )
#define STEP_mul24sdwa(j)                                                                                                         \
	asm volatile("v_mul_i32_i24_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(c[j]) : "v"(a), "v"(b));
OPKERNEL(mul24sdwa, )
#define STEP_perm(j) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(c[j]) : "v"(a), "v"(b));
OPKERNEL(perm, )
#define STEP_add3(j) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(c[j]) : "v"(a), "v"(b));
OPKERNEL(add3, )
#define STEP_lshladd(j) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(c[j]) : "v"(a), "v"(b));
OPKERNEL(lshladd, )
#define STEP_dot4(j) asm volatile("v_dot4_u32_u8 %0, %0, %1, %2" : "+v"(c[j]) : "v"(a), "v"(b));
OPKERNEL(dot4, )
#define STEP_pkmin(j) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(c[j]) : "v"(a), "v"(b));
OPKERNEL(pkmin, )
#define STEP_pkadd(j) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(c[j]) : "v"(a), "v"(b));
OPKERNEL(pkadd, )
#define STEP_alignbit(j) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(c[j]) : "v"(a), "v"(b));
OPKERNEL(alignbit, )
#define STEP_bfe(j) asm volatile("v_bfe_u32 %0, %0, 3, 17" : "+v"(c[j]) : "v"(a), "v"(b));
OPKERNEL(bfe, )
#define STEP_min(j) asm volatile("v_min_u32 %0, %0, %1" : "+v"(c[j]) : "v"(a), "v"(b));
OPKERNEL(min, )
#define STEP_min3(j) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(c[j]) : "v"(a), "v"(b));
OPKERNEL(min3, )
#define STEP_andor(j) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(c[j]) : "v"(a), "v"(b));
OPKERNEL(andor, )
#define STEP_umad24(j) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(c[j]) : "v"(a), "v"(b));
OPKERNEL(umad24, )
#define STEP_bitop3(j) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(c[j]) : "v"(a), "v"(b));
OPKERNEL(bitop3, )
#define STEP_mullo(j) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(c[j]) : "v"(a), "v"(b));
OPKERNEL(mullo, )
#define STEP_mov(j) asm volatile("v_mov_b32 %0, %1" : "+v"(c[j]) : "v"(a), "v"(b));
OPKERNEL(mov, )
#define STEP_movdpp(j) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(c[j]) : "v"(a), "v"(b));
OPKERNEL(movdpp, )
#define STEP_cmp(j) asm volatile("v_cmp_eq_u32 vcc, %0, %1" : "+v"(c[j]) : "v"(a), "v"(b) : "vcc");
OPKERNEL(cmp, )
#define STEP_cndmask(j) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(c[j]) : "v"(a), "v"(b) : "vcc");
OPKERNEL(cndmask, )
#define STEP_cvtub(j) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(c[j]) : "v"(a), "v"(b));
OPKERNEL(cvtub, )
#define STEP_addsdwa(j)                                                                                                           \
	asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(c[j]) : "v"(a), "v"(b));
OPKERNEL(addsdwa, )

#define STEP_or(j) asm volatile("v_or_b32 %0, %0, %1" : "+v"(c[j]) : "v"(a), "v"(b));
OPKERNEL(or, )
#define STEP_xor(j) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(c[j]) : "v"(a), "v"(b));
OPKERNEL(xor, )
#define STEP_sub(j) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(c[j]) : "v"(a), "v"(b));
OPKERNEL(sub, )
#define STEP_lshl(j) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(c[j]) : "v"(a), "v"(b));
OPKERNEL(lshl, )
#define STEP_ashr(j) asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(c[j]) : "v"(a), "v"(b));
OPKERNEL(ashr, )
#define STEP_addf(j) asm volatile("v_add_f32 %0, %0, %1" : "+v"(c[j]) : "v"(a), "v"(b));
OPKERNEL(addf, )
#define STEP_minf(j) asm volatile("v_min_f32 %0, %0, %1" : "+v"(c[j]) : "v"(a), "v"(b));
OPKERNEL(minf, )
#define STEP_maxf(j) asm volatile("v_max_f32 %0, %0, %1" : "+v"(c[j]) : "v"(a), "v"(b));
OPKERNEL(maxf, )
#define STEP_fmac(j) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(c[j]) : "v"(a), "v"(b));
OPKERNEL(fmac, )
#define STEP_cvtu32(j) asm volatile("v_cvt_u32_f32 %0, %0" : "+v"(c[j]) : "v"(a), "v"(b));
OPKERNEL(cvtu32, )
#define STEP_cvtf32(j) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(c[j]) : "v"(a), "v"(b));
OPKERNEL(cvtf32, )
#define STEP_cvtub0(j) asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(c[j]) : "v"(a), "v"(b));
OPKERNEL(cvtub0, )
#define STEP_lshlor(j) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(c[j]) : "v"(a), "v"(b));
OPKERNEL(lshlor, )
#define STEP_or3(j) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(c[j]) : "v"(a), "v"(b));
OPKERNEL(or3, )
#define STEP_xad(j) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(c[j]) : "v"(a), "v"(b));
OPKERNEL(xad, )
#define STEP_max(j) asm volatile("v_max_u32 %0, %0, %1" : "+v"(c[j]) : "v"(a), "v"(b));
OPKERNEL(max, )
#define STEP_cndmask2(j) asm volatile("v_cndmask_b32 %0, %0, %1, s[10:11]" : "+v"(c[j]) : "v"(a), "v"(b));
OPKERNEL(cndmask2, )
#define STEP_addco(j) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(c[j]) : "v"(a), "v"(b) : "vcc");
OPKERNEL(addco, )
#define STEP_mbcnt(j) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(c[j]) : "v"(a), "v"(b));
OPKERNEL(mbcnt, )
#define STEP_sad(j) asm volatile("v_sad_u32 %0, %0, %1, %2" : "+v"(c[j]) : "v"(a), "v"(b));
OPKERNEL(sad, )
/* two-dword results: each lane's chain is a register pair */
#define PKKERNEL(name, ASM)                                                                                                       \
	__global__ void __launch_bounds__(1024) k_##name(u32 iters, u32 *out, unsigned long long *clk)                           \
	{                                                                                                                         \
		typedef float f2 __attribute__((ext_vector_type(2)));                                                             \
		f2 c[8], a = {1.0000001f, 0.9999999f}, b = {1e-9f, -1e-9f};                                                        \
		for (int j = 0; j < 8; j++)                                                                                       \
			c[j] = (f2){(float)(threadIdx.x + j), (float)(blockIdx.x + j)};                                           \
		asm volatile("" : "+v"(a), "+v"(b));                                                                              \
		for (u32 it = 0; it < iters; it++) {                                                                              \
			R128(STEP_##name)                                                                                         \
		}                                                                                                                 \
		float s = 0;                                                                                                      \
		for (int j = 0; j < 8; j++)                                                                                       \
			s += c[j].x + c[j].y;                                                                                     \
		out[blockIdx.x * blockDim.x + threadIdx.x] = __float_as_uint(s);                                                   \
		if (threadIdx.x == 0)                                                                                             \
			clk[blockIdx.x] = 0;                                                                                      \
	}
#define STEP_pkfma(j) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(c[j]) : "v"(a), "v"(b));
PKKERNEL(pkfma, )
#define STEP_pkmulf(j) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(c[j]) : "v"(a), "v"(b));
PKKERNEL(pkmulf, )
#define STEP_pkaddf(j) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(c[j]) : "v"(a), "v"(b));
PKKERNEL(pkaddf, )

typedef void (*kern_t)(u32, u32 *, unsigned long long *);

static void run(const char *name, kern_t k, int num_cu, u32 *out, unsigned long long *clk)
{
	const u32 iters = 2048;
	printf("%-22s", name);
	for (int threads = 256; threads <= 1024; threads *= 2) {
		hipEvent_t ev[2];
		CHECK(hipEventCreate(&ev[0]));
		CHECK(hipEventCreate(&ev[1]));
		hipLaunchKernelGGL(k, dim3(num_cu), dim3(threads), 0, 0, iters / 4, out, clk);
		CHECK(hipDeviceSynchronize());
		float best = 1e30f;
		unsigned long long ticks = 0;
		for (int rep = 0; rep < 3; rep++) {
			CHECK(hipEventRecord(ev[0]));
			hipLaunchKernelGGL(k, dim3(num_cu), dim3(threads), 0, 0, iters, out, clk);
			CHECK(hipEventRecord(ev[1]));
			CHECK(hipEventSynchronize(ev[1]));
			float ms;
			CHECK(hipEventElapsedTime(&ms, ev[0], ev[1]));
			if (ms < best) {
				best = ms;
				CHECK(hipMemcpy(&ticks, clk, 8, hipMemcpyDeviceToHost));
			}
		}
		const double waves_per_simd = threads / 256.0, insts = (double)iters * 128.0 * waves_per_simd;
		/* s_memtime ticks at a constant 100 MHz on this chip; the wall time and the nominal clock give cycles */
		printf("  %d w/SIMD: %6.3f ns/inst", (int)waves_per_simd, (double)best * 1e6 / insts);
		(void)ticks;
		CHECK(hipEventDestroy(ev[0]));
		CHECK(hipEventDestroy(ev[1]));
	}
	printf("\n");
}

int main()
{
	hipDeviceProp_t prop;
	CHECK(hipGetDeviceProperties(&prop, 0));
	const int num_cu = prop.multiProcessorCount;
	printf("%s, %d CUs, nominal %.2f GHz (a cycle = %.3f ns): ns per wave64 instruction per SIMD\n", prop.name, num_cu, prop.clockRate * 1e-6,
	       1e6 / prop.clockRate);
	u32 *out;
	unsigned long long *clk;
	CHECK(hipMalloc(&out, (size_t)num_cu * 1024 * 4));
	CHECK(hipMalloc(&clk, (size_t)num_cu * 8));
#define RUN(n) run(#n, k_##n, num_cu, out, clk)
	RUN(add);
	RUN(and);
	RUN(lshr);
	RUN(mov);
	RUN(fma);
	RUN(mulf);
	RUN(mad24);
	RUN(umad24);
	RUN(mul24);
	RUN(mul24sdwa);
	RUN(addsdwa);
	RUN(perm);
	RUN(add3);
	RUN(lshladd);
	RUN(andor);
	RUN(bitop3);
	RUN(bfe);
	RUN(alignbit);
	RUN(min);
	RUN(min3);
	RUN(dot4);
	RUN(pkmin);
	RUN(pkadd);
	RUN(mullo);
	RUN(movdpp);
	RUN(cmp);
	RUN(cndmask);
	RUN(cvtub);
	RUN(or);
	RUN(xor);
	RUN(sub);
	RUN(lshl);
	RUN(ashr);
	RUN(addf);
	RUN(minf);
	RUN(maxf);
	RUN(fmac);
	RUN(pkfma);
	RUN(pkmulf);
	RUN(pkaddf);
	RUN(cvtu32);
	RUN(cvtf32);
	RUN(cvtub0);
	RUN(lshlor);
	RUN(or3);
	RUN(xad);
	RUN(max);
	RUN(cndmask2);
	RUN(addco);
	RUN(mbcnt);
	RUN(sad);
	return 0;
}
