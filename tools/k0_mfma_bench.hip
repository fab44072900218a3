/* K0's chroma rows on the matrix core: is it exact, and what does it leave of the VALU's time?  (round 6; measurement aid)
 *
 * The fused kernels are short of VALU issue slots and K0's two chroma rows are 9 of their instructions per texel: one masked
 * byte, then per row two SDWA byte multiplies, a multiply-add and a three-way add in 2^-24 fixed point (scope_k0_row,
 * csrc/scope_device.h).  The rows are a 2 x 3 matrix times the texel's bytes -- a contraction, which v_mfma_i32_16x16x32_i8 does
 * on int8 operands: the 24-bit coefficients go in as three balanced signed byte limbs (three MFMAs), the texel bytes as they are
 * loaded with their top bits flipped (x - 128 as int8; the 128 * sum of the coefficients joins the constant term).  B = the
 * lanes' own eight bytes (two texels), A = the coefficients laid out so that accumulator jj of lane l is row (jj & 1) of the
 * lane's OWN texel (jj >> 1): lane l holds column l % 16, K block l / 16 of B and reads back rows 4 (l / 16) .. + 3 of column
 * l % 16, so A's row 4 q + jj is nonzero in K block q alone, bytes 4 (jj >> 1) .. + 3.  Then E = D0 + (D1 << 8) + (D2 << 16):
 * two v_lshl_add_u32 per value.  Per slot of two texels: 2 flips + 3 MFMA issues + 8 adds against 18.
 *
 * This program (i) checks the MFMA form against the VALU form on random texels, every lane, and (ii) times both forms in a
 * register-only loop with `extra` independent VALU instructions per slot standing in for the rest of a kernel's slot, at 4
 * waves per SIMD: cycles per slot per SIMD.
 *
 * build: hipcc --offload-arch=gfx950 -O3 -o k0_mfma_bench k0_mfma_bench.hip */
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef uint32_t u32;
typedef int v4i __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                                                                  \
	do {                                                                                                                      \
		hipError_t e_ = (x);                                                                                              \
		if (e_ != hipSuccess) {                                                                                           \
			fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));                                   \
			exit(1);                                                                                                  \
		}                                                                                                                 \
	} while (0)

struct Rows {
	int fu[4], fv[4]; /* coefficient * 2^24 for R, G, B and the constant term (with the guard), as ScopeK0 has them */
};

static __device__ __forceinline__ int mul24_b1(u32 px, int f)
{
	int r;
	asm("v_mul_i32_i24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(r) : "v"(px), "v"(f));
	return r;
}
static __device__ __forceinline__ int mul24_b2(u32 px, int f)
{
	int r;
	asm("v_mul_i32_i24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(r) : "v"(px), "v"(f));
	return r;
}
static __device__ __forceinline__ u32 row_valu(u32 px, int b, const int (&f)[4])
{
	return (u32)((__mul24(f[2], b) + f[3]) + mul24_b1(px, f[1]) + mul24_b2(px, f[0]));
}

/* the lane's share of A for one limb: bytes 4 tt .. 4 tt + 3 = the limb of (c_b, c_g, c_r, 0) of row r, where this lane is row
 * i = lane % 16 of K block lane / 16: nonzero only for i / 4 == lane / 16; tt = (i % 4) / 2, r = i % 2 */
static __device__ __forceinline__ int limb_of(int c, int which)
{
	const int l0 = ((c + 128) & 255) - 128;
	const int c1 = (c - l0) >> 8;
	const int l1 = ((c1 + 128) & 255) - 128;
	const int l2 = (c1 - l1) >> 8;
	return which == 0 ? l0 : which == 1 ? l1 : l2;
}
static __device__ __forceinline__ long a_share(const Rows &k, int which)
{
	const int lane = threadIdx.x & 63, i = lane & 15, q = lane >> 4;
	if ((i >> 2) != q)
		return 0;
	const int tt = (i & 3) >> 1, r = i & 1;
	const int *f = r ? k.fv : k.fu;
	const u32 w = (u32)(limb_of(f[2], which) & 255) | (u32)(limb_of(f[1], which) & 255) << 8 | (u32)(limb_of(f[0], which) & 255) << 16;
	return tt ? (long)((unsigned long)w << 32) : (long)w;
}

/* (a << 8) + b, kept apart from the next one: the compiler would fold a pair of them into two shifts and a three-way add (three
 * instructions for two).  An EMPTY asm: no instruction the hazard recognizer cannot see may read an MFMA's result. */
static __device__ __forceinline__ u32 lshl8_add(u32 a, u32 b)
{
	u32 r = (a << 8) + b;
	asm volatile("" : "+v"(r));
	return r;
}

struct MfmaK0 {
	long a0, a1, a2;
	v4i kk; /* the constant term + 128 * (sum of the row's coefficients), rows U, V, U, V */
};
static __device__ __forceinline__ MfmaK0 mfma_setup(const Rows &k)
{
	MfmaK0 m;
	m.a0 = a_share(k, 0);
	m.a1 = a_share(k, 1);
	m.a2 = a_share(k, 2);
	const int ku = k.fu[3] + 128 * (k.fu[0] + k.fu[1] + k.fu[2]), kv = k.fv[3] + 128 * (k.fv[0] + k.fv[1] + k.fv[2]);
	m.kk = (v4i){ku, kv, ku, kv};
	return m;
}
/* two texels in, E_u0, E_v0, E_u1, E_v1 out */
static __device__ __forceinline__ v4i k0_mfma(u32 p0, u32 p1, const MfmaK0 &m)
{
	const long b = (long)((unsigned long)(p0 ^ 0x80808080u) | (unsigned long)(p1 ^ 0x80808080u) << 32);
	const v4i z = {0, 0, 0, 0};
	const v4i d0 = __builtin_amdgcn_mfma_i32_16x16x32_i8(m.a0, b, m.kk, 0, 0, 0);
	const v4i d1 = __builtin_amdgcn_mfma_i32_16x16x32_i8(m.a1, b, z, 0, 0, 0);
	const v4i d2 = __builtin_amdgcn_mfma_i32_16x16x32_i8(m.a2, b, z, 0, 0, 0);
	v4i e;
#pragma unroll
	for (int j = 0; j < 4; j++)
		e[j] = (int)lshl8_add(lshl8_add((u32)d2[j], (u32)d1[j]), (u32)d0[j]);
	return e;
}
static __device__ __forceinline__ v4i k0_valu(u32 p0, u32 p1, const Rows &k)
{
	const int b0 = (int)(p0 & 255u), b1 = (int)(p1 & 255u);
	v4i e;
	e[0] = (int)row_valu(p0, b0, k.fu);
	e[1] = (int)row_valu(p0, b0, k.fv);
	e[2] = (int)row_valu(p1, b1, k.fu);
	e[3] = (int)row_valu(p1, b1, k.fv);
	return e;
}

__global__ void check_kernel(const u32 *px, u32 n_pairs, Rows k, u32 *bad, u32 *first_bad)
{
	const MfmaK0 m = mfma_setup(k);
	const u32 stride = gridDim.x * blockDim.x;
	const u32 rounds = (n_pairs + stride - 1) / stride;
	for (u32 r = 0; r < rounds; r++) { /* every lane stays in the loop: the MFMA is a wave's instruction */
		const u32 i = r * stride + blockIdx.x * blockDim.x + threadIdx.x;
		const bool ok = i < n_pairs;
		const u32 p0 = ok ? px[2 * i] : 0, p1 = ok ? px[2 * i + 1] : 0;
		const v4i a = k0_valu(p0, p1, k), b = k0_mfma(p0, p1, m);
		if (ok && (a[0] != b[0] || a[1] != b[1] || a[2] != b[2] || a[3] != b[3])) {
			if (atomicAdd(bad, 1u) == 0) {
				first_bad[0] = p0, first_bad[1] = p1;
				for (int j = 0; j < 4; j++)
					first_bad[2 + j] = (u32)a[j], first_bad[6 + j] = (u32)b[j];
			}
		}
	}
}

/* register-only loop: FORM 0 VALU rows, 1 MFMA rows; EXTRA independent VALU instructions per slot */
template <int FORM, int EXTRA> __global__ void __launch_bounds__(1024) time_kernel(u32 iters, Rows k, u32 *out)
{
	Rows kp = k;
	MfmaK0 m = mfma_setup(k);
	u32 p0 = threadIdx.x * 0x9E3779B9u + blockIdx.x, p1 = p0 * 0x85EBCA6Bu + 1u;
	u32 acc = 0, side[8];
#pragma unroll
	for (int j = 0; j < 8; j++)
		side[j] = p0 + j;
	for (u32 it = 0; it < iters; it++) {
		const v4i e = FORM ? k0_mfma(p0, p1, m) : k0_valu(p0, p1, kp);
		/* what the kernels do with the sums: U and V of the pair picked up two at a time, the fractions' minimum */
		const u32 pu = __builtin_amdgcn_perm((u32)e[2], (u32)e[0], 0x0c070c03u), pv = __builtin_amdgcn_perm((u32)e[3], (u32)e[1], 0x0c070c03u);
		acc ^= pu + (pv << 3);
#pragma unroll
		for (int j = 0; j < EXTRA; j++)
			side[j & 7] = __builtin_amdgcn_alignbit(side[j & 7], side[j & 7], 7 + j); /* one full-rate VALU instruction each, independent chains */
		p0 = __builtin_amdgcn_alignbit(p0, p0, 5) ^ acc;
		p1 = __builtin_amdgcn_alignbit(p1, p0, 11);
	}
#pragma unroll
	for (int j = 0; j < 8; j++)
		acc ^= side[j];
	out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

static void fixed_row(const float c[3], double offset, int out[4])
{
	for (int i = 0; i < 3; i++)
		out[i] = (int)lrint((double)c[i] * (double)(1 << 24));
	out[3] = (int)(uint32_t)(llrint((255.0 * offset + 0.5) * (double)(1 << 24)) + 512);
}

template <int FORM, int EXTRA> static double run(u32 iters, const Rows &k, u32 *out, int num_cu, double *sum)
{
	hipEvent_t ev[2];
	CHECK(hipEventCreate(&ev[0]));
	CHECK(hipEventCreate(&ev[1]));
	time_kernel<FORM, EXTRA><<<num_cu, 1024>>>(iters / 8, k, out);
	CHECK(hipDeviceSynchronize());
	float best = 1e30f;
	for (int rep = 0; rep < 3; rep++) {
		CHECK(hipEventRecord(ev[0]));
		time_kernel<FORM, EXTRA><<<num_cu, 1024>>>(iters, k, out);
		CHECK(hipEventRecord(ev[1]));
		CHECK(hipEventSynchronize(ev[1]));
		float ms;
		CHECK(hipEventElapsedTime(&ms, ev[0], ev[1]));
		best = ms < best ? ms : best;
	}
	std::vector<u32> h(1024);
	CHECK(hipMemcpy(h.data(), out, 4096, hipMemcpyDeviceToHost));
	double s = 0;
	for (u32 v : h)
		s += v;
	*sum = s;
	/* 4 waves per SIMD, each `iters` slots: ns per slot per SIMD */
	return (double)best * 1e6 / ((double)iters * 4.0);
}

int main()
{
	hipDeviceProp_t prop;
	CHECK(hipGetDeviceProperties(&prop, 0));
	const int num_cu = prop.multiProcessorCount;
	const double ghz = prop.clockRate * 1e-6;
	printf("%s, %d CUs, %.2f GHz nominal\n", prop.name, num_cu, ghz);
	Rows k[2];
	{
		const float u601[3] = {-0.147643f, -0.289855f, +0.437500f}, v601[3] = {+0.437500f, -0.366351f, -0.071147f};
		const float u709[3] = {-0.100643f, -0.338571f, +0.439216f}, v709[3] = {+0.439216f, -0.398941f, -0.040273f};
		fixed_row(u601, 0.5 - 0.00390625, k[0].fu), fixed_row(v601, 0.5, k[0].fv);
		fixed_row(u709, 0.5 - 0.00390625, k[1].fu), fixed_row(v709, 0.5, k[1].fv);
	}
	/* (i) exactness: all 2^24 colours as the first texel, a scrambled colour as the second, alpha random */
	const u32 n_pairs = 1u << 24;
	std::vector<u32> h(2 * (size_t)n_pairs);
	u32 s = 12345u;
	for (u32 i = 0; i < n_pairs; i++) {
		s = s * 1664525u + 1013904223u;
		h[2 * i] = i | (s & 0xff000000u);
		h[2 * i + 1] = (i * 0x9E3779B1u) ^ s;
	}
	u32 *d_px, *d_bad, *d_first, *d_out;
	CHECK(hipMalloc(&d_px, h.size() * 4));
	CHECK(hipMalloc(&d_bad, 4));
	CHECK(hipMalloc(&d_first, 40));
	CHECK(hipMalloc(&d_out, (size_t)num_cu * 1024 * 4));
	CHECK(hipMemcpy(d_px, h.data(), h.size() * 4, hipMemcpyHostToDevice));
	for (int m = 0; m < 2; m++) {
		CHECK(hipMemset(d_bad, 0, 4));
		check_kernel<<<num_cu * 4, 256>>>(d_px, n_pairs, k[m], d_bad, d_first);
		CHECK(hipDeviceSynchronize());
		u32 bad, fb[10];
		CHECK(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));
		CHECK(hipMemcpy(fb, d_first, 40, hipMemcpyDeviceToHost));
		printf("matrix %s: %u of %u texel pairs differ between the VALU rows and the MFMA rows\n", m ? "709" : "601", bad, n_pairs);
		if (bad)
			printf("  first: px %08x %08x valu %08x %08x %08x %08x mfma %08x %08x %08x %08x\n", fb[0], fb[1], fb[2], fb[3], fb[4], fb[5], fb[6],
			       fb[7], fb[8], fb[9]);
	}
	/* (ii) time */
	const u32 iters = 1u << 16;
	double sum;
	printf("\nns per slot (two texels) per SIMD at 4 waves per SIMD; cycles at the nominal clock in brackets\n");
	printf("%-28s %12s %12s\n", "extra VALU per slot", "VALU rows", "MFMA rows");
#define LINE(E)                                                                                                                   \
	do {                                                                                                                      \
		const double a = run<0, E>(iters, k[1], d_out, num_cu, &sum), b = run<1, E>(iters, k[1], d_out, num_cu, &sum);      \
		printf("%-28d %7.2f (%5.1f) %7.2f (%5.1f)   %+.1f %%\n", E, a, a * ghz, b, b * ghz, (a / b - 1) * 100);             \
	} while (0)
	LINE(0);
	LINE(8);
	LINE(16);
	LINE(24);
	LINE(32);
	return 0;
}
