// How should the 256 workgroups of a ONE-frame vectorscope launch combine their LDS tables?  (round 2)
//
// Every workgroup holds a 128 KiB table (32768 words, two 16-bit counters per word) and has to get it into ONE
// 256x256 u8 table with min(sum, 255).  Round 1 wrote 256 clamped u8 partials (16.8 MB) and ran a second kernel
// over them: 2.0x the algorithmic traffic and a second launch.  Candidates measured here, all with 256 workgroups
// of 1024 threads and a synthetic table of a given density of non-zero counters:
//   mode 0  nothing (kernel + LDS fill: the baseline that is subtracted)
//   mode 1  clamped u8 partial, plain 16-byte stores                      (round 1)
//   mode 2  the same, write-through (sc1) stores
//   mode 3  device-scope atomic add of every non-zero word into ONE table  (32-bit: two counters per atomic)
//   mode 4  the same with 64-bit atomics (four counters per atomic)
//   mode 5  workgroup-scope ("XCD-local": no sc1, performed in the XCD's own L2) atomic add into table[XCC_ID], 32 bit
//   mode 6  the same, 64 bit
//   mode 7  mode 3 + arrival ticket + the last workgroup clamps the table to u8, writes it and re-zeroes the accumulator
//   mode 8  mode 4 + the same tail
//   mode 9  mode 4 + tail that reads-and-zeroes with returning 64-bit atomic exchanges
// Counter values are 1 (the u16 halves cannot carry).  Modes 3..9 are verified against the host's sum.
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/flush_bench tools/flush_bench.hip
//   run:   tools/flush_bench            (the whole table)
//          tools/flush_bench MODE DENSITY_PERMILLE REPS   (one configuration; for rocprofv3 --pmc passes)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef uint32_t u32;
typedef uint64_t u64;
#define WORDS 32768u
#define NXCD 8

extern __shared__ __attribute__((aligned(16))) u32 lds[];

__host__ __device__ static inline u32 pattern_word(u32 wg, u32 w, u32 thresh)
{
	u32 x = (wg * WORDS + w) * 2654435761u;
	x ^= x >> 15;
	x *= 2246822519u;
	x ^= x >> 13;
	const u32 lo = (x & 0xffffu) < thresh ? 1u : 0u;
	const u32 hi = (x >> 16) < thresh ? 1u : 0u;
	return lo | (hi << 16);
}

__device__ __forceinline__ u32 xcc_id()
{
	// s_getreg_b32 hwreg(HW_REG_XCC_ID = 20), bits [3:0]
	return __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 0xfu;
}

__device__ __forceinline__ u32 clamp_pair(u32 w0, u32 w1)
{
	return min(w0 & 0xffffu, 255u) | (min(w0 >> 16, 255u) << 8) | (min(w1 & 0xffffu, 255u) << 16) | (min(w1 >> 16, 255u) << 24);
}

typedef u32 v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store16_sc1(void *p, uint4 v)
{
	const v4u x = {v.x, v.y, v.z, v.w};
	asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(x) : "memory");
}
__device__ __forceinline__ uint4 load16_sc1(const void *p)
{
	v4u v;
	asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
	return make_uint4(v.x, v.y, v.z, v.w);
}

struct Args {
	u32 *acc;       // mode 3,4,7,8,9: one table of WORDS words; mode 5,6: NXCD tables
	uint8_t *part;  // mode 1,2: gridDim.x partial tables of 65536 bytes
	uint8_t *out;   // mode 7..9: the final table
	u32 *ticket;
	u32 *xcc_of_wg;
	u32 thresh;
};

template <int MODE> __global__ __launch_bounds__(1024) void flush_kernel(Args a)
{
	const u32 t = threadIdx.x, wg = blockIdx.x;
	for (u32 w = t; w < WORDS; w += 1024)
		lds[w] = pattern_word(wg, w, a.thresh);
	__syncthreads();
	if (MODE == 0) {
		if (lds[t] == 0x12345678u)
			a.acc[0] = 1;
		return;
	}
	if (MODE == 1 || MODE == 2) {
		uint4 *out = (uint4 *)(a.part + (size_t)wg * 65536);
		for (u32 g = t; g < 65536 / 16; g += 1024) {
			const uint4 lo = *(const uint4 *)&lds[g * 8], hi = *(const uint4 *)&lds[g * 8 + 4];
			const uint4 v = make_uint4(clamp_pair(lo.x, lo.y), clamp_pair(lo.z, lo.w), clamp_pair(hi.x, hi.y), clamp_pair(hi.z, hi.w));
			if (MODE == 2)
				store16_sc1(out + g, v);
			else
				out[g] = v;
		}
		return;
	}
	if (MODE == 3 || MODE == 7) {
		for (u32 w = t; w < WORDS; w += 1024) {
			const u32 v = lds[w];
			if (v)
				__hip_atomic_fetch_add(&a.acc[w], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
	}
	if (MODE == 4 || MODE == 8 || MODE == 9) {
		for (u32 w = t; w < WORDS / 2; w += 1024) {
			const u64 v = ((const u64 *)lds)[w];
			if (v)
				__hip_atomic_fetch_add(&((u64 *)a.acc)[w], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
	}
	if (MODE == 5 || MODE == 6) {
		const u32 x = xcc_id();
		if (t == 0)
			a.xcc_of_wg[wg] = x;
		u32 *tab = a.acc + (size_t)x * WORDS;
		if (MODE == 5) {
			for (u32 w = t; w < WORDS; w += 1024) {
				const u32 v = lds[w];
				if (v)
					__hip_atomic_fetch_add(&tab[w], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			}
		} else {
			for (u32 w = t; w < WORDS / 2; w += 1024) {
				const u64 v = ((const u64 *)lds)[w];
				if (v)
					__hip_atomic_fetch_add(&((u64 *)tab)[w], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			}
		}
		return;
	}
	if (MODE >= 7) {
		// every wave waits for its atomics to be acknowledged, then one lane takes the ticket
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__syncthreads();
		if (t == 0)
			lds[0] = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u;
		__syncthreads();
		if (!lds[0])
			return;
		if (MODE == 9) {
			for (u32 g = t; g < WORDS / 2; g += 1024) {
				const u64 v = __hip_atomic_exchange(&((u64 *)a.acc)[g], (u64)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				((u32 *)a.out)[g] = clamp_pair((u32)v, (u32)(v >> 32));
			}
		} else {
			const uint4 zero = make_uint4(0, 0, 0, 0);
			for (u32 g = t; g < WORDS / 4; g += 1024) {
				const uint4 v = load16_sc1((const uint4 *)a.acc + g);
				((uint2 *)a.out)[g] = make_uint2(clamp_pair(v.x, v.y), clamp_pair(v.z, v.w));
				store16_sc1((uint4 *)a.acc + g, zero);
			}
		}
		if (t == 0)
			__hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
}

typedef void (*kern_t)(Args);
static kern_t kernels[10] = {flush_kernel<0>, flush_kernel<1>, flush_kernel<2>, flush_kernel<3>, flush_kernel<4>,
			     flush_kernel<5>, flush_kernel<6>, flush_kernel<7>, flush_kernel<8>, flush_kernel<9>};

#define CK(x)                                                                       \
	do {                                                                        \
		hipError_t e_ = (x);                                                \
		if (e_ != hipSuccess) {                                             \
			printf("%s: %s\n", #x, hipGetErrorString(e_));              \
			exit(1);                                                    \
		}                                                                   \
	} while (0)

static const int G = 256;

static void run(int mode, int permille, int reps, Args a, bool verify)
{
	const u32 thresh = (u32)(65536.0 * permille / 1000.0);
	a.thresh = thresh;
	CK(hipMemset(a.acc, 0, (size_t)NXCD * WORDS * 4));
	CK(hipMemset(a.ticket, 0, 4));
	CK(hipMemset(a.out, 0xEE, 65536));
	hipLaunchKernelGGL(kernels[mode], dim3(G), dim3(1024), WORDS * 4, 0, a);
	CK(hipDeviceSynchronize());
	const char *verdict = "";
	if (verify && mode >= 3) {
		std::vector<u32> want(WORDS, 0), got((size_t)NXCD * WORDS);
		for (u32 g = 0; g < (u32)G; g++)
			for (u32 w = 0; w < WORDS; w++)
				want[w] += pattern_word(g, w, thresh);
		bool ok = true;
		if (mode >= 7) {
			std::vector<uint8_t> out(65536);
			CK(hipMemcpy(out.data(), a.out, 65536, hipMemcpyDeviceToHost));
			CK(hipMemcpy(got.data(), a.acc, WORDS * 4, hipMemcpyDeviceToHost));
			for (u32 w = 0; w < WORDS && ok; w++) {
				const u32 lo = want[w] & 0xffffu, hi = want[w] >> 16;
				ok = out[2 * w] == (lo > 255 ? 255 : lo) && out[2 * w + 1] == (hi > 255 ? 255 : hi) && got[w] == 0;
			}
		} else {
			CK(hipMemcpy(got.data(), a.acc, (size_t)NXCD * WORDS * 4, hipMemcpyDeviceToHost));
			const int tabs = (mode == 5 || mode == 6) ? NXCD : 1;
			for (u32 w = 0; w < WORDS && ok; w++) {
				u32 s = 0;
				for (int x = 0; x < tabs; x++)
					s += got[(size_t)x * WORDS + w];
				ok = s == want[w];
			}
		}
		verdict = ok ? "sum ok" : "SUM WRONG";
		CK(hipMemset(a.acc, 0, (size_t)NXCD * WORDS * 4));
	}
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	for (int i = 0; i < 3; i++)
		hipLaunchKernelGGL(kernels[mode], dim3(G), dim3(1024), WORDS * 4, 0, a);
	CK(hipEventRecord(e0, 0));
	for (int i = 0; i < reps; i++)
		hipLaunchKernelGGL(kernels[mode], dim3(G), dim3(1024), WORDS * 4, 0, a);
	CK(hipEventRecord(e1, 0));
	CK(hipEventSynchronize(e1));
	float ms = 0;
	CK(hipEventElapsedTime(&ms, e0, e1));
	printf("mode %d density %4d/1000  %8.2f us per launch (back to back, incl. boundary)  %s\n", mode, permille,
	       ms / reps * 1e3, verdict);
	fflush(stdout);
	CK(hipEventDestroy(e0));
	CK(hipEventDestroy(e1));
}

int main(int argc, char **argv)
{
	Args a;
	CK(hipMalloc((void **)&a.acc, (size_t)NXCD * WORDS * 4));
	CK(hipMalloc((void **)&a.part, (size_t)G * 65536));
	CK(hipMalloc((void **)&a.out, 65536));
	CK(hipMalloc((void **)&a.ticket, 256));
	CK(hipMalloc((void **)&a.xcc_of_wg, G * 4));
	for (int m = 0; m < 10; m++)
		CK(hipFuncSetAttribute((const void *)kernels[m], hipFuncAttributeMaxDynamicSharedMemorySize, WORDS * 4));
	if (argc >= 4) {
		run(atoi(argv[1]), atoi(argv[2]), atoi(argv[3]), a, false);
		return 0;
	}
	// warm the clocks
	for (int i = 0; i < 2000; i++)
		hipLaunchKernelGGL(kernels[1], dim3(G), dim3(1024), WORDS * 4, 0, a);
	CK(hipDeviceSynchronize());
	const int dens[] = {1000, 500, 250, 100, 10};
	for (int m = 0; m < 10; m++)
		for (int d : dens) {
			if ((m <= 2) && d != 1000)
				continue;
			run(m, d, 200, a, true);
		}
	// which XCD did each workgroup run on (mode 5 records it)
	std::vector<u32> x(G);
	CK(hipMemcpy(x.data(), a.xcc_of_wg, G * 4, hipMemcpyDeviceToHost));
	int hist[16] = {0}, mod8 = 0;
	for (int g = 0; g < G; g++) {
		hist[x[g] & 15]++;
		mod8 += (int)x[g] == g % 8;
	}
	printf("workgroups per XCC_ID:");
	for (int i = 0; i < 8; i++)
		printf(" %d", hist[i]);
	printf("   (XCC_ID == blockIdx %% 8 for %d of %d)\n", mod8, G);
	return 0;
}
