// LDS atomic throughput microbenchmark for gfx950 (tuning aid; not part of the product library).
// Measures ds_add_u32 (no return) lanes/clk/CU under controlled conflict patterns so that the
// vectorscope's bin->LDS-address hash and its wave-level merging can be priced instead of guessed.
//   hipcc --offload-arch=gfx950 -O3 tools/lds_atomic_bench.hip -o tools/lds_atomic_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

extern __shared__ __attribute__((aligned(16))) uint32_t lds[];

// mode 0: conflict-free (lane -> own bank, own address)
// mode 1: k lanes of each 32-lane group share ONE address (same-address multiplicity k)
// mode 2: k lanes of each 32-lane group share one BANK with different addresses
// mode 3: uniformly random word in [0, nbins)
// mode 4: like 3 but two 16-bit halves per word (inc = 1 or 65536), nbins words
__global__ __launch_bounds__(1024) void bench(int mode, int k, uint32_t nbins, int iters, uint32_t *sink)
{
	for (uint32_t i = threadIdx.x; i < 32768; i += blockDim.x)
		lds[i] = 0;
	__syncthreads();
	const uint32_t lane = threadIdx.x & 63u, l32 = lane & 31u, half = lane >> 5, wave = threadIdx.x >> 6;
	uint32_t base, inc = 1;
	if (mode == 0)
		base = l32 + 32u * (2u * wave + half);
	else if (mode == 1)
		base = (l32 / (uint32_t)k) + 32u * (2u * wave + half);
	else if (mode == 2)
		base = (l32 % (32u / (uint32_t)k)) + 32u * ((l32 / (32u / (uint32_t)k)) + 8u * (2u * wave + half));
	else
		base = 0;
	uint32_t rng = 0x9E3779B9u * (blockIdx.x * 1024u + threadIdx.x + 1u);
	if (mode < 3) {
		for (int it = 0; it < iters; it++) {
			const uint32_t w = (base + 1024u * (uint32_t)(it & 7)) & 32767u;
			atomicAdd(&lds[w], inc);
		}
	} else {
		for (int it = 0; it < iters; it++) {
			rng = rng * 1664525u + 1013904223u;
			const uint32_t r = rng >> 8;
			const uint32_t w = r % nbins;
			if (mode == 4)
				inc = (r & 0x800000u) ? 65536u : 1u;
			atomicAdd(&lds[w], inc);
		}
	}
	__syncthreads();
	uint32_t s = 0;
	for (uint32_t i = threadIdx.x; i < 32768; i += blockDim.x)
		s += lds[i];
	if (s == 0xdeadbeefu)
		sink[0] = s;
}

// same loop without the atomic: cost of the address generation alone
__global__ __launch_bounds__(1024) void bench_noatomic(uint32_t nbins, int iters, uint32_t *sink)
{
	uint32_t rng = 0x9E3779B9u * (blockIdx.x * 1024u + threadIdx.x + 1u), acc = 0;
	for (int it = 0; it < iters; it++) {
		rng = rng * 1664525u + 1013904223u;
		acc += (rng >> 8) % nbins;
	}
	if (acc == 0xdeadbeefu)
		sink[0] = acc;
}

int main()
{
	hipDeviceProp_t p;
	hipGetDeviceProperties(&p, 0);
	const int cus = p.multiProcessorCount;
	const double ghz = p.clockRate / 1e6;
	printf("# %s %s CUs=%d clock=%.2f GHz\n", p.name, p.gcnArchName, cus, ghz);
	hipFuncSetAttribute((const void *)bench, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
	uint32_t *sink;
	hipMalloc(&sink, 4);
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	const int iters = 4096;
	auto run = [&](int mode, int k, uint32_t nbins, const char *label) {
		float best = 1e9f;
		for (int rep = 0; rep < 6; rep++) {
			hipEventRecord(e0, 0);
			hipLaunchKernelGGL(bench, dim3(cus), dim3(1024), 131072, 0, mode, k, nbins, iters, sink);
			hipEventRecord(e1, 0);
			hipEventSynchronize(e1);
			float ms;
			hipEventElapsedTime(&ms, e0, e1);
			if (rep >= 2 && ms < best)
				best = ms;
		}
		const double lanes = 1024.0 * iters;  // per CU
		const double clk = best * 1e-3 * ghz * 1e9;
		printf("%-34s k=%2d nbins=%6u  %.3f ms  %.2f lanes/clk/CU  %.1f clk/wave-instr\n", label, k, nbins, best,
		       lanes / clk, clk / (16.0 * iters));
	};
	// warm the clocks
	for (int i = 0; i < 20; i++)
		hipLaunchKernelGGL(bench, dim3(cus), dim3(1024), 131072, 0, 0, 1, 1u, iters, sink);
	hipDeviceSynchronize();
	run(0, 1, 0, "conflict-free");
	for (int k : {2, 4, 8, 16, 32})
		run(1, k, 0, "same-address x k");
	for (int k : {2, 4, 8, 16, 32})
		run(2, k, 0, "same-bank, distinct address x k");
	for (uint32_t nb : {1u, 2u, 4u, 8u, 16u, 32u, 64u, 256u, 1024u, 8192u, 32768u})
		run(3, 0, nb, "random word in nbins");
	for (uint32_t nb : {8u, 64u, 1024u, 32768u})
		run(4, 0, nb, "random word, 2 halves");
	{
		float best = 1e9f;
		for (int rep = 0; rep < 5; rep++) {
			hipEventRecord(e0, 0);
			hipLaunchKernelGGL(bench_noatomic, dim3(cus), dim3(1024), 0, 0, 32768u, iters, sink);
			hipEventRecord(e1, 0);
			hipEventSynchronize(e1);
			float ms;
			hipEventElapsedTime(&ms, e0, e1);
			if (rep >= 1 && ms < best)
				best = ms;
		}
		printf("address generation only: %.3f ms\n", best);
	}
	return 0;
}
