// ds_pk_add_f16 against ds_add_u32 on gfx950 (tuning aid; not part of the product library): would 16-bit FLOAT counters do
// for the vectorscope's LDS table?  A half counts exactly to 2048 and then stops growing (x + 1 rounds back to x): it saturates by
// itself, far above the 255 the reference saturates at (src/vectorscope.c:233-234), and never carries into its neighbour --
// no clamp passes.  This prints (1) what the hardware makes of N adds of 1.0 / 2.0 to one half, (2) the rate of both atomics.
//   hipcc --offload-arch=gfx950 -O3 tools/lds_pkf16_bench.hip -o tools/lds_pkf16_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) h2 lds_h2;
typedef __attribute__((address_space(3))) unsigned lds_u;
extern __shared__ __attribute__((aligned(16))) uint32_t lds[];

__device__ __forceinline__ void add_f16(unsigned byte, unsigned bits)
{
	__builtin_amdgcn_ds_atomic_fadd_v2f16((lds_h2 *)(__UINTPTR_TYPE__)byte, __builtin_bit_cast(h2, bits));
}
__device__ __forceinline__ void add_u32(unsigned byte, unsigned n)
{
	__hip_atomic_fetch_add((lds_u *)(__UINTPTR_TYPE__)byte, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// n adds of `bits` to word 0, spread over the 1024 threads (thread t makes the adds t, t + 1024, ...): out[0] = the word
__global__ __launch_bounds__(1024) void probe(unsigned n, unsigned bits, uint32_t *out)
{
	if (threadIdx.x < 64)
		lds[threadIdx.x] = 0;
	__syncthreads();
	for (unsigned i = threadIdx.x; i < n; i += 1024)
		add_f16(0, bits);
	__syncthreads();
	if (threadIdx.x == 0) {
		out[0] = lds[0];
		out[1] = lds[1];
	}
}

template <bool F16> __global__ __launch_bounds__(1024) void bench(int mode, uint32_t nbins, int iters, uint32_t *sink)
{
	for (uint32_t i = threadIdx.x; i < 32768; i += blockDim.x)
		lds[i] = 0;
	__syncthreads();
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	uint32_t rng = 0x9E3779B9u * (blockIdx.x * 1024u + threadIdx.x + 1u);
	if (mode == 0) { // conflict-free: lane -> own bank, own address
		const uint32_t base = lane + 64u * wave;
		for (int it = 0; it < iters; it++) {
			const uint32_t w = (base + 1024u * (uint32_t)(it & 7)) & 32767u;
			if (F16)
				add_f16(w * 4u, 0x3C00u);
			else
				add_u32(w * 4u, 1u);
		}
	} else { // random word, one of its two halves
		for (int it = 0; it < iters; it++) {
			rng = rng * 1664525u + 1013904223u;
			const uint32_t r = rng >> 8, w = r % nbins, hi = (r >> 23) & 1u;
			if (F16)
				add_f16(w * 4u, 0x3C00u << (16u * hi));
			else
				add_u32(w * 4u, 1u << (16u * hi));
		}
	}
	__syncthreads();
	uint32_t s = 0;
	for (uint32_t i = threadIdx.x; i < 32768; i += blockDim.x)
		s += lds[i];
	if (s == 0xdeadbeefu)
		sink[0] = s;
}

static float half_bits_to_float(unsigned h)
{
	const unsigned e = (h >> 10) & 31u, m = h & 1023u;
	if (e == 0)
		return (float)m / 16777216.0f;
	if (e == 31)
		return m ? __builtin_nanf("") : __builtin_inff();
	float v = 1.0f + (float)m / 1024.0f;
	for (unsigned i = 15; i < e; i++)
		v *= 2.0f;
	for (unsigned i = e; i < 15; i++)
		v *= 0.5f;
	return v;
}

int main()
{
	hipDeviceProp_t p;
	hipGetDeviceProperties(&p, 0);
	const int cus = p.multiProcessorCount;
	const double ghz = p.clockRate / 1e6;
	printf("# %s %s CUs=%d clock=%.2f GHz\n", p.name, p.gcnArchName, cus, ghz);
	uint32_t *out, h[2];
	hipMalloc(&out, 8);
	printf("# N adds to ONE half of one word (1024 threads): the half afterwards, and the rest of the LDS word / its neighbour\n");
	for (unsigned bits : {0x3C00u, 0x4000u, 0x3C000000u, 0x3C003C00u})
		for (unsigned n : {1u, 2u, 255u, 256u, 300u, 2047u, 2048u, 2049u, 4000u, 5000u, 70000u, 1000000u}) {
			hipLaunchKernelGGL(probe, dim3(1), dim3(1024), 1024, 0, n, bits, out);
			hipMemcpy(h, out, 8, hipMemcpyDeviceToHost);
			printf("  add 0x%08x x %7u -> word 0x%08x = (lo %g, hi %g), next word 0x%08x\n", bits, n, h[0], half_bits_to_float(h[0] & 0xffffu),
			       half_bits_to_float(h[0] >> 16), h[1]);
		}
	hipFuncSetAttribute((const void *)bench<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
	hipFuncSetAttribute((const void *)bench<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	const int iters = 4096;
	auto run = [&](bool f16, int mode, uint32_t nbins, const char *label) {
		float best = 1e9f;
		for (int rep = 0; rep < 6; rep++) {
			hipEventRecord(e0, 0);
			if (f16)
				hipLaunchKernelGGL(bench<true>, dim3(cus), dim3(1024), 131072, 0, mode, nbins, iters, out);
			else
				hipLaunchKernelGGL(bench<false>, dim3(cus), dim3(1024), 131072, 0, mode, nbins, iters, out);
			hipEventRecord(e1, 0);
			hipEventSynchronize(e1);
			float ms;
			hipEventElapsedTime(&ms, e0, e1);
			if (rep >= 2 && ms < best)
				best = ms;
		}
		const double clk = best * 1e-3 * ghz * 1e9;
		printf("%-14s %-28s nbins=%6u  %.3f ms  %.2f lanes/clk/CU  %.1f clk/wave-instr\n", f16 ? "ds_pk_add_f16" : "ds_add_u32", label, nbins, best,
		       1024.0 * iters / clk, clk / (16.0 * iters));
	};
	for (int i = 0; i < 20; i++)
		hipLaunchKernelGGL(bench<false>, dim3(cus), dim3(1024), 131072, 0, 0, 1u, iters, out);
	hipDeviceSynchronize();
	for (int f = 0; f < 2; f++) {
		run(f, 0, 0, "conflict-free");
		for (uint32_t nb : {64u, 1024u, 26100u, 32768u})
			run(f, 1, nb, "random word, one of 2 halves");
	}
	return 0;
}
