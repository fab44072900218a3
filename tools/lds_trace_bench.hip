// Replays recorded LDS-atomic address traces on gfx950 (tuning aid; not part of the product).
// A trace is u32[n_instr][64]: the LDS word each lane of a wave adds to in one ds_add_u32.
// Every wave of every workgroup replays the same trace (waves start at different instructions),
// which prices a bin->address hash / lane mapping on real pixel data without writing a kernel.
//   ./lds_trace_bench trace.bin [n_instr]      prints clk per wave-instruction
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

extern __shared__ __attribute__((aligned(16))) uint32_t lds[];

template <int N> __global__ __launch_bounds__(1024) void replay(const uint32_t *trace, int reps, uint32_t *sink)
{
	for (uint32_t i = threadIdx.x; i < 32768; i += blockDim.x)
		lds[i] = 0;
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	uint32_t a[N];
#pragma unroll
	for (int i = 0; i < N; i++)
		a[i] = trace[(size_t)((i + wave * 3) % N) * 64 + lane] & 32767u;
	__syncthreads();
	for (int r = 0; r < reps; r++) {
#pragma unroll
		for (int i = 0; i < N; i++)
			atomicAdd(&lds[a[i]], 1u);
	}
	__syncthreads();
	uint32_t s = 0;
	for (uint32_t i = threadIdx.x; i < 32768; i += blockDim.x)
		s += lds[i];
	if (s == 0xdeadbeefu)
		sink[0] = s;
}

int main(int argc, char **argv)
{
	if (argc < 2)
		return 1;
	constexpr int N = 32;
	FILE *f = fopen(argv[1], "rb");
	std::vector<uint32_t> h(N * 64);
	uint32_t *d, *sink;
	hipMalloc(&d, N * 64 * 4);
	hipMalloc(&sink, 4);
	hipFuncSetAttribute((const void *)replay<N>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
	hipDeviceProp_t p;
	hipGetDeviceProperties(&p, 0);
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	const int reps = 128;
	char name[128];
	// file: repeated records of [64-byte label][N*64 u32]
	while (fread(name, 1, 64, f) == 64 && fread(h.data(), 4, N * 64, f) == (size_t)N * 64) {
		hipMemcpy(d, h.data(), N * 64 * 4, hipMemcpyHostToDevice);
		float best = 1e9f;
		for (int it = 0; it < 8; it++) {
			hipEventRecord(e0, 0);
			hipLaunchKernelGGL(replay<N>, dim3(p.multiProcessorCount), dim3(1024), 131072, 0, d, reps, sink);
			hipEventRecord(e1, 0);
			hipEventSynchronize(e1);
			float ms;
			hipEventElapsedTime(&ms, e0, e1);
			if (it >= 3 && ms < best)
				best = ms;
		}
		const double clk = best * 1e-3 * (p.clockRate * 1e3);
		printf("%-56s %.3f ms  %6.1f clk/wave-instr\n", name, best, clk / (16.0 * reps * N));
	}
	return 0;
}
