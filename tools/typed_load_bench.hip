// buffer_load_format_xyzw (the texture path's format conversion: a BGRA8 texel arrives as four zero-extended dwords, its bytes already
// apart) against raw buffer loads of the same bytes, streaming 2 GiB on gfx950.  Tuning aid, not part of the product library: would
// the fused kernels get their channel bytes for free?  (K0's rows would be three multiply-adds each instead of four instructions and
// the blue byte's mask.)
//   hipcc --offload-arch=gfx950 -O3 tools/typed_load_bench.hip -o tools/typed_load_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef int i4 __attribute__((ext_vector_type(4)));
typedef int i2 __attribute__((ext_vector_type(2)));
__device__ i4 llvm_buf_load_format_v4i32(i4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.format.v4i32");
__device__ i4 llvm_buf_load_v4i32(i4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4i32");
__device__ i2 llvm_buf_load_v2i32(i4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v2i32");
__device__ int llvm_buf_load_i32(i4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.i32");

__device__ __forceinline__ i4 make_rsrc(const void *base, unsigned bytes, unsigned word3)
{
	const unsigned long long a = (unsigned long long)base;
	i4 r;
	r.x = (int)__builtin_amdgcn_readfirstlane((unsigned)a);
	r.y = (int)(__builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) & 0xffffu);
	r.z = (int)__builtin_amdgcn_readfirstlane(bytes);
	r.w = (int)word3;
	return r;
}

// MODE 0: typed 8_8_8_8 UINT, 4 B per lane per load; 1: raw dword (4 B); 2: raw dwordx2 (8 B); 3: raw dwordx4 (16 B).
// Every workgroup streams its own contiguous 8 MiB-ish share in chunks of (1024 lanes x bytes per load), U loads in flight.
template <int MODE, int U> __global__ __launch_bounds__(1024) void stream(const uint8_t *src, unsigned long long bytes, uint32_t *sink)
{
	constexpr unsigned B = MODE == 0 ? 4u : MODE == 1 ? 4u : MODE == 2 ? 8u : 16u;
	const unsigned long long share = bytes / gridDim.x;
	const uint8_t *base = src + share * blockIdx.x;
	const i4 rs = make_rsrc(base, (unsigned)share, MODE == 0 ? 0x00054FACu : 0x00020000u);
	unsigned off = threadIdx.x * B;
	const unsigned step = 1024u * B;
	unsigned acc = 0;
	for (; off + (U - 1) * step < (unsigned)share; off += U * step) {
#pragma unroll
		for (int k = 0; k < U; k++) {
			if (MODE == 0) {
				const i4 v = llvm_buf_load_format_v4i32(rs, (int)(off + k * step), 0, 2);
				acc += (unsigned)v.x + 3u * (unsigned)v.y + 5u * (unsigned)v.z + 7u * (unsigned)v.w;
			} else if (MODE == 1) {
				acc += (unsigned)llvm_buf_load_i32(rs, (int)(off + k * step), 0, 2);
			} else if (MODE == 2) {
				const i2 v = llvm_buf_load_v2i32(rs, (int)(off + k * step), 0, 2);
				acc += (unsigned)v.x + 3u * (unsigned)v.y;
			} else {
				const i4 v = llvm_buf_load_v4i32(rs, (int)(off + k * step), 0, 2);
				acc += (unsigned)v.x + 3u * (unsigned)v.y + 5u * (unsigned)v.z + 7u * (unsigned)v.w;
			}
		}
	}
	if (acc == 0x12345u)
		sink[0] = acc;
	if (MODE == 0 && blockIdx.x == 0 && threadIdx.x == 0) { // what the conversion hands out: texel 0 of the share
		const i4 v = llvm_buf_load_format_v4i32(rs, 0, 0, 0);
		sink[4] = (unsigned)v.x;
		sink[5] = (unsigned)v.y;
		sink[6] = (unsigned)v.z;
		sink[7] = (unsigned)v.w;
	}
}

__global__ void fill(uint32_t *p, unsigned long long n)
{
	for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x)
		p[i] = (uint32_t)(i * 2654435761u) ^ 0x04030201u * (i == 0);
}

int main()
{
	hipDeviceProp_t p;
	(void)hipGetDeviceProperties(&p, 0);
	const int cus = p.multiProcessorCount;
	const unsigned long long bytes = 2ull << 30;
	uint8_t *src;
	uint32_t *sink, h[8];
	(void)hipMalloc(&src, bytes);
	(void)hipMalloc(&sink, 64);
	hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, (uint32_t *)src, bytes / 4);
	(void)hipDeviceSynchronize();
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0);
	(void)hipEventCreate(&e1);
	printf("# %s CUs=%d; 2 GiB streamed by %d workgroups of 1024, nt loads\n", p.gcnArchName, cus, cus);
#define RUN(MODE, U, label)                                                                                          \
	do {                                                                                                          \
		float best = 1e9f;                                                                                    \
		for (int rep = 0; rep < 5; rep++) {                                                                   \
			(void)hipEventRecord(e0, 0);                                                                        \
			hipLaunchKernelGGL((stream<MODE, U>), dim3(cus), dim3(1024), 0, 0, src, bytes, sink);         \
			(void)hipEventRecord(e1, 0);                                                                        \
			(void)hipEventSynchronize(e1);                                                                      \
			float ms;                                                                                     \
			(void)hipEventElapsedTime(&ms, e0, e1);                                                             \
			if (rep >= 1 && ms < best)                                                                    \
				best = ms;                                                                            \
		}                                                                                                     \
		printf("%-44s loads in flight %2d  %8.3f ms  %7.1f GB/s\n", label, U, best, bytes / (best * 1e-3) / 1e9); \
	} while (0)
	RUN(3, 4, "raw buffer_load_dwordx4 (16 B per lane)");
	RUN(2, 6, "raw buffer_load_dwordx2 (8 B per lane)");
	RUN(2, 12, "raw buffer_load_dwordx2 (8 B per lane)");
	RUN(1, 8, "raw buffer_load_dword (4 B per lane)");
	RUN(1, 16, "raw buffer_load_dword (4 B per lane)");
	RUN(0, 4, "buffer_load_format_xyzw 8_8_8_8 UINT (4 B)");
	RUN(0, 8, "buffer_load_format_xyzw 8_8_8_8 UINT (4 B)");
	RUN(0, 12, "buffer_load_format_xyzw 8_8_8_8 UINT (4 B)");
	(void)hipMemcpy(h, sink, 32, hipMemcpyDeviceToHost);
	printf("# texel 0 = bytes 01 02 03 04 in memory -> x %u y %u z %u w %u\n", h[4], h[5], h[6], h[7]);
	return 0;
}
