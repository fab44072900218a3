// strip_read_bench.hip -- what does HBM give a kernel that reads frames in COLUMN STRIPS (the waveform's and the three-scope
// kernel's access pattern) instead of row after row?  64 frames of 3840x2160 BGRA8, one 1024-thread workgroup per CU, workgroup b
// = slice b % 4 of frame b / 4 walking strips slice, slice + 4, ... (scope_trio_kernel's interleaved schedule); a wave instruction
// reads 512 bytes as 512 / S row segments of S bytes whose rows lie H / (512 / S) apart, each lane 8 bytes, non-temporal, D loads
// per lane in flight; the values are XOR-ed away.  S = 128 is the three-scope kernel's 32-column strip, 256 the waveform's
// 64-column strip.  Measurement aid only (visit e5: 128 B 6.3 TB/s, 256 B and wider 6.6 TB/s, 64 B 3.2 TB/s).
//   hipcc --offload-arch=gfx950 -O3 -o tools/strip_read_bench tools/strip_read_bench.hip && ./tools/strip_read_bench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef unsigned long long u64;
typedef uint32_t u32;

template <int D>
__global__ __launch_bounds__(1024) void strip_read(const uint8_t *base, size_t frame_bytes, u32 linesize, u32 H, u32 strips, u32 slices,
						  u32 S, u64 *sink)
{
	const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
	const u32 LR = S / 8u < 64u ? S / 8u : 64u; /* lanes per row segment */
	const u32 RW = 64u / LR;                   /* row segments per wave instruction */
	const u32 chunks = S / (LR * 8u);           /* 512-byte pieces of a segment wider than 512 bytes */
	const u32 pair = lane % LR, rsub = lane / LR;
	const u32 rps = (H + RW - 1u) / RW;
	const u32 frame = blockIdx.x / slices, slice = blockIdx.x % slices;
	const uint8_t *plane = base + (size_t)frame * frame_bytes;
	u64 acc = 0;
	for (u32 s = slice; s < strips; s += slices)
		for (u32 c = 0; c < chunks; c++) {
			const u32 x0 = s * S + c * LR * 8u + pair * 8u;
			for (u32 g = wave; g < rps; g += nwaves * D) {
				u64 v[D];
#pragma unroll
				for (int d = 0; d < D; d++) {
					const u32 i = g + (u32)d * nwaves, row = rsub * rps + i;
					const bool ok = i < rps && row < H;
					v[d] = __builtin_nontemporal_load((const u64 *)(plane + (size_t)(ok ? row : 0u) * linesize + x0));
				}
#pragma unroll
				for (int d = 0; d < D; d++)
					acc ^= v[d];
			}
		}
	if (acc == 0x123456789abcdefull)
		sink[0] = acc;
}

int main()
{
	const u32 W = 3840, H = 2160, N = 64, linesize = W * 4;
	const size_t frame_bytes = (size_t)linesize * H;
	uint8_t *buf;
	u64 *sink;
	hipMalloc(&buf, frame_bytes * N);
	hipMalloc(&sink, 8);
	hipMemset(buf, 0x5a, frame_bytes * N);
	hipEvent_t a, b;
	hipEventCreate(&a);
	hipEventCreate(&b);
	const u32 segs[] = {64, 128, 256, 512, 1024};
	for (u32 S : segs) {
		const u32 strips = linesize / S, slices = 4;
		for (int D : {4, 8}) {
			auto launch = [&]() {
				if (D == 4)
					hipLaunchKernelGGL(strip_read<4>, dim3(N * slices), dim3(1024), 0, 0, buf, frame_bytes, linesize, H, strips, slices, S, sink);
				else
					hipLaunchKernelGGL(strip_read<8>, dim3(N * slices), dim3(1024), 0, 0, buf, frame_bytes, linesize, H, strips, slices, S, sink);
			};
			for (int i = 0; i < 5; i++)
				launch();
			hipDeviceSynchronize();
			hipEventRecord(a, 0);
			const int reps = 20;
			for (int i = 0; i < reps; i++)
				launch();
			hipEventRecord(b, 0);
			hipEventSynchronize(b);
			float ms;
			hipEventElapsedTime(&ms, a, b);
			printf("segment %5u bytes (%2u row segments per wave instruction), %d loads per lane in flight: %7.1f GB/s = %.3f of 8 TB/s\n", S,
			       S >= 512 ? 1u : 512u / S, D, (double)frame_bytes * N * reps / (ms * 1e-3) / 1e9, (double)frame_bytes * N * reps / (ms * 1e-3) / 8e12);
			fflush(stdout);
		}
	}
	return 0;
}
