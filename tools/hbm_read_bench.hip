// What does a plain streaming read reach on this chip?  The yardstick for the scope kernels' roofline fractions:
// every variant reads the same 1 GiB (32 x 4K BGRA8 frames' worth) of HBM-resident data with 16-byte loads and
// reduces it to one word per lane (XOR), so nothing but the loads can be the bound.
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/hbm_read_bench tools/hbm_read_bench.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef uint32_t v4 __attribute__((vector_size(16)));

template <int U, bool NT> __global__ __launch_bounds__(1024) void read_kernel(const v4 *src, uint64_t n16, uint32_t *sink)
{
	// workgroup g streams its contiguous share; lanes interleave at 16 B (a wave covers 1 KiB per load)
	const uint64_t per = (n16 + gridDim.x - 1) / gridDim.x;
	const uint64_t b0 = (uint64_t)blockIdx.x * per, b1 = (b0 + per < n16) ? b0 + per : n16;
	v4 acc = {0, 0, 0, 0};
	uint64_t i = b0 + threadIdx.x;
	for (; i + (uint64_t)(U - 1) * blockDim.x < b1; i += (uint64_t)U * blockDim.x) {
		v4 t[U];
#pragma unroll
		for (int k = 0; k < U; k++)
			t[k] = NT ? __builtin_nontemporal_load(src + i + (uint64_t)k * blockDim.x) : src[i + (uint64_t)k * blockDim.x];
#pragma unroll
		for (int k = 0; k < U; k++)
			acc ^= t[k];
	}
	for (; i < b1; i += blockDim.x)
		acc ^= src[i];
	const uint32_t r = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
	if (r == 0x12345678u)
		sink[0] = r; // never true for the test pattern; keeps the loads alive
}

// Persistent workgroups, chunk-interleaved: workgroup g reads chunks g, g+G, g+2G, ... of blockDim*U*16 bytes, so the
// chip as a whole sweeps memory linearly (what a huge grid of small workgroups does) and every lane keeps two batches
// of U loads in flight (ping-pong), as the scope kernels do.  STREAMS > 1 splits a wave into lane groups that read
// STREAMS far-apart sub-ranges of the workgroup's data instead (the vectorscope's lane groups).
template <int U, bool NT, int MODE> __global__ __launch_bounds__(1024) void read_pp(const v4 *src, uint64_t n16, uint32_t *sink)
{
	v4 acc = {0, 0, 0, 0};
	const uint64_t chunk = (uint64_t)blockDim.x * U;
	const uint64_t nchunks = n16 / chunk; // the tail is ignored (< 0.01 %)
	v4 a[U], b[U];
	auto idx = [&](uint64_t c, int k) -> uint64_t {
		if (MODE == 0) // interleaved chunks
			return (c * gridDim.x + blockIdx.x) * chunk + (uint64_t)k * blockDim.x + threadIdx.x;
		// contiguous share per workgroup (MODE 1), optionally 8 lane-group streams inside it (MODE 2)
		const uint64_t per = nchunks / gridDim.x;
		if (MODE == 1)
			return ((uint64_t)blockIdx.x * per + c) * chunk + (uint64_t)k * blockDim.x + threadIdx.x;
		const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, grp = lane >> 3, sub = lane & 7u;
		const uint64_t stream_cells = per * chunk / 8; // cells per lane-group stream
		const uint64_t base = (uint64_t)blockIdx.x * per * chunk + grp * stream_cells;
		// within a stream: waves interleave 8 lanes (128 B) each
		return base + (c * U + k) * (uint64_t)(blockDim.x >> 3) + wave * 8u + sub;
	};
	const uint64_t mine = (MODE == 0) ? (nchunks - blockIdx.x + gridDim.x - 1) / gridDim.x : nchunks / gridDim.x;
	auto issue = [&](v4(&t)[U], uint64_t c) {
		const uint64_t cc = c < mine ? c : 0;
#pragma unroll
		for (int k = 0; k < U; k++)
			t[k] = NT ? __builtin_nontemporal_load(src + idx(cc, k)) : src[idx(cc, k)];
	};
	auto use = [&](v4(&t)[U]) {
#pragma unroll
		for (int k = 0; k < U; k++)
			acc ^= t[k];
	};
	issue(a, 0);
	for (uint64_t c = 0; c < mine; c += 2) {
		issue(b, c + 1);
		use(a);
		issue(a, c + 2);
		use(b);
	}
	const uint32_t r = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
	if (r == 0x12345678u)
		sink[0] = r;
}

template <int U, bool NT, int MODE> static float run_pp(const v4 *src, uint64_t n16, uint32_t *sink, int grid, int block, int reps)
{
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	for (int i = 0; i < 3; i++)
		hipLaunchKernelGGL((read_pp<U, NT, MODE>), dim3(grid), dim3(block), 0, 0, src, n16, sink);
	hipEventRecord(e0, 0);
	for (int i = 0; i < reps; i++)
		hipLaunchKernelGGL((read_pp<U, NT, MODE>), dim3(grid), dim3(block), 0, 0, src, n16, sink);
	hipEventRecord(e1, 0);
	hipEventSynchronize(e1);
	float ms = 0;
	hipEventElapsedTime(&ms, e0, e1);
	hipEventDestroy(e0);
	hipEventDestroy(e1);
	return ms / reps;
}

// Wave specialisation: the first LW waves of a 1024-thread workgroup only load (two chunks of U loads per lane in
// flight, ping-pong) and hand the data over through a double-buffered LDS ring; the other waves consume it (here:
// XOR).  One barrier per chunk.  Is this a way to combine the shallow, small-workgroup load pattern the HBM likes with
// the 12+ waves of compute the scope kernels need?
template <int U, int LW> __global__ __launch_bounds__(1024) void read_split(const v4 *src, uint64_t n16, uint32_t *sink)
{
	extern __shared__ v4 ring[]; // 2 x CH cells
	constexpr uint32_t NL = LW * 64u, CH = NL * U;
	const bool loader = threadIdx.x < NL;
	const uint64_t nchunks = n16 / CH;
	const uint64_t mine = (nchunks - blockIdx.x + gridDim.x - 1) / gridDim.x;
	v4 acc = {0, 0, 0, 0};
	v4 a[U], b[U];
	auto issue = [&](v4(&t)[U], uint64_t c) {
		const uint64_t cc = c < mine ? c : 0;
		const v4 *p = src + (cc * gridDim.x + blockIdx.x) * CH + threadIdx.x;
#pragma unroll
		for (int k = 0; k < U; k++)
			t[k] = __builtin_nontemporal_load(p + (uint64_t)k * NL);
	};
	auto hand_over = [&](v4(&t)[U], uint64_t c) {
		v4 *w = ring + (c & 1) * CH + threadIdx.x;
#pragma unroll
		for (int k = 0; k < U; k++)
			w[k * NL] = t[k];
	};
	auto consume = [&](uint64_t c) {
		const v4 *r = ring + (c & 1) * CH;
		for (uint32_t i = threadIdx.x - NL; i < CH; i += blockDim.x - NL)
			acc ^= r[i];
	};
	if (loader) {
		issue(a, 0);
		issue(b, 1);
	}
	for (uint64_t c = 0; c < mine; c += 2) {
		if (loader) {
			hand_over(a, c);
			issue(a, c + 2);
		}
		__syncthreads();
		if (!loader)
			consume(c);
		if (c + 1 < mine) {
			if (loader) {
				hand_over(b, c + 1);
				issue(b, c + 3);
			}
			__syncthreads();
			if (!loader)
				consume(c + 1);
		}
	}
	const uint32_t r = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
	if (r == 0x12345678u)
		sink[0] = r;
}

template <int U, int LW> static float run_split(const v4 *src, uint64_t n16, uint32_t *sink, int grid, int reps)
{
	const size_t lds = 2u * LW * 64u * U * 16u;
	hipFuncSetAttribute((const void *)read_split<U, LW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	for (int i = 0; i < 3; i++)
		hipLaunchKernelGGL((read_split<U, LW>), dim3(grid), dim3(1024), lds, 0, src, n16, sink);
	hipEventRecord(e0, 0);
	for (int i = 0; i < reps; i++)
		hipLaunchKernelGGL((read_split<U, LW>), dim3(grid), dim3(1024), lds, 0, src, n16, sink);
	hipEventRecord(e1, 0);
	hipEventSynchronize(e1);
	float ms = 0;
	hipEventElapsedTime(&ms, e0, e1);
	hipEventDestroy(e0);
	hipEventDestroy(e1);
	return ms / reps;
}

template <int U, bool NT> static float run(const v4 *src, uint64_t n16, uint32_t *sink, int grid, int block, int reps)
{
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	for (int i = 0; i < 3; i++)
		hipLaunchKernelGGL((read_kernel<U, NT>), dim3(grid), dim3(block), 0, 0, src, n16, sink);
	hipEventRecord(e0, 0);
	for (int i = 0; i < reps; i++)
		hipLaunchKernelGGL((read_kernel<U, NT>), dim3(grid), dim3(block), 0, 0, src, n16, sink);
	hipEventRecord(e1, 0);
	hipEventSynchronize(e1);
	float ms = 0;
	hipEventElapsedTime(&ms, e0, e1);
	hipEventDestroy(e0);
	hipEventDestroy(e1);
	return ms / reps;
}

__global__ void fill_kernel(uint32_t *p, uint64_t n)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
		p[i] = (uint32_t)(i * 2654435761u) | 1u;
}

// Round 2: why does the same kernel read 3-4 % slower from some allocations ("two placement modes")?
// Ten 1 GiB allocations of one process, each read (a) as chunks interleaved over the workgroups -- the chip sweeps memory
// linearly and touches few pages at a time -- and (b) as one contiguous share per workgroup walked in 8 lane-group
// streams, which is what the vectorscope kernel does: 2048 streams that are 0.5 MB apart.  If only (b) depends on the
// allocation, the address translation (page fragment size of the allocation) is the suspect, not the DRAM.
static int placement_main()
{
	const uint64_t bytes = 32ull * 3840 * 2160 * 4;
	const uint64_t n16 = bytes / 16;
	const int NB = 10;
	uint32_t *buf[NB], *sink;
	hipMalloc((void **)&sink, 64);
	for (int i = 0; i < NB; i++) {
		if (hipMalloc((void **)&buf[i], bytes) != hipSuccess) {
			fprintf(stderr, "hipMalloc failed\n");
			return 1;
		}
		hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, buf[i], bytes / 4);
	}
	hipDeviceSynchronize();
	for (int i = 0; i < 300; i++)
		hipLaunchKernelGGL((read_kernel<4, true>), dim3(256), dim3(1024), 0, 0, (const v4 *)buf[0], n16, sink);
	hipDeviceSynchronize();
	for (int round = 0; round < 2; round++)
		for (int i = 0; i < NB; i++) {
			const float a = run_pp<4, true, 0>((const v4 *)buf[i], n16, sink, 256, 1024, 30);
			const float b = run_pp<4, true, 2>((const v4 *)buf[i], n16, sink, 256, 1024, 30);
			const float c = run_pp<4, true, 1>((const v4 *)buf[i], n16, sink, 256, 1024, 30);
			printf("round %d buffer %d at %p (mod 2MiB = %#llx, mod 1GiB = %#llx): interleaved %7.1f  contiguous+8 streams %7.1f  contiguous %7.1f GB/s\n",
			       round, i, (void *)buf[i], (unsigned long long)((uintptr_t)buf[i] & 0x1fffff),
			       (unsigned long long)((uintptr_t)buf[i] & 0x3fffffff), bytes / (a * 1e-3) / 1e9, bytes / (b * 1e-3) / 1e9,
			       bytes / (c * 1e-3) / 1e9);
			fflush(stdout);
		}
	return 0;
}

// Round 2, second look: is it the position in the allocation sequence (= where in VRAM the driver put the buffer) that
// decides?  (1) 24 allocations in a row; (2) free the first twelve and allocate twelve new ones -- do the new ones take
// over the freed (fast) region and its speed?  (3) everything freed, ONE allocation of 16 x that size read window by
// window.  Interleaved reading only (placement_main showed that the access order does not matter).
static float placement_read(const uint32_t *p, uint64_t n16, uint32_t *sink)
{
	return run_pp<4, true, 0>((const v4 *)p, n16, sink, 256, 1024, 20);
}

static int placement2_main()
{
	const uint64_t bytes = 32ull * 3840 * 2160 * 4;
	const uint64_t n16 = bytes / 16;
	const int NB = 24;
	uint32_t *buf[NB], *sink;
	size_t fr = 0, tot = 0;
	hipMalloc((void **)&sink, 64);
	for (int i = 0; i < NB; i++) {
		if (hipMalloc((void **)&buf[i], bytes) != hipSuccess)
			return 1;
		hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, buf[i], bytes / 4);
	}
	hipDeviceSynchronize();
	for (int i = 0; i < 300; i++)
		hipLaunchKernelGGL((read_kernel<4, true>), dim3(256), dim3(1024), 0, 0, (const v4 *)buf[0], n16, sink);
	hipDeviceSynchronize();
	hipMemGetInfo(&fr, &tot);
	printf("# (1) %d allocations of %.2f GB in a row (free now %.1f of %.1f GB)\n", NB, bytes / 1e9, fr / 1e9, tot / 1e9);
	for (int i = 0; i < NB; i++) {
		printf("alloc %2d at %p  %7.1f GB/s\n", i, (void *)buf[i], bytes / (placement_read(buf[i], n16, sink) * 1e-3) / 1e9);
		fflush(stdout);
	}
	printf("# (2) the first twelve freed, twelve new ones allocated\n");
	for (int i = 0; i < 12; i++)
		hipFree(buf[i]);
	for (int i = 0; i < 12; i++) {
		if (hipMalloc((void **)&buf[i], bytes) != hipSuccess)
			return 1;
		hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, buf[i], bytes / 4);
	}
	hipDeviceSynchronize();
	for (int i = 0; i < 12; i++) {
		printf("new   %2d at %p  %7.1f GB/s\n", i, (void *)buf[i], bytes / (placement_read(buf[i], n16, sink) * 1e-3) / 1e9);
		fflush(stdout);
	}
	for (int i = 12; i < NB; i += 4)
		printf("old   %2d at %p  %7.1f GB/s\n", i, (void *)buf[i], bytes / (placement_read(buf[i], n16, sink) * 1e-3) / 1e9);
	for (int i = 0; i < NB; i++)
		hipFree(buf[i]);
	printf("# (3) everything freed; one allocation of 16 x %.2f GB, read window by window\n", bytes / 1e9);
	uint32_t *big;
	if (hipMalloc((void **)&big, bytes * 16) != hipSuccess)
		return 1;
	hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, big, bytes * 16 / 4);
	hipDeviceSynchronize();
	for (int i = 0; i < 16; i++) {
		const uint32_t *w = big + (uint64_t)i * (bytes / 4);
		printf("window %2d at %p  %7.1f GB/s\n", i, (const void *)w, bytes / (placement_read(w, n16, sink) * 1e-3) / 1e9);
		fflush(stdout);
	}
	printf("# (4) the same process, a second allocation of one buffer size\n");
	uint32_t *late;
	if (hipMalloc((void **)&late, bytes) != hipSuccess)
		return 1;
	hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, late, bytes / 4);
	hipDeviceSynchronize();
	printf("late      at %p  %7.1f GB/s\n", (void *)late, bytes / (placement_read(late, n16, sink) * 1e-3) / 1e9);
	return 0;
}

// Round 2: what does ONE 4K frame (33 MB) cost when nothing but the read is done?  The yardstick for the one- and
// three-frame launches of the scopes (tools/latency_probe.py): launches back to back over a pool of 24 distinct frames.
template <typename K> static float run_small(K launch, int reps)
{
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	for (int i = 0; i < 20; i++)
		launch(i);
	hipEventRecord(e0, 0);
	for (int i = 0; i < reps; i++)
		launch(i);
	hipEventRecord(e1, 0);
	hipEventSynchronize(e1);
	float ms = 0;
	hipEventElapsedTime(&ms, e0, e1);
	hipEventDestroy(e0);
	hipEventDestroy(e1);
	return ms / reps * 1e3f;
}

static int small_main()
{
	const uint64_t frame = 3840ull * 2160 * 4, pool = 24;
	uint32_t *buf, *sink;
	if (hipMalloc((void **)&buf, frame * pool) != hipSuccess || hipMalloc((void **)&sink, 64) != hipSuccess)
		return 1;
	hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, buf, frame * pool / 4);
	for (int i = 0; i < 300; i++)
		hipLaunchKernelGGL((read_kernel<4, true>), dim3(256), dim3(1024), 0, 0, (const v4 *)buf, frame * pool / 16, sink);
	hipDeviceSynchronize();
	for (int nf : {1, 3, 6}) {
		const uint64_t n16 = frame * nf / 16;
		const int groups = (int)(pool / nf);
		auto at = [&](int i) { return (const v4 *)((const uint8_t *)buf + (uint64_t)(i % groups) * nf * frame); };
		const int reps = 600;
		printf("%d frame(s) per launch, us per launch (back to back):\n", nf);
#define SMALL(label, expr) printf("  %-58s %7.2f us  %7.1f GB/s\n", label, t = run_small([&](int i) { expr; }, reps), frame * nf / (t * 1e-6) / 1e9)
		float t;
		SMALL("256 x 1024 threads, contiguous share, 4 loads/lane", hipLaunchKernelGGL((read_kernel<4, true>), dim3(256), dim3(1024), 0, 0, at(i), n16, sink));
		SMALL("256 x 1024 threads, contiguous share, 2 loads/lane", hipLaunchKernelGGL((read_kernel<2, true>), dim3(256), dim3(1024), 0, 0, at(i), n16, sink));
		SMALL("256 x 1024, ping-pong 4+4, interleaved chunks", hipLaunchKernelGGL((read_pp<4, true, 0>), dim3(256), dim3(1024), 0, 0, at(i), n16, sink));
		SMALL("256 x 1024, ping-pong 4+4, contiguous share", hipLaunchKernelGGL((read_pp<4, true, 1>), dim3(256), dim3(1024), 0, 0, at(i), n16, sink));
		SMALL("256 x 1024, ping-pong 2+2, contiguous share", hipLaunchKernelGGL((read_pp<2, true, 1>), dim3(256), dim3(1024), 0, 0, at(i), n16, sink));
		SMALL("256 x 1024, ping-pong 4+4, contiguous + 8 lane-group streams", hipLaunchKernelGGL((read_pp<4, true, 2>), dim3(256), dim3(1024), 0, 0, at(i), n16, sink));
		SMALL("512 x 512, ping-pong 4+4, interleaved chunks", hipLaunchKernelGGL((read_pp<4, true, 0>), dim3(512), dim3(512), 0, 0, at(i), n16, sink));
		SMALL("1024 x 256, ping-pong 4+4, interleaved chunks", hipLaunchKernelGGL((read_pp<4, true, 0>), dim3(1024), dim3(256), 0, 0, at(i), n16, sink));
		SMALL("2048 x 256 threads, contiguous share, 4 loads/lane", hipLaunchKernelGGL((read_kernel<4, true>), dim3(2048), dim3(256), 0, 0, at(i), n16, sink));
		SMALL("8192 x 256 threads, contiguous share, 4 loads/lane", hipLaunchKernelGGL((read_kernel<4, true>), dim3(8192), dim3(256), 0, 0, at(i), n16, sink));
		SMALL("empty kernel, 256 x 1024", hipLaunchKernelGGL((read_kernel<4, true>), dim3(256), dim3(1024), 0, 0, at(i), (uint64_t)0, sink));
	}
	return 0;
}

// "mix": what does the memory system deliver when a stream of reads carries a share of writes?  The waveform moves 33.2 MB
// of frame in and 3.9 MB of table out per 4K frame (10.6 % of its bytes are writes); this is the bare form of that mix:
// every workgroup streams its contiguous share with 16-byte nt loads, four per lane in flight, and after every K-th batch
// stores one batch worth of 16-byte nt words (the running XOR: real data) to a contiguous share of a second buffer.
// SPREAD: one store per lane after every `every / U` batches instead of U stores after every `every` batches (same share).
template <int U, bool SPREAD> __global__ __launch_bounds__(1024) void mix_kernel(const v4 *src, uint64_t n16, v4 *dst, uint32_t every)
{
	const uint32_t spread_every = every / U ? every / U : 1;
	const uint64_t per = (n16 + gridDim.x - 1) / gridDim.x;
	const uint64_t b0 = (uint64_t)blockIdx.x * per, b1 = (b0 + per < n16) ? b0 + per : n16;
	v4 acc = {0, 0, 0, 0};
	uint64_t i = b0 + threadIdx.x, o = b0 / (every ? every : 1) + threadIdx.x;
	uint32_t phase = 0;
	for (; i + (uint64_t)(U - 1) * blockDim.x < b1; i += (uint64_t)U * blockDim.x) {
		v4 t[U];
#pragma unroll
		for (int k = 0; k < U; k++)
			t[k] = __builtin_nontemporal_load(src + i + (uint64_t)k * blockDim.x);
#pragma unroll
		for (int k = 0; k < U; k++)
			acc ^= t[k];
		if (every && ++phase == (SPREAD ? spread_every : every)) { /* uniform */
			phase = 0;
#pragma unroll
			for (int k = 0; k < (SPREAD ? 1 : U); k++) {
				acc[0] += k;
				__builtin_nontemporal_store(acc, dst + o + (uint64_t)k * blockDim.x);
			}
			o += (uint64_t)(SPREAD ? 1 : U) * blockDim.x;
		}
	}
	if (!every && (acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u)
		dst[0] = acc;
}

static int mix_main()
{
	const uint64_t bytes = 64ull * 3840 * 2160 * 4;
	const uint64_t n16 = bytes / 16;
	uint32_t *buf;
	v4 *dst;
	if (hipMalloc((void **)&buf, bytes) != hipSuccess || hipMalloc((void **)&dst, bytes + (1u << 20)) != hipSuccess) {
		fprintf(stderr, "hipMalloc failed\n");
		return 1;
	}
	hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, buf, bytes / 4);
	hipDeviceSynchronize();
	for (int i = 0; i < 100; i++)
		hipLaunchKernelGGL((mix_kernel<4, false>), dim3(256), dim3(1024), 0, 0, (const v4 *)buf, n16, dst, 0u);
	hipDeviceSynchronize();
	printf("# 64 x 4K frames (%llu bytes) read per launch, 256 workgroups x 1024 threads, four 16-byte nt loads per lane in flight;\n"
	       "# after every K-th batch one batch of 16-byte nt stores.  GB/s over 30 back-to-back launches (HIP events).\n",
	       (unsigned long long)bytes);
	printf("# %8s %10s %12s %12s %12s\n", "K", "writes", "ms", "read GB/s", "total GB/s");
	const uint32_t ks[] = {0, 64, 32, 16, 9, 8, 4, 2, 1};
	for (int rep = 0; rep < 3; rep++)
		for (uint32_t k : ks) {
			const bool spread = rep == 2;
			if (spread && (k % 4 || !k))
				continue;
			hipEvent_t e0, e1;
			hipEventCreate(&e0);
			hipEventCreate(&e1);
			hipEventRecord(e0, 0);
			for (int i = 0; i < 30; i++)
				if (spread)
					hipLaunchKernelGGL((mix_kernel<4, true>), dim3(256), dim3(1024), 0, 0, (const v4 *)buf, n16, dst, k);
				else
					hipLaunchKernelGGL((mix_kernel<4, false>), dim3(256), dim3(1024), 0, 0, (const v4 *)buf, n16, dst, k);
			hipEventRecord(e1, 0);
			hipEventSynchronize(e1);
			float ms;
			hipEventElapsedTime(&ms, e0, e1);
			ms /= 30;
			const double w = k ? (double)bytes / k : 0.0;
			printf("  %8u %9.1f%% %12.4f %12.1f %12.1f%s\n", k, 100.0 * w / (bytes + w), ms, bytes / (ms * 1e-3) / 1e9,
			       (bytes + w) / (ms * 1e-3) / 1e9, spread ? "  (one store per lane every K/4 batches)" : "");
			hipEventDestroy(e0);
			hipEventDestroy(e1);
		}
	hipFree(buf);
	hipFree(dst);
	return 0;
}

// "vmm": does memory mapped through the virtual-memory API (hipMemCreate in chunks of the allocation granularity, one
// contiguous virtual range) read at another rate than what hipMalloc hands out?  Six buffers of 64 x 4K frames each way,
// alternately, each read by the persistent two-batch stream (the scope kernels' shape).
static int vmm_main()
{
	setvbuf(stdout, NULL, _IONBF, 0);
	const uint64_t bytes = 64ull * 3840 * 2160 * 4;
	const uint64_t n16 = bytes / 16;
	uint32_t *sink;
	hipMalloc((void **)&sink, 64);
	hipMemAllocationProp prop = {};
	prop.type = hipMemAllocationTypePinned;
	prop.location.type = hipMemLocationTypeDevice;
	prop.location.id = 0;
	size_t gran = 0, gran_rec = 0;
	hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum);
	hipMemGetAllocationGranularity(&gran_rec, &prop, hipMemAllocationGranularityRecommended);
	printf("# allocation granularity: minimum %zu, recommended %zu bytes\n", gran, gran_rec);
	hipMemAccessDesc acc = {};
	acc.location = prop.location;
	acc.flags = hipMemAccessFlagsProtReadWrite;
	for (int round = 0; round < 3; round++) {
		uint32_t *m;
		if (hipMalloc((void **)&m, bytes) != hipSuccess) {
			fprintf(stderr, "hipMalloc failed\n");
			return 1;
		}
		hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, m, bytes / 4);
		hipDeviceSynchronize();
		const float t_malloc = placement_read(m, n16, sink);
		for (int chunked = 0; chunked < 1; chunked++) { /* (granularity is 4 KiB here: mapping half a million chunks takes minutes) */
			const size_t chunk = chunked ? (gran_rec ? gran_rec : gran) : ((bytes + gran - 1) / gran) * gran;
			const size_t total = ((bytes + chunk - 1) / chunk) * chunk;
			void *va = NULL;
			printf("  reserving %zu bytes, %zu chunk(s) of %zu ...\n", total, total / chunk, chunk);
			if (hipMemAddressReserve(&va, total, 0, NULL, 0) != hipSuccess) {
				fprintf(stderr, "hipMemAddressReserve failed\n");
				return 1;
			}
			const size_t nchunks = total / chunk;
			hipMemGenericAllocationHandle_t *h = (hipMemGenericAllocationHandle_t *)calloc(nchunks, sizeof(*h));
			bool ok = true;
			for (size_t i = 0; i < nchunks && ok; i++)
				ok = hipMemCreate(&h[i], chunk, &prop, 0) == hipSuccess && hipMemMap((char *)va + i * chunk, chunk, 0, h[i], 0) == hipSuccess;
			printf("  mapped: %d\n", (int)ok);
			ok = ok && hipMemSetAccess(va, total, &acc, 1) == hipSuccess;
			printf("  access set: %d\n", (int)ok);
			if (!ok) {
				fprintf(stderr, "virtual-memory API failed (%s)\n", hipGetErrorString(hipGetLastError()));
				return 1;
			}
			hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, (uint32_t *)va, bytes / 4);
			hipDeviceSynchronize();
			const float t = placement_read((const uint32_t *)va, n16, sink);
			printf("round %d  hipMalloc at %p %7.1f GB/s   hipMemCreate x %zu chunk(s) of %zu MiB at %p %7.1f GB/s\n", round, (void *)m,
			       bytes / (t_malloc * 1e-3) / 1e9, nchunks, chunk >> 20, va, bytes / (t * 1e-3) / 1e9);
			hipMemUnmap(va, total);
			for (size_t i = 0; i < nchunks; i++)
				hipMemRelease(h[i]);
			hipMemAddressFree(va, total);
			free(h);
		}
		if (round & 1)
			hipFree(m); /* every other hipMalloc buffer stays: later rounds see a fuller device */
	}
	return 0;
}

int main(int argc, char **argv)
{
	if (argc > 1 && !strcmp(argv[1], "placement"))
		return placement_main();
	if (argc > 1 && !strcmp(argv[1], "placement2"))
		return placement2_main();
	if (argc > 1 && !strcmp(argv[1], "small"))
		return small_main();
	if (argc > 1 && !strcmp(argv[1], "mix"))
		return mix_main();
	if (argc > 1 && !strcmp(argv[1], "vmm"))
		return vmm_main();
	const uint64_t bytes = 32ull * 3840 * 2160 * 4;
	const uint64_t n16 = bytes / 16;
	uint32_t *buf, *sink;
	if (hipMalloc((void **)&buf, bytes) != hipSuccess || hipMalloc((void **)&sink, 64) != hipSuccess) {
		fprintf(stderr, "hipMalloc failed\n");
		return 1;
	}
	hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, buf, bytes / 4);
	hipDeviceSynchronize();
	// clocks up
	for (int i = 0; i < 200; i++)
		hipLaunchKernelGGL((read_kernel<4, true>), dim3(256), dim3(1024), 0, 0, (const v4 *)buf, n16, sink);
	hipDeviceSynchronize();
	printf("# %llu bytes per pass; GB/s = bytes / time of one kernel (HIP events over 50 back-to-back launches)\n",
	       (unsigned long long)bytes);
	const int grids[] = {256, 512, 1024, 2048, 8192, 32768};
	const int blocks[] = {1024, 512, 256};
	for (int nt = 0; nt < 2; nt++)
		for (int u = 0; u < 3; u++)
			for (int b : blocks)
				for (int g : grids) {
					float ms;
					if (nt)
						ms = u == 0 ? run<2, true>((const v4 *)buf, n16, sink, g, b, 50)
						   : u == 1 ? run<4, true>((const v4 *)buf, n16, sink, g, b, 50)
							    : run<8, true>((const v4 *)buf, n16, sink, g, b, 50);
					else
						ms = u == 0 ? run<2, false>((const v4 *)buf, n16, sink, g, b, 50)
						   : u == 1 ? run<4, false>((const v4 *)buf, n16, sink, g, b, 50)
							    : run<8, false>((const v4 *)buf, n16, sink, g, b, 50);
					printf("nt=%d unroll=%d block=%4d grid=%6d  %8.4f ms  %8.1f GB/s\n", nt, u == 0 ? 2 : u == 1 ? 4 : 8, b, g,
					       ms, bytes / (ms * 1e-3) / 1e9);
				}
	printf("# persistent workgroups with two batches in flight: mode 0 chunk-interleaved, 1 contiguous share, 2 contiguous share\n"
	       "# walked as 8 lane-group streams\n");
#define PP(U, MODE)                                                                                                     \
	for (int b : {1024, 512, 256})                                                                                  \
		for (int g : {256, 512}) {                                                                              \
			const float ms = run_pp<U, true, MODE>((const v4 *)buf, n16, sink, g, b, 50);                   \
			printf("pp nt=1 unroll=%d mode=%d block=%4d grid=%6d  %8.4f ms  %8.1f GB/s\n", U, MODE, b, g, ms,      \
			       bytes / (ms * 1e-3) / 1e9);                                                              \
		}
	PP(1, 0) PP(2, 0) PP(4, 0) PP(1, 1) PP(2, 1) PP(4, 1) PP(1, 2) PP(2, 2) PP(4, 2)
	printf("# wave-specialised: LW loader waves hand chunks to the other waves of a 1024-thread workgroup through LDS\n");
#define SP(U, LW)                                                                                                       \
	{                                                                                                               \
		const float ms = run_split<U, LW>((const v4 *)buf, n16, sink, 256, 50);                                 \
		printf("split unroll=%d loader_waves=%d ring=%d KiB  %8.4f ms  %8.1f GB/s\n", U, LW, 2 * LW * 64 * U * 16 / 1024, ms, \
		       bytes / (ms * 1e-3) / 1e9);                                                                      \
	}
	SP(1, 4) SP(2, 4) SP(4, 4) SP(1, 8) SP(2, 8) SP(4, 2) SP(8, 2) SP(2, 2) SP(1, 2) SP(1, 6) SP(2, 6)
	hipFree(buf);
	hipFree(sink);
	return 0;
}
