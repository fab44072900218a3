/*
 * placement_vmm_probe.hip -- WHICH property of a frame pool's memory decides the 3-6 % a streaming kernel's rate moves by from pool
 * to pool on MI355X (profiles/HISTORY.md section 5 knows only that it is "the memory behind an allocation")?
 * A measurement aid, not product code: it drives the PRODUCT kernel (scope_vectorscope_batch of libscope_hip.so, 64 x 4K staged
 * frames per launch, timed by the kernel's own begin / end stamps) over pools whose virtual and physical layout it controls through
 * HIP's virtual-memory calls (hipMemCreate / hipMemAddressReserve / hipMemMap), and beside every rate it measures the pool's
 * ADDRESS TRANSLATION directly: the latency of dependent loads that step through the pool with a stride of 4 KiB, 64 KiB, 2 MiB and
 * 32 MiB (+ 64 bytes: never the same cache line twice) -- a stride inside the page-table fragment the driver gave the memory hits
 * the TLB, a stride beyond it walks the tables every time.
 *
 *   1  hipMalloc pools and single-handle hipMemCreate pools (mapped at 1 GiB-aligned addresses), all held at once, read in turn
 *   2  only the fastest and the slowest of them, alternating          -- does it matter what else is in the rotation?
 *   3  the fastest and the slowest HANDLE swap their addresses, everything read again -- physical memory or address?
 *   4  one handle mapped 2 MiB / 64 MiB off a 1 GiB boundary          -- the address's alignment (fragment size the tables can use)
 *   5  pools made of separately created chunks of 2 / 32 / 256 MiB mapped side by side -- physical contiguity
 *   6  ONE handle of many GiB, the pool at successive offsets inside it -- along one physical allocation
 *   7  one pool, the 64 frames laid out with another stride          -- the relative phases of the 256 concurrent streams
 *   8  section 1 once more                                           -- drift over the run
 *
 * Every rate is the median over R rounds; the rounds visit the candidates of a section in turn.
 *   build: make -C tools placement_vmm_probe      run (GPU box): ./tools/placement_vmm_probe [rounds] [slab GiB]
 */
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../include/scope_hip.h"

static scope_ctx *g_ctx;
#define CK(x)                                                                                              \
	do {                                                                                               \
		hipError_t e_ = (x);                                                                       \
		if (e_ != hipSuccess) {                                                                    \
			fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
			exit(2);                                                                           \
		}                                                                                          \
	} while (0)
#define SK(x)                                                                                                                          \
	do {                                                                                                                           \
		int s_ = (x);                                                                                                          \
		if (s_ != SCOPE_OK) {                                                                                                  \
			fprintf(stderr, "%s:%d %s -> %s (%s)\n", __FILE__, __LINE__, #x, scope_strerror(s_), scope_last_error(g_ctx)); \
			exit(3);                                                                                                       \
		}                                                                                                                      \
	} while (0)

static const uint32_t W = 3840, H = 2160, NF = 64;
static const size_t FRAME = (size_t)W * H * 4, POOL = FRAME * NF;
static uint8_t *g_src, *g_out; /* the 64 staged planes (master copy); the 64 tables */
static unsigned long long *g_ticks;
static int g_rounds = 5;

__global__ void noise_kernel(uint32_t *p, size_t n, uint32_t seed)
{
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		uint64_t x = (i + 1) * 0x9E3779B97F4A7C15ull + seed;
		x ^= x >> 29;
		x *= 0xBF58476D1CE4E5B9ull;
		x ^= x >> 32;
		p[i] = (uint32_t)x | 0xff000000u; /* B, G, R uniform; alpha 255 */
	}
}

/* n dependent loads of ONE lane, `stride` bytes apart (wrapping inside the pool, shifted by a few lines per wrap): every texel of a
 * staged plane has 255 in byte 3, which makes the next address depend on the value loaded.  100 MHz ticks in out[0..1]. */
__global__ void chase_kernel(const uint8_t *base, size_t pool, size_t stride, uint32_t n, unsigned long long *out)
{
	if (threadIdx.x != 0)
		return;
	size_t off = 0;
	uint32_t sink = 0;
	const unsigned long long t0 = wall_clock64();
	for (uint32_t i = 0; i < n; i++) {
		const uint32_t v = __builtin_nontemporal_load((const uint32_t *)(base + off));
		sink += v;
		off += stride + ((v >> 24) - 255u);
		if (off >= pool - 64)
			off = off - (pool - 64) + 4160; /* 4 KiB + 64: other pages' other lines on the next sweep */
		if (off >= pool - 64)
			off = 4160;
	}
	const unsigned long long t1 = wall_clock64();
	out[0] = t1 - t0;
	out[1] = sink;
}

static const size_t CHASE_STRIDES[4] = {(4u << 10) + 64, (64u << 10) + 64, (2u << 20) + 64, (32u << 20) + 64};
static double chase_ns(const uint8_t *base, size_t stride)
{
	const uint32_t n = 1500;
	chase_kernel<<<1, 64>>>(base, POOL, stride, 64, g_ticks); /* warm: code, the first tables */
	chase_kernel<<<1, 64>>>(base, POOL, stride, n, g_ticks);
	CK(hipDeviceSynchronize());
	unsigned long long t[2];
	CK(hipMemcpy(t, g_ticks, sizeof t, hipMemcpyDeviceToHost));
	return (double)t[0] * 10.0 / n;
}

/* one pool = a base address and a frame stride; rate in GB/s of algorithmic bytes (4 B per texel + the table) */
struct Cand {
	std::string name;
	uint8_t *base;
	size_t stride;
	std::vector<double> gbps;
	double lat[4];
};
static Cand cand(const std::string &name, uint8_t *base, size_t stride = FRAME)
{
	Cand c;
	c.name = name;
	c.base = base;
	c.stride = stride;
	for (double &l : c.lat)
		l = 0;
	return c;
}

static double measure(uint8_t *base, size_t stride)
{
	const uint8_t *planes[NF];
	for (uint32_t f = 0; f < NF; f++)
		planes[f] = base + f * stride;
	scope_batch b = {planes, NF, W * 4, W, H, 0, 2};
	const uint32_t flags = SCOPE_IN_DEVICE | SCOPE_OUT_DEVICE | SCOPE_ASYNC;
	for (int i = 0; i < 8; i++)
		SK(scope_vectorscope_batch(g_ctx, &b, flags, g_out));
	SK(scope_ctx_sync(g_ctx));
	SK(scope_ctx_profile(g_ctx, 1));
	for (int i = 0; i < 24; i++)
		SK(scope_vectorscope_batch(g_ctx, &b, flags, g_out));
	uint32_t n = 0;
	float ms = 0;
	SK(scope_ctx_kernel_ms(g_ctx, &n, &ms));
	SK(scope_ctx_profile(g_ctx, 0));
	return (double)NF * (FRAME + 65536) / (ms / n * 1e-3) / 1e9;
}

static void fill(uint8_t *base, size_t stride)
{
	if (stride == FRAME) {
		CK(hipMemcpy(base, g_src, POOL, hipMemcpyDeviceToDevice));
		return;
	}
	for (uint32_t f = 0; f < NF; f++)
		CK(hipMemcpyAsync(base + f * stride, g_src + f * FRAME, FRAME, hipMemcpyDeviceToDevice, 0));
	CK(hipDeviceSynchronize());
}

static double median(std::vector<double> v)
{
	std::sort(v.begin(), v.end());
	return v[v.size() / 2];
}

static void run_section(const char *title, std::vector<Cand> &cs, bool refill = false, bool filled = false)
{
	printf("\n== %s\n", title);
	if (!refill && !filled)
		for (auto &c : cs)
			if (c.base != g_src)
				fill(c.base, c.stride);
	for (auto &c : cs)
		c.gbps.clear();
	for (int r = 0; r < g_rounds; r++)
		for (auto &c : cs) {
			if (refill) /* the candidates share their memory: each is laid out anew before it is read */
				fill(c.base, c.stride);
			c.gbps.push_back(measure(c.base, c.stride));
		}
	if (!refill)
		for (auto &c : cs)
			if (c.stride == FRAME)
				for (int s = 0; s < 4; s++)
					c.lat[s] = chase_ns(c.base, CHASE_STRIDES[s]);
	double lo = 1e30, hi = 0;
	printf("%-58s %-16s %-30s %s\n", "pool", "address", "vectorscope kernel, median", "dependent loads, ns each, stride 4 KiB / 64 KiB / 2 MiB / 32 MiB");
	for (auto &c : cs) {
		const double m = median(c.gbps);
		lo = std::min(lo, m);
		hi = std::max(hi, m);
		printf("%-58s %#014llx  %7.1f GB/s = %.3f of peak   %5.0f %5.0f %5.0f %5.0f   all", c.name.c_str(), (unsigned long long)(uintptr_t)c.base, m,
		       m / 8000.0, c.lat[0], c.lat[1], c.lat[2], c.lat[3]);
		for (double g : c.gbps)
			printf(" %.0f", g);
		printf("\n");
	}
	printf("   spread of the medians: %.1f .. %.1f GB/s (%.1f %%)\n", lo, hi, 100.0 * (hi / lo - 1.0));
	fflush(stdout);
}

/* ---- virtual-memory helpers ------------------------------------------------------------------------------------------ */
static hipMemAllocationProp g_prop;
static size_t g_gran;
static size_t round_up(size_t x, size_t a)
{
	return (x + a - 1) / a * a;
}
static hipMemGenericAllocationHandle_t phys_create(size_t bytes)
{
	hipMemGenericAllocationHandle_t h;
	CK(hipMemCreate(&h, bytes, &g_prop, 0));
	return h;
}
static void set_rw(void *va, size_t bytes)
{
	hipMemAccessDesc d;
	memset(&d, 0, sizeof d);
	d.location.type = hipMemLocationTypeDevice;
	d.location.id = 0;
	d.flags = hipMemAccessFlagsProtReadWrite;
	CK(hipMemSetAccess(va, bytes, &d, 1));
}
static void map_rw(void *va, size_t bytes, hipMemGenericAllocationHandle_t h)
{
	CK(hipMemMap(va, bytes, 0, h, 0));
	set_rw(va, bytes);
}

int main(int argc, char **argv)
{
	if (argc > 1)
		g_rounds = atoi(argv[1]);
	const size_t slab_gib = argc > 2 ? (size_t)atoi(argv[2]) : 24;
	CK(hipSetDevice(0));
	if (scope_ctx_create(0, &g_ctx) != SCOPE_OK) {
		fprintf(stderr, "scope_ctx_create failed\n");
		return 1;
	}
	memset(&g_prop, 0, sizeof g_prop);
	g_prop.type = hipMemAllocationTypePinned;
	g_prop.location.type = hipMemLocationTypeDevice;
	g_prop.location.id = 0;
	size_t gmin = 0, grec = 0;
	CK(hipMemGetAllocationGranularity(&gmin, &g_prop, hipMemAllocationGranularityMinimum));
	CK(hipMemGetAllocationGranularity(&grec, &g_prop, hipMemAllocationGranularityRecommended));
	g_gran = std::max(gmin, (size_t)(2u << 20));
	size_t free_b = 0, total_b = 0;
	CK(hipMemGetInfo(&free_b, &total_b));
	printf("placement_vmm_probe: %u x %ux%u staged frames per launch (%.2f GB per pool), %d rounds; allocation granularity min %zu recommended %zu; HBM free %.1f of %.1f GB\n",
	       NF, W, H, POOL / 1e9, g_rounds, gmin, grec, free_b / 1e9, total_b / 1e9);

	/* the master copy: uniform noise through the product's own K0, as bench.py's frames */
	uint8_t *rgb;
	CK(hipMalloc(&rgb, FRAME));
	CK(hipMalloc(&g_src, POOL));
	CK(hipMalloc(&g_out, (size_t)NF * 65536));
	CK(hipMalloc(&g_ticks, 64));
	for (uint32_t f = 0; f < NF; f++) {
		noise_kernel<<<1024, 256>>>((uint32_t *)rgb, FRAME / 4, 7u + f);
		CK(hipDeviceSynchronize());
		SK(scope_convert_rgb_yuv(g_ctx, rgb, W * 4, g_src + f * FRAME, W * 4, W, H, 2, SCOPE_IN_DEVICE));
	}
	CK(hipFree(rgb));

	/* ---- 1: hipMalloc pools and single-handle pools, all held ------------------------------------------------------------ */
	const int KA = 8, KB = 8;
	const size_t GIB = 1ull << 30, SZ = round_up(POOL, g_gran), SLOT = round_up(SZ, GIB) + GIB;
	uint8_t *va = nullptr;
	CK(hipMemAddressReserve((void **)&va, SLOT * (KB + 2), GIB, nullptr, 0));
	std::vector<hipMemGenericAllocationHandle_t> hb(KB);
	std::vector<Cand> all;
	all.push_back(cand("hipMalloc #0 (the master copy, the process's first pool)", g_src));
	for (int k = 1; k < KA; k++) {
		uint8_t *p;
		CK(hipMalloc(&p, POOL));
		all.push_back(cand("hipMalloc #" + std::to_string(k), p));
	}
	for (int k = 0; k < KB; k++) {
		hb[k] = phys_create(SZ);
		map_rw(va + k * SLOT, SZ, hb[k]);
		all.push_back(cand("hipMemCreate handle " + std::to_string(k) + " at slot " + std::to_string(k), va + k * SLOT));
	}
	run_section("1: hipMalloc pools and single-handle hipMemCreate pools (1 GiB-aligned addresses), all held at once, read in turn", all);
	int fast = 0, slow = 0, vfast = KA, vslow = KA;
	for (int k = 0; k < (int)all.size(); k++) {
		if (median(all[k].gbps) > median(all[fast].gbps))
			fast = k;
		if (median(all[k].gbps) < median(all[slow].gbps))
			slow = k;
		if (k >= KA && median(all[k].gbps) > median(all[vfast].gbps))
			vfast = k;
		if (k >= KA && median(all[k].gbps) < median(all[vslow].gbps))
			vslow = k;
	}
	/* ---- 2: only two in the rotation ---------------------------------------------------------------------------------- */
	{
		std::vector<Cand> two = {all[fast], all[slow]};
		run_section("2: only the fastest and the slowest pool of section 1, alternating", two, false, true);
	}
	/* ---- 3: the fastest and the slowest handle swap their addresses ----------------------------------------------------------- */
	{
		const int hf = vfast - KA, hs = vslow - KA;
		if (hf != hs) {
			CK(hipMemUnmap(va + hf * SLOT, SZ));
			CK(hipMemUnmap(va + hs * SLOT, SZ));
			map_rw(va + hf * SLOT, SZ, hb[hs]);
			map_rw(va + hs * SLOT, SZ, hb[hf]);
			all[vfast].name = "SLOW handle " + std::to_string(hs) + " now at the fast one's slot " + std::to_string(hf);
			all[vslow].name = "FAST handle " + std::to_string(hf) + " now at the slow one's slot " + std::to_string(hs);
			std::swap(hb[hf], hb[hs]);
		}
		run_section("3: the fastest and the slowest handle have swapped their addresses; every pool of section 1 read again", all, false, true);
	}
	/* ---- 4: the address's alignment ------------------------------------------------------------------------------------ */
	{
		uint8_t *slot = va + KB * SLOT;
		const size_t offs[3] = {0, g_gran, 64u << 20};
		const int h = vfast - KA;
		CK(hipMemUnmap(va + h * SLOT, SZ));
		for (int i = 0; i < 3; i++) {
			std::vector<Cand> cs;
			map_rw(slot + offs[i], SZ, hb[h]);
			cs.push_back(cand("handle of slot " + std::to_string(h) + " at a 1 GiB boundary + " + std::to_string(offs[i] >> 20) + " MiB", slot + offs[i]));
			cs.push_back(all[fast < KA ? fast : 1]);
			cs.push_back(all[slow < KA ? slow : 2]);
			run_section("4: the same physical memory behind another address alignment (two hipMalloc pools beside it as controls)", cs);
			CK(hipMemUnmap(slot + offs[i], SZ));
		}
		map_rw(va + h * SLOT, SZ, hb[h]);
		fill(va + h * SLOT, FRAME);
	}
	/* ---- 6 / 7: along ONE physical allocation; the frames' stride ------------------------------------------------------------ */
	{
		const size_t slab = slab_gib * GIB;
		uint8_t *sva = nullptr;
		CK(hipMemAddressReserve((void **)&sva, slab, GIB, nullptr, 0));
		hipMemGenericAllocationHandle_t h = phys_create(slab);
		map_rw(sva, slab, h);
		std::vector<Cand> cs;
		const size_t step = round_up(POOL, 256u << 20); /* windows that do not overlap */
		for (size_t off = 0; off + POOL <= slab; off += step)
			cs.push_back(cand("one " + std::to_string(slab_gib) + " GiB handle, window at +" + std::to_string(off >> 20) + " MiB", sva + off));
		run_section("6: the pool at successive offsets inside ONE physical allocation", cs);
		const size_t pads[6] = {0, 4096, 65536, 1u << 20, (2u << 20) + 4096, FRAME / 4 + 8192};
		std::vector<Cand> fs;
		for (int pi = 0; pi < 6; pi++)
			if (NF * (FRAME + pads[pi]) <= slab)
				fs.push_back(cand("frame stride = frame + " + std::to_string(pads[pi]) + " B", sva, FRAME + pads[pi]));
		run_section("7: the 64 frames laid out with another stride (from the start of section 6's slab)", fs, true);
		CK(hipMemUnmap(sva, slab));
		CK(hipMemRelease(h));
		CK(hipMemAddressFree(sva, slab));
	}
	run_section("8: every pool of section 1 once more, at the end of the run", all, false, true);
	/* ---- 5 (run last: a driver that refuses a piecewise mapping must not cost the other sections): pools made of separately
	 * created chunks, each pool behind an address reservation of its own ------------------------------------------------ */
	{
		const size_t chunk_mib[3] = {256, 32, 2};
		for (int ci = 0; ci < 3; ci++) {
			const size_t chunk = std::max(chunk_mib[ci] << 20, g_gran);
			const size_t nchunks = (POOL + chunk - 1) / chunk;
			std::vector<Cand> cs;
			std::vector<std::vector<hipMemGenericAllocationHandle_t>> hs(2);
			uint8_t *slot[2] = {nullptr, nullptr};
			bool ok = true;
			for (int inst = 0; inst < 2 && ok; inst++) {
				ok = hipMemAddressReserve((void **)&slot[inst], nchunks * chunk, GIB, nullptr, 0) == hipSuccess;
				for (size_t c = 0; c < nchunks && ok; c++) {
					hipMemGenericAllocationHandle_t h;
					ok = hipMemCreate(&h, chunk, &g_prop, 0) == hipSuccess;
					if (!ok)
						break;
					hs[inst].push_back(h);
					ok = hipMemMap(slot[inst] + c * chunk, chunk, 0, h, 0) == hipSuccess;
					hipMemAccessDesc d;
					memset(&d, 0, sizeof d);
					d.location.type = hipMemLocationTypeDevice;
					d.location.id = 0;
					d.flags = hipMemAccessFlagsProtReadWrite;
					ok = ok && hipMemSetAccess(slot[inst] + c * chunk, chunk, &d, 1) == hipSuccess;
				}
				if (ok)
					cs.push_back(cand(std::to_string(nchunks) + " chunks of " + std::to_string(chunk >> 20) + " MiB, instance " + std::to_string(inst), slot[inst]));
			}
			if (!ok) {
				printf("\n== 5: chunks of %zu MiB: the driver refused the piecewise mapping (%s); skipped\n", chunk >> 20, hipGetErrorString(hipGetLastError()));
				fflush(stdout);
				break;
			}
			cs.push_back(all[fast < KA ? fast : 1]);
			run_section("5: a pool of separately created physical chunks mapped side by side (+ a hipMalloc pool as control)", cs);
			for (int inst = 0; inst < 2; inst++) {
				for (size_t c = 0; c < nchunks; c++) {
					CK(hipMemUnmap(slot[inst] + c * chunk, chunk));
					CK(hipMemRelease(hs[inst][c]));
				}
				CK(hipMemAddressFree(slot[inst], nchunks * chunk));
			}
		}
	}
	scope_ctx_destroy(g_ctx);
	printf("\ndone\n");
	return 0;
}
