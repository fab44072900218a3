/*
 * tests/emul/emul_runtime.cpp -- the other half of tests/emul/hip/hip_runtime.h.
 * TEST INFRASTRUCTURE ONLY: a fiber-per-GPU-thread executor and malloc-backed "device" memory.
 */
#include <hip/hip_runtime.h>

#include <mutex>
#include <stdlib.h>
#include <sys/mman.h>
#include <ucontext.h>
#include <memory>
#include <stdio.h>
#include <thread>
#include <vector>

thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;

/* the dynamic-LDS window of the one workgroup that runs at a time (160 KiB like a gfx950 CU) */
alignas(16) uint32_t scope_lds[160 * 1024 / 4];

namespace emul {

/* The GPU threads of a workgroup are FIBERS (ucontext) of the launching OS thread: thread t runs until its next
 * __syncthreads() -- or to its end -- then thread t + 1, and when every thread of the workgroup stands at the barrier the
 * sweep starts over.  (One OS thread per GPU thread, with a std::barrier, spent most of the test suite's time in futex
 * calls.)  Legal for kernels in which every thread of a workgroup reaches the same barriers, which HIP demands anyway;
 * nothing in the product's kernels spins on another thread.  Workgroups run one after the other (one LDS window), launches
 * from several host threads are serialised.
 *
 * Two wave models.  Default: a wave is ONE lane wide -- a ballot holds the caller's bit only, there is no lane below, a wave
 * sum is the lane's own value: cheap, and enough for everything that is per-lane arithmetic.  SCOPE_EMUL_WAVE64=1: the 64
 * threads of a wave run in LOCK STEP between cross-lane operations -- every ballot / readlane / lane-below / wave sum /
 * wave barrier is a rendezvous (wave_gather) of the wave's live lanes, which exchange their values there -- so the kernels'
 * wave votes, run-length merges (DPP), packed scratch rows (mbcnt) and wave-wide flushes run with the semantics of the
 * hardware, on the CPU (tests/test_wave64_emul.py).  Its rule is the hardware's rule for these operations as the kernels use
 * them: every live lane of a wave reaches the same cross-lane operation (uniform control flow around them); a wave whose
 * lanes stand at different operations, or partly at a workgroup barrier, aborts the process with a message. */
enum State { RUN, AT_WAVE, AT_BLOCK, DONE };
struct Fiber {
	ucontext_t uc;
	State st;
};
struct WaveBox {
	uint64_t vals[2][64];
	uint64_t mask[2], building;
	unsigned seq, site;
};

static std::mutex launch_mu;
static thread_local ucontext_t main_uc;
static thread_local Fiber *cur_fiber = nullptr;
static void (*g_fn)(void *) = nullptr;
static void *g_ctx = nullptr;
static WaveBox g_wave[32];

bool wave64()
{
	static const bool on = [] {
		const char *e = getenv("SCOPE_EMUL_WAVE64");
		return e && atoi(e) != 0;
	}();
	return on;
}

static void fiber_entry()
{
	g_fn(g_ctx);
	cur_fiber->st = DONE;
	swapcontext(&cur_fiber->uc, &main_uc);
}

void block_barrier()
{
	cur_fiber->st = AT_BLOCK;
	swapcontext(&cur_fiber->uc, &main_uc);
}

const uint64_t *wave_gather(uint64_t v, uint64_t *mask, unsigned site)
{
	const unsigned t = threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z);
	WaveBox &b = g_wave[t >> 6];
	const unsigned par = b.seq & 1u, lane = t & 63u;
	if (b.building == 0)
		b.site = site;
	else if (b.site != site) {
		fprintf(stderr, "emul (wave64): the lanes of wave %u stand at different cross-lane operations (%u and %u): divergent "
				"control flow around a wave operation\n", t >> 6, b.site, site);
		abort();
	}
	b.vals[par][lane] = v;
	b.building |= 1ull << lane;
	cur_fiber->st = AT_WAVE;
	swapcontext(&cur_fiber->uc, &main_uc);
	*mask = b.mask[par];
	return b.vals[par];
}

void run_grid(dim3 grid, dim3 block, void (*thread_fn)(void *), void *ctx)
{
	std::lock_guard<std::mutex> one(launch_mu);
	const unsigned n = block.x * block.y * block.z;
	constexpr size_t STACK = 256 * 1024;
	static std::vector<Fiber> fibers;
	static std::vector<char *> stacks;
	while (stacks.size() < n) {
		void *m = mmap(nullptr, STACK, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
		if (m == MAP_FAILED) {
			fprintf(stderr, "emul: cannot map a fiber stack\n");
			abort();
		}
		stacks.push_back((char *)m);
	}
	if (fibers.size() < n)
		fibers.resize(n);
	if (n > 64 * (sizeof g_wave / sizeof g_wave[0])) {
		fprintf(stderr, "emul: a workgroup of %u threads\n", n);
		abort();
	}
	g_fn = thread_fn;
	g_ctx = ctx;
	blockDim = block;
	gridDim = grid;
	const unsigned nwaves = (n + 63u) / 64u;
	for (unsigned by = 0; by < grid.y; by++)
		for (unsigned bx = 0; bx < grid.x; bx++) {
			blockIdx = dim3(bx, by, 0);
			for (unsigned t = 0; t < n; t++) {
				Fiber &f = fibers[t];
				getcontext(&f.uc);
				f.uc.uc_stack.ss_sp = stacks[t];
				f.uc.uc_stack.ss_size = STACK;
				f.uc.uc_link = nullptr;
				f.st = RUN;
				makecontext(&f.uc, fiber_entry, 0);
			}
			for (unsigned w = 0; w < nwaves; w++)
				g_wave[w].building = 0, g_wave[w].seq = 0;
			for (;;) {
				/* a sweep: every runnable thread runs to its next rendezvous (or to its end) */
				for (unsigned t = 0; t < n; t++) {
					Fiber &f = fibers[t];
					if (f.st != RUN)
						continue;
					threadIdx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
					cur_fiber = &f;
					swapcontext(&main_uc, &f.uc);
				}
				/* waves whose live lanes all stand at a cross-lane operation go on; a workgroup barrier opens when every live
				 * thread stands at it */
				unsigned live = 0, at_block = 0, released = 0;
				for (unsigned w = 0; w < nwaves; w++) {
					unsigned lw = 0, aw = 0, ab = 0;
					for (unsigned t = 64 * w; t < n && t < 64 * (w + 1); t++) {
						lw += fibers[t].st != DONE;
						aw += fibers[t].st == AT_WAVE;
						ab += fibers[t].st == AT_BLOCK;
					}
					live += lw;
					at_block += ab;
					if (aw == 0)
						continue;
					if (aw != lw) {
						fprintf(stderr, "emul (wave64): wave %u of workgroup %u: %u of %u live lanes stand at a cross-lane "
								"operation (site %u), %u at a workgroup barrier\n", w, bx, aw, lw, g_wave[w].site, ab);
						abort();
					}
					WaveBox &b = g_wave[w];
					b.mask[b.seq & 1u] = b.building;
					b.building = 0;
					b.seq++;
					for (unsigned t = 64 * w; t < n && t < 64 * (w + 1); t++)
						if (fibers[t].st == AT_WAVE)
							fibers[t].st = RUN, released++;
				}
				if (live == 0)
					break;
				if (released)
					continue;
				if (at_block != live) {
					fprintf(stderr, "emul: workgroup %u cannot go on (%u live threads, %u at the barrier)\n", bx, live, at_block);
					abort();
				}
				for (unsigned t = 0; t < n; t++)
					if (fibers[t].st == AT_BLOCK)
						fibers[t].st = RUN;
			}
		}
	cur_fiber = nullptr;
}

} // namespace emul

/* ---- self test of the wave64 model: the cross-lane vocabulary against closed forms (tests/test_wave64_emul.py) ------------ */
#include "scope_vec.h"
static unsigned g_selftest_bad;
static void selftest_thread(void *)
{
	const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	unsigned bad = 0;
	const uint64_t third = __ballot(lane % 3u == 0u); /* lanes 0, 3, 6, ... */
	uint64_t want = 0;
	for (unsigned l = 0; l < 64; l += 3)
		want |= 1ull << l;
	bad += third != want;
	bad += (unsigned)__popcll(__ballot(1)) != 64u;
	bad += __all(lane < 64u) != 1 || __all(lane != 17u) != 0;
	const unsigned below = scope_lane_below(1000u * wave + lane, 7u); /* row_shr:1 inside rows of 16, `fill` at a row's first lane */
	bad += below != ((lane & 15u) == 0u ? 7u : 1000u * wave + lane - 1u);
	bad += scope_lanes_below(third) != (lane + 2u) / 3u; /* multiples of three below this lane */
	bad += scope_wave_sum(lane + wave) != 2016u + 64u * wave;
	bad += __builtin_amdgcn_readlane(lane * lane + wave, 9u) != 81u + wave;
	if (wave == 1 && lane >= 40u) /* lanes that leave early are no part of what follows */
		goto out;
	bad += (unsigned)__popcll(__ballot(1)) != (wave == 1 ? 40u : 64u);
	__builtin_amdgcn_wave_barrier();
out:
	if (bad)
		__atomic_fetch_add(&g_selftest_bad, bad, __ATOMIC_RELAXED);
}
extern "C" int emul_wave64_selftest(void)
{
	if (!emul::wave64())
		return -1;
	g_selftest_bad = 0;
	emul::run_grid(dim3(2, 1, 1), dim3(192, 1, 1), selftest_thread, nullptr);
	return (int)g_selftest_bad;
}

static int emul_cus()
{
	const char *e = getenv("SCOPE_EMUL_CUS");
	const int n = e ? atoi(e) : 0;
	return n > 0 ? n : 6;
}

hipError_t hipGetDeviceCount(int *n)
{
	*n = 1;
	return hipSuccess;
}
hipError_t hipSetDevice(int)
{
	return hipSuccess;
}
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int)
{
	memset(p, 0, sizeof *p);
	snprintf(p->name, sizeof p->name, "CPU emulation (tests only)");
	snprintf(p->gcnArchName, sizeof p->gcnArchName, "emul");
	p->multiProcessorCount = emul_cus();
	p->maxSharedMemoryPerMultiProcessor = 160 * 1024;
	p->clockRate = 1;
	return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned)
{
	*s = (hipStream_t)malloc(1);
	return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t s)
{
	free(s);
	return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t)
{
	return hipSuccess;
}
hipError_t hipMalloc(void **p, size_t n)
{
	*p = aligned_alloc(256, (n + 255) / 256 * 256);
	if (*p)
		memset(*p, 0xCD, n); /* fresh HBM is not zero */
	return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipFree(void *p)
{
	free(p);
	return hipSuccess;
}
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t)
{
	memcpy(d, s, n);
	return hipSuccess;
}
hipError_t hipMemcpy2DAsync(void *d, size_t dp, const void *s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t)
{
	for (size_t r = 0; r < h; r++)
		memcpy((char *)d + r * dp, (const char *)s + r * sp, w);
	return hipSuccess;
}
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t)
{
	memset(d, v, n);
	return hipSuccess;
}
hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int)
{
	return hipSuccess;
}
hipError_t hipGetLastError(void)
{
	return hipSuccess;
}
const char *hipGetErrorString(hipError_t e)
{
	return e == hipSuccess ? "hipSuccess" : "emulated failure";
}

#include <chrono>
struct emul_event_t {
	std::chrono::steady_clock::time_point t;
};
hipError_t hipEventCreate(hipEvent_t *e)
{
	*e = new emul_event_t();
	return hipSuccess;
}
hipError_t hipEventDestroy(hipEvent_t e)
{
	delete e;
	return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t)
{
	e->t = std::chrono::steady_clock::now();
	return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t)
{
	return hipSuccess;
}
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b)
{
	*ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
	return hipSuccess;
}

hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned)
{
	return hipSuccess; /* everything is synchronous here */
}
hipError_t hipHostMalloc(void **p, size_t n, unsigned)
{
	*p = aligned_alloc(256, (n + 255) / 256 * 256);
	return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipHostFree(void *p)
{
	free(p);
	return hipSuccess;
}
