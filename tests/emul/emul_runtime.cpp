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
 * from several host threads are serialised. */
struct Fiber {
	ucontext_t uc;
	bool done;
};

static std::mutex launch_mu;
static thread_local ucontext_t main_uc;
static thread_local Fiber *cur_fiber = nullptr;
static void (*g_fn)(void *) = nullptr;
static void *g_ctx = nullptr;

static void fiber_entry()
{
	g_fn(g_ctx);
	cur_fiber->done = true;
	swapcontext(&cur_fiber->uc, &main_uc);
}

void block_barrier()
{
	swapcontext(&cur_fiber->uc, &main_uc);
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
	g_fn = thread_fn;
	g_ctx = ctx;
	blockDim = block;
	gridDim = grid;
	for (unsigned by = 0; by < grid.y; by++)
		for (unsigned bx = 0; bx < grid.x; bx++) {
			blockIdx = dim3(bx, by, 0);
			for (unsigned t = 0; t < n; t++) {
				Fiber &f = fibers[t];
				getcontext(&f.uc);
				f.uc.uc_stack.ss_sp = stacks[t];
				f.uc.uc_stack.ss_size = STACK;
				f.uc.uc_link = nullptr;
				f.done = false;
				makecontext(&f.uc, fiber_entry, 0);
			}
			for (unsigned live = n; live;) {
				live = 0;
				for (unsigned t = 0; t < n; t++) {
					Fiber &f = fibers[t];
					if (f.done)
						continue;
					threadIdx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
					cur_fiber = &f;
					swapcontext(&main_uc, &f.uc);
					live += f.done ? 0u : 1u;
				}
			}
		}
	cur_fiber = nullptr;
}

} // namespace emul

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
