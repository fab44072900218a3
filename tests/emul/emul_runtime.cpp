/*
 * tests/emul/emul_runtime.cpp -- the other half of tests/emul/hip/hip_runtime.h.
 * TEST INFRASTRUCTURE ONLY: a thread-per-GPU-thread executor and malloc-backed "device" memory.
 */
#include <hip/hip_runtime.h>

#include <barrier>
#include <memory>
#include <stdio.h>
#include <thread>
#include <vector>

thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;

/* the dynamic-LDS window of the one workgroup that runs at a time (160 KiB like a gfx950 CU) */
alignas(16) uint32_t scope_lds[160 * 1024 / 4];

namespace emul {

static thread_local std::barrier<> *tls_barrier = nullptr;

void block_barrier()
{
	tls_barrier->arrive_and_wait();
}

void run_grid(dim3 grid, dim3 block, void (*thread_fn)(void *), void *ctx)
{
	const unsigned nthreads = block.x * block.y * block.z;
	std::barrier<> bar(nthreads);
	std::vector<std::thread> pool;
	pool.reserve(nthreads);
	for (unsigned t = 0; t < nthreads; t++) {
		pool.emplace_back([=, &bar]() {
			tls_barrier = &bar;
			blockDim = block;
			gridDim = grid;
			threadIdx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
			for (unsigned by = 0; by < grid.y; by++)
				for (unsigned bx = 0; bx < grid.x; bx++) {
					blockIdx = dim3(bx, by, 0);
					thread_fn(ctx);
					/* workgroups share the one LDS window: finish this one first */
					bar.arrive_and_wait();
				}
		});
	}
	for (auto &th : pool)
		th.join();
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
