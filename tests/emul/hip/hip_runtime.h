/*
 * tests/emul/hip/hip_runtime.h -- CPU stand-in for the slice of the HIP runtime and device
 * vocabulary that obs-color-monitor_amd/csrc/*.hip uses.  TEST INFRASTRUCTURE ONLY.
 *
 * It lets `g++ -I tests/emul` compile the UNMODIFIED kernel and C-ABI sources into
 * tests/emul/libscope_emul.so so that the kernels' index arithmetic (slicing, pitches, tails,
 * LDS swizzles, flush/un-swizzle, partial-table reduction) and the host glue can be checked
 * against the oracle on machines without a GPU.  It is never built into, linked with or loaded
 * by the product library (libscope_hip.so), which has no CPU path at all.
 *
 * Model: one fiber (ucontext) per GPU thread of a workgroup, run in turn up to their next __syncthreads() by the
 * launching OS thread; workgroups run one after another, LDS is one static buffer.  Wave-level operations see a
 * wave of ONE active lane (a legal execution mask on the hardware), so code must already be
 * correct for arbitrary masks; the full-wave behaviour is covered by the -m gpu tests.
 */
#pragma once

#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <tuple>
#include <utility>

#define __global__
#define __device__
#define __host__
#define __shared__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __builtin_nontemporal_load(p) (*(p))
#define __builtin_nontemporal_store(v, p) ((void)(*(p) = (v)))
static inline uint32_t __brev(uint32_t v)
{
	uint32_t r = 0;
	for (int i = 0; i < 32; i++)
		r |= ((v >> i) & 1u) << (31 - i);
	return r;
}

/* ---- vector / launch types ---------------------------------------------------------------- */
struct alignas(8) uint2 {
	uint32_t x, y;
};
struct alignas(16) uint4 {
	uint32_t x, y, z, w;
};
static inline uint2 make_uint2(uint32_t x, uint32_t y)
{
	return uint2{x, y};
}
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w)
{
	return uint4{x, y, z, w};
}
struct dim3 {
	unsigned x, y, z;
	dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

extern thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;

/* ---- runtime types ------------------------------------------------------------------------ */
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
typedef struct emul_stream_t *hipStream_t;
enum { hipStreamNonBlocking = 1 };
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
struct hipDeviceProp_t {
	char name[256];
	char gcnArchName[256];
	int multiProcessorCount;
	size_t maxSharedMemoryPerMultiProcessor;
	int clockRate;
	size_t totalGlobalMem;
};

hipError_t hipGetDeviceCount(int *n);
hipError_t hipSetDevice(int d);
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int d);
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipMalloc(void **p, size_t n);
hipError_t hipFree(void *p);
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind k, hipStream_t st);
hipError_t hipMemcpy2DAsync(void *d, size_t dp, const void *s, size_t sp, size_t w, size_t h, hipMemcpyKind k,
			    hipStream_t st);
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t st);
hipError_t hipFuncSetAttribute(const void *f, hipFuncAttribute a, int v);
typedef struct emul_event_t *hipEvent_t;
hipError_t hipEventCreate(hipEvent_t *e);
enum { hipEventDisableTiming = 2 };
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned)
{
	return hipEventCreate(e);
}
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipHostMalloc(void **p, size_t n, unsigned flags);
hipError_t hipHostFree(void *p);
hipError_t hipGetLastError(void);
const char *hipGetErrorString(hipError_t e);

/* ---- device vocabulary -------------------------------------------------------------------- */
namespace emul {
void block_barrier();
/* SCOPE_EMUL_WAVE64=1: the 64 threads of a wave in lock step between cross-lane operations (emul_runtime.cpp) */
bool wave64();
/* the rendezvous of a wave's live lanes: every lane leaves `v`; returns the 64 slots and which of them are lanes that took part */
const uint64_t *wave_gather(uint64_t v, uint64_t *mask, unsigned site);
enum { SITE_BALLOT = 1, SITE_READLANE, SITE_BELOW, SITE_SUM, SITE_BARRIER, SITE_READFIRST };
/* runs fn(blockIdx) for every thread of every block */
void run_grid(dim3 grid, dim3 block, void (*thread_fn)(void *), void *ctx);
} // namespace emul

static inline void __syncthreads()
{
	emul::block_barrier();
}
static inline uint32_t atomicAdd(uint32_t *p, uint32_t v)
{
	return __atomic_fetch_add(p, v, __ATOMIC_RELAXED);
}
static inline uint32_t min(uint32_t a, uint32_t b)
{
	return a < b ? a : b;
}
static inline uint32_t max(uint32_t a, uint32_t b)
{
	return a > b ? a : b;
}
static inline float __fmaf_rn(float a, float b, float c)
{
	return fmaf(a, b, c);
}
/* the emulation build uses -ffp-contract=off, so these stay separate roundings */
static inline float __fmul_rn(float a, float b)
{
	return a * b;
}
static inline float __fadd_rn(float a, float b)
{
	return a + b;
}
static inline float __fsub_rn(float a, float b)
{
	return a - b;
}
/* wave of one active lane */
static inline uint32_t __builtin_amdgcn_readfirstlane(uint32_t x)
{
	return x;
}
static inline uint32_t __builtin_amdgcn_readlane(uint32_t x, uint32_t lane)
{
	if (emul::wave64()) {
		uint64_t m;
		const uint64_t *v = emul::wave_gather(x, &m, emul::SITE_READLANE);
		return (uint32_t)v[lane & 63u]; /* (an inactive lane's register is whatever it was; the kernels never ask for one) */
	}
	return x; /* the only active lane is the caller */
}
static inline uint64_t __ballot(int p)
{
	if (emul::wave64()) {
		uint64_t m, r = 0;
		const uint64_t *v = emul::wave_gather(p != 0, &m, emul::SITE_BALLOT);
		for (unsigned l = 0; l < 64; l++)
			if (((m >> l) & 1u) && v[l])
				r |= 1ull << l;
		return r;
	}
	return p ? (1ull << (threadIdx.x & 63u)) : 0ull;
}
static inline int __all(int p)
{
	if (emul::wave64())
		return __ballot(!p) == 0;
	return p != 0;
}
static inline int __ffsll(long long v)
{
	return __builtin_ffsll(v);
}
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n((p), (v), __ATOMIC_SEQ_CST)
#define __hip_atomic_load(p, order, scope) __atomic_load_n((p), __ATOMIC_SEQ_CST)
#define __hip_atomic_fetch_add(p, v, order, scope) __atomic_fetch_add((p), (v), __ATOMIC_SEQ_CST)
#define __hip_atomic_exchange(p, v, order, scope) __atomic_exchange_n((p), (v), __ATOMIC_SEQ_CST)
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
#define __builtin_amdgcn_s_setprio(x) ((void)0) /* issue priority: nothing to emulate */
#define __builtin_amdgcn_fence(order, scope) ((void)0) /* one-lane waves run to completion between yields: nothing to order */
static inline void __builtin_amdgcn_wave_barrier()
{
	if (emul::wave64()) { /* lock step: what the lanes stored before this point is there for every lane behind it */
		uint64_t m;
		(void)emul::wave_gather(0, &m, emul::SITE_BARRIER);
	}
}
#define __builtin_amdgcn_s_sleep(x) ((void)0)
static inline void __threadfence()
{
	__sync_synchronize();
}
static inline unsigned __umul24(unsigned a, unsigned b)
{
	return (a & 0xffffffu) * (b & 0xffffffu);
}

static inline int __mul24(int a, int b)
{
	/* v_mul_i32_i24: operands are the sign-extended low 24 bits, result the low 32 bits of the product */
	const int64_t x = (int32_t)((uint32_t)a << 8) >> 8, y = (int32_t)((uint32_t)b << 8) >> 8;
	return (int)(uint32_t)(uint64_t)(x * y);
}
static inline int __popcll(unsigned long long v)
{
	return __builtin_popcountll(v);
}

/* ---- kernel launch ------------------------------------------------------------------------ */
namespace emul {
template <typename... KArgs> struct launch_ctx {
	void (*kernel)(KArgs...);
	std::tuple<KArgs...> args;
	static void thread_entry(void *p)
	{
		launch_ctx *c = static_cast<launch_ctx *>(p);
		std::apply(c->kernel, c->args);
	}
};
} // namespace emul

template <typename... KArgs, typename... Args>
static inline void hipLaunchKernelGGL(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t /*lds*/, hipStream_t,
				      Args &&...args)
{
	emul::launch_ctx<KArgs...> c{kernel, std::tuple<KArgs...>(std::forward<Args>(args)...)};
	emul::run_grid(grid, block, &emul::launch_ctx<KArgs...>::thread_entry, &c);
}

/* <hip/hip_ext.h>: the launch that also delivers the kernel's time stamps into two events */
template <typename... KArgs, typename... Args>
static inline void hipExtLaunchKernelGGL(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t lds, hipStream_t st,
					 hipEvent_t start, hipEvent_t stop, unsigned /*flags*/, Args &&...args)
{
	hipEventRecord(start, st);
	hipLaunchKernelGGL(kernel, grid, block, lds, st, std::forward<Args>(args)...);
	hipEventRecord(stop, st);
}
