/*
 * tests/emul/shm_rccl.c -- a stand-in for the handful of RCCL entry points libscope_hip.so binds
 * (scope_api.hip: rccl()), for ranks that are PROCESSES ON ONE HOST.  TEST INFRASTRUCTURE ONLY.
 *
 * RCCL refuses two ranks on one GPU and does not exist for the CPU emulation build, so the N > 1 code of bench.py and
 * of the C-ABI (scope_table_allreduce, the second-stream overlap, the launcher) could never run anywhere but on the
 * driver's multi-GPU node.  This library gives those paths something to run against on the boxes that exist:
 *
 *   - built with gcc against tests/emul (libshm_rccl_emul.so): buffers are host memory, a "stream" is synchronous;
 *   - built with hipcc -DSHM_RCCL_HIP (tools/libshm_rccl_hip.so): buffers are device memory, and the all-reduce is
 *     STREAM-ORDERED AND ASYNCHRONOUS like the real one: D2H copy into pinned staging, a host function on the stream
 *     that exchanges through the shared segment, H2D copy back.  Two processes may share one GPU.
 *
 * Selected with SCOPE_RCCL_LIBRARY=<path>.  Ranks meet in a POSIX shared-memory segment named by the unique id;
 * every wait has a deadline (SHM_RCCL_TIMEOUT_S, default 120 s) after which the call fails instead of hanging.
 * Only ncclUint32 + ncclSum are implemented: that is all the scope path sends.
 */
#define _GNU_SOURCE
#include <errno.h>
#include <fcntl.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#ifdef SHM_RCCL_HIP
#include <hip/hip_runtime_api.h>
typedef hipStream_t stream_t;
#else
typedef void *stream_t;
#endif

#define SLOT_BYTES (8u << 20) /* one rank's share of the exchange window; larger payloads go in rounds */
#define MAX_RANKS 64

typedef struct {
	char internal[128];
} ncclUniqueId;

struct seg_hdr {
	_Atomic uint32_t arrived;
	_Atomic uint32_t generation;
	_Atomic uint32_t joined;
	_Atomic uint32_t failed;
};

struct comm {
	struct seg_hdr *hdr;
	uint8_t *slots;
	size_t map_bytes;
	int n_ranks, rank;
	char name[128];
	uint32_t *staging; /* HIP build: pinned bounce buffer */
	size_t staging_bytes;
	double timeout_s;
};

static double now_s(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

/* sense-reversing barrier over the segment; 0 on success, -1 when a peer did not arrive in time */
static int seg_barrier(struct comm *c)
{
	struct seg_hdr *h = c->hdr;
	const uint32_t gen = atomic_load(&h->generation);
	if (atomic_fetch_add(&h->arrived, 1) + 1 == (uint32_t)c->n_ranks) {
		atomic_store(&h->arrived, 0);
		atomic_fetch_add(&h->generation, 1);
		return 0;
	}
	const double t_end = now_s() + c->timeout_s;
	unsigned spins = 0;
	while (atomic_load(&h->generation) == gen) {
		if (atomic_load(&h->failed))
			return -1;
		if (++spins > 2000) {
			usleep(50);
			if (now_s() > t_end) {
				atomic_store(&h->failed, 1);
				fprintf(stderr, "shm_rccl: rank %d waited %.0f s for its peers at a barrier: giving up\n", c->rank,
					c->timeout_s);
				return -1;
			}
		}
	}
	return 0;
}

/* `words` is host-visible memory holding this rank's contribution; on return it holds the sum over ranks */
static int exchange(struct comm *c, uint32_t *words, size_t count)
{
	const size_t per_round = SLOT_BYTES / 4;
	for (size_t done = 0; done < count; done += per_round) {
		const size_t n = count - done < per_round ? count - done : per_round;
		memcpy(c->slots + (size_t)c->rank * SLOT_BYTES, words + done, n * 4);
		if (seg_barrier(c))
			return -1;
		for (int r = 0; r < c->n_ranks; r++) {
			if (r == c->rank)
				continue;
			const uint32_t *peer = (const uint32_t *)(c->slots + (size_t)r * SLOT_BYTES);
			for (size_t i = 0; i < n; i++)
				words[done + i] += peer[i];
		}
		if (seg_barrier(c)) /* nobody overwrites a slot that is still being read */
			return -1;
	}
	return 0;
}

int ncclGetVersion(int *version)
{
	*version = 0; /* "not RCCL": bench.py prints what the library says */
	return 0;
}

const char *ncclGetErrorString(int rc)
{
	return rc == 0 ? "no error" : (rc == 5 ? "shm_rccl: invalid usage" : "shm_rccl: a peer did not arrive (timeout) or a system call failed");
}

int ncclGetUniqueId(ncclUniqueId *id)
{
	memset(id, 0, sizeof *id);
	snprintf(id->internal, sizeof id->internal, "/shm_rccl_%d_%llx", (int)getpid(), (unsigned long long)(now_s() * 1e6));
	return 0;
}

int ncclCommInitRank(void **out, int n_ranks, ncclUniqueId id, int rank)
{
	if (n_ranks < 1 || n_ranks > MAX_RANKS || rank < 0 || rank >= n_ranks || id.internal[0] != '/')
		return 5;
	struct comm *c = (struct comm *)calloc(1, sizeof *c);
	if (!c)
		return 2;
	const char *t = getenv("SHM_RCCL_TIMEOUT_S");
	c->timeout_s = t ? atof(t) : 120.0;
	c->n_ranks = n_ranks;
	c->rank = rank;
	snprintf(c->name, sizeof c->name, "%s", id.internal);
	c->map_bytes = 4096 + (size_t)n_ranks * SLOT_BYTES;
	const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
	if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) {
		perror("shm_rccl: shm_open/ftruncate");
		free(c);
		return 2;
	}
	void *m = mmap(NULL, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
	close(fd);
	if (m == MAP_FAILED) {
		perror("shm_rccl: mmap");
		free(c);
		return 2;
	}
	c->hdr = (struct seg_hdr *)m; /* a fresh segment is all zero */
	c->slots = (uint8_t *)m + 4096;
	atomic_fetch_add(&c->hdr->joined, 1);
	const double t_end = now_s() + c->timeout_s;
	while (atomic_load(&c->hdr->joined) < (uint32_t)n_ranks) {
		usleep(200);
		if (now_s() > t_end) {
			fprintf(stderr, "shm_rccl: rank %d: only %u of %d ranks joined %s\n", rank, atomic_load(&c->hdr->joined), n_ranks, c->name);
			munmap(m, c->map_bytes);
			free(c);
			return 1;
		}
	}
	if (seg_barrier(c)) {
		munmap(m, c->map_bytes);
		free(c);
		return 1;
	}
	if (rank == 0)
		shm_unlink(c->name); /* everybody has it mapped: the name can go, the memory lives until the last unmap */
	*out = c;
	return 0;
}

int ncclCommCount(const void *comm, int *count)
{
	*count = ((const struct comm *)comm)->n_ranks;
	return 0;
}

int ncclCommUserRank(const void *comm, int *rank)
{
	*rank = ((const struct comm *)comm)->rank;
	return 0;
}

int ncclCommDestroy(void *comm)
{
	struct comm *c = (struct comm *)comm;
#ifdef SHM_RCCL_HIP
	if (c->staging)
		(void)hipHostFree(c->staging);
#endif
	munmap(c->hdr, c->map_bytes);
	free(c);
	return 0;
}

#ifdef SHM_RCCL_HIP
struct host_job {
	struct comm *c;
	size_t offset_words, count;
};

static void host_exchange(void *p)
{
	struct host_job *j = (struct host_job *)p;
	if (exchange(j->c, j->c->staging + j->offset_words, j->count))
		atomic_store(&j->c->hdr->failed, 1);
	free(j);
}
#endif

int ncclAllReduce(const void *send, void *recv, size_t count, int datatype, int op, void *comm, stream_t stream)
{
	struct comm *c = (struct comm *)comm;
	if (datatype != 3 /* ncclUint32 */ || op != 0 /* ncclSum */ || !c)
		return 5;
	/* fault injection for the launcher's tests: this rank never enters the collective, its peers wait */
	const char *hang = getenv("SHM_RCCL_HANG_RANK");
	if (hang && atoi(hang) == c->rank)
		for (;;)
			sleep(1);
#ifdef SHM_RCCL_HIP
	/* stream-ordered and asynchronous: the caller returns at once, as from the real ncclAllReduce.  The bounce buffer is
	 * a ring of payload-sized regions so that a second collective may be enqueued while the first is in flight. */
	const size_t bytes = count * 4;
	if (c->staging_bytes < 4 * bytes) {
		/* growing the ring needs the old one to be idle */
		if (hipStreamSynchronize(stream) != hipSuccess)
			return 1;
		if (c->staging)
			(void)hipHostFree(c->staging);
		c->staging_bytes = 4 * bytes;
		if (hipHostMalloc((void **)&c->staging, c->staging_bytes, 0) != hipSuccess)
			return 2;
	}
	static _Thread_local unsigned turn;
	const size_t off = (size_t)(turn++ & 3) * count;
	struct host_job *j = (struct host_job *)malloc(sizeof *j);
	j->c = c;
	j->offset_words = off;
	j->count = count;
	if (hipMemcpyAsync(c->staging + off, send, bytes, hipMemcpyDeviceToHost, stream) != hipSuccess)
		return 1;
	if (hipLaunchHostFunc(stream, host_exchange, j) != hipSuccess)
		return 1;
	if (hipMemcpyAsync(recv, c->staging + off, bytes, hipMemcpyHostToDevice, stream) != hipSuccess)
		return 1;
	return atomic_load(&c->hdr->failed) ? 1 : 0;
#else
	(void)stream;
	if (recv != send)
		memcpy(recv, send, count * 4);
	return exchange(c, (uint32_t *)recv, count) ? 1 : 0;
#endif
}
