/*
 * scope_api.hip -- the C-ABI of libscope_hip.so (include/scope_hip.h).
 *
 * Host-side glue only: argument checking, host<->HBM staging when the caller hands over host
 * buffers (the reference's mapped staging surface, src/common.c:343-364), partitioning of a
 * frame over the chip's workgroups, launching the kernels of scope_kernels.hip and combining
 * their partial tables.  There is no CPU implementation of the path in this library: without a
 * HIP device scope_ctx_create() fails with SCOPE_ENODEV and nothing else can be called.
 */
#include "../../include/scope_hip.h"
#include "scope_kernels.h"

#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <mutex>
#include <new>
#include <stdlib.h>
#include <string.h>

#define SCOPE_PROFILE_SLOTS 512u

struct scope_ctx {
	int device;
	int num_cu;
	hipStream_t own_stream;
	hipStream_t stream;
	uint8_t *partial; /* partial tables of one launch */
	size_t partial_bytes;
	uint8_t *pmask; /* which 16-bin groups of the vectorscope's sparse partials were written */
	size_t pmask_bytes;
	/* SCOPE_SIDE_REDUCE: a second stream that carries the partial-table reductions, two sets of partial tables used in
	 * turn, and the events that order them: acc_done[k] (the accumulation has filled set k, recorded on `stream`),
	 * side_done[k] (the reduction has read set k, recorded on `side`) */
	hipStream_t side;
	uint8_t *side_partial[2], *side_pmask[2];
	size_t side_partial_bytes[2], side_pmask_bytes[2];
	hipEvent_t acc_done[2], side_done[2];
	bool side_busy[2], side_pending;
	uint32_t side_slot;
	uint32_t *his_acc; /* SCOPE_MAX_FRAMES x 1024 accumulators of the histogram's in-launch combine; zero between launches */
	uint8_t *stage_in; /* host planes uploaded here */
	size_t stage_in_bytes;
	uint8_t *stage_out; /* tables on their way to a host buffer */
	size_t stage_out_bytes;
	/* Every entry point that touches the context takes this lock, so one context may be shared by the plugin's
	 * pipeline threads and its render thread (the reference runs one "color-monitor" thread per stand-alone
	 * source, src/common.c:375-403).  Recursive: scope_surface_begin() keeps it until scope_surface_end(). */
	std::recursive_mutex mu;
	uint64_t stat_vss_rows, stat_his_rows, stat_vss_fold, stat_trio_fold; /* scope_ctx_get_stat: launches that took a form (tests) */
	/* scope_surface_begin .. scope_surface_end: host planes uploaded once and reused by every scope call that is
	 * handed the same plane (the ROI / dock fan-out) */
	struct {
		bool active, fuse_k0;
		const uint8_t *host[2]; /* host plane a device copy stands for (NULL = slot free) */
		uint8_t *dev[2];
		size_t dev_bytes[2];
		uint32_t linesize, width, height; /* geometry of the cached planes */
		uint32_t pitch;                   /* of the device copies */
		uint32_t uploads, hits;
		/* the histogram the waveform pass of this surface produced on the side (scope_waveform inside begin/end):
		 * a later scope_histogram* call for the same plane and a subset of its channels is served from here */
		bool his_valid;
		const uint8_t *his_plane;
		int his_fuse;
		uint32_t his_chmask;
		uint32_t his_linesize, his_width, his_height; /* the counts are those of THIS geometry ... */
		int his_colorspace;                           /* ... and, through a fused K0, of this matrix */
		uint32_t his_counts[SCOPE_HIS_WORDS];
	} surf;
	uint32_t *tickets; /* SCOPE_MAX_FRAMES arrival counters of the in-launch combine; zero between launches */
	ScopeTuning tn;
	int slices_override;
	int profile;
	uint32_t timed; /* accumulation launches bracketed since profiling was switched on */
	hipEvent_t ev[2 * SCOPE_PROFILE_SLOTS];
	char err[256];
};

namespace {

int fail(scope_ctx *ctx, int status, const char *what, hipError_t e)
{
	if (ctx)
		snprintf(ctx->err, sizeof ctx->err, "%s: %s", what, e == hipSuccess ? "invalid argument" : hipGetErrorString(e));
	return status;
}

#define HIP_TRY(ctx, call)                                      \
	do {                                                    \
		hipError_t e_ = (call);                         \
		if (e_ != hipSuccess)                           \
			return fail(ctx, SCOPE_EIO, #call, e_); \
	} while (0)

int grow(scope_ctx *ctx, uint8_t **buf, size_t *cap, size_t need)
{
	if (need <= *cap)
		return SCOPE_OK;
	if (*buf) {
		/* the old buffer may still be in use by work queued on the stream */
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
		HIP_TRY(ctx, hipFree(*buf));
		*buf = NULL;
		*cap = 0;
	}
	void *p = NULL;
	hipError_t e = hipMalloc(&p, need);
	if (e != hipSuccess)
		return fail(ctx, SCOPE_ENOMEM, "hipMalloc", e);
	*buf = (uint8_t *)p;
	*cap = need;
	return SCOPE_OK;
}

/* Accumulation launch; with profiling on, the kernel's own begin / end time stamps go into the next pair of events
 * (the launcher's last argument `tm_`). */
#define TIMED_LAUNCH(ctx, call)                                                                    \
	do {                                                                                       \
		const uint32_t slot_ = 2u * ((ctx)->timed % SCOPE_PROFILE_SLOTS);                  \
		const ScopeTiming timing_ = {(ctx)->ev[slot_], (ctx)->ev[slot_ + 1]};               \
		const ScopeTiming *tm_ = (ctx)->profile ? &timing_ : NULL;                         \
		HIP_TRY(ctx, call);                                                                \
		if ((ctx)->profile)                                                                \
			(ctx)->timed++;                                                            \
	} while (0)

size_t round_up(size_t v, size_t a)
{
	return (v + a - 1) / a * a;
}

/* Geometry checks shared by every scope. */
int check_geometry(scope_ctx *ctx, uint32_t linesize, uint32_t width, uint32_t height)
{
	if ((uint64_t)width * 4 > linesize)
		return fail(ctx, SCOPE_EINVAL, "linesize < 4*width", hipSuccess);
	if ((uint64_t)width * height >= (1ull << 31))
		return fail(ctx, SCOPE_EINVAL, "frame larger than 2^31 pixels", hipSuccess);
	return SCOPE_OK;
}

/* Puts up to SCOPE_MAX_FRAMES planes into HBM (no-op for device planes) and describes them. */
int stage_planes(scope_ctx *ctx, const uint8_t *const *planes, uint32_t n, uint32_t linesize, uint32_t width,
		 uint32_t height, uint32_t flags, ScopeFrameSet *fs)
{
	fs->n_frames = n;
	fs->width = width;
	fs->height = height;
	if (flags & SCOPE_IN_DEVICE) {
		if (linesize & 3u)
			return fail(ctx, SCOPE_EINVAL, "device plane pitch must be a multiple of 4", hipSuccess);
		for (uint32_t i = 0; i < n; i++) {
			if (((uintptr_t)planes[i]) & 3u)
				return fail(ctx, SCOPE_EINVAL, "device plane must be 4-byte aligned", hipSuccess);
			fs->plane[i] = planes[i];
		}
		fs->linesize = linesize;
		return SCOPE_OK;
	}
	const size_t pitch = round_up((size_t)width * 4, 16);
	const size_t plane_bytes = round_up(pitch * height, 256);
	if (ctx->surf.active && n == 1) { /* one upload per plane between scope_surface_begin and _end */
		if (ctx->surf.host[0] || ctx->surf.host[1]) {
			if (ctx->surf.linesize != linesize || ctx->surf.width != width || ctx->surf.height != height)
				return fail(ctx, SCOPE_EINVAL, "geometry changed inside scope_surface_begin/end", hipSuccess);
		} else {
			ctx->surf.linesize = linesize;
			ctx->surf.width = width;
			ctx->surf.height = height;
			ctx->surf.pitch = (uint32_t)pitch;
		}
		int slot = -1;
		for (int k = 0; k < 2; k++)
			if (ctx->surf.host[k] == planes[0])
				slot = k;
		if (slot < 0) {
			slot = ctx->surf.host[0] ? 1 : 0;
			if (ctx->surf.host[slot])
				return fail(ctx, SCOPE_EINVAL, "more than two planes inside scope_surface_begin/end", hipSuccess);
			int st2 = grow(ctx, &ctx->surf.dev[slot], &ctx->surf.dev_bytes[slot], plane_bytes);
			if (st2)
				return st2;
			HIP_TRY(ctx, hipMemcpy2DAsync(ctx->surf.dev[slot], pitch, planes[0], linesize, (size_t)width * 4, height,
						      hipMemcpyHostToDevice, ctx->stream));
			ctx->surf.host[slot] = planes[0];
			ctx->surf.uploads++;
		} else {
			ctx->surf.hits++;
		}
		fs->plane[0] = ctx->surf.dev[slot];
		fs->linesize = (uint32_t)pitch;
		return SCOPE_OK;
	}
	int st = grow(ctx, &ctx->stage_in, &ctx->stage_in_bytes, plane_bytes * n);
	if (st)
		return st;
	for (uint32_t i = 0; i < n; i++) {
		uint8_t *d = ctx->stage_in + plane_bytes * i;
		HIP_TRY(ctx, hipMemcpy2DAsync(d, pitch, planes[i], linesize, (size_t)width * 4, height, hipMemcpyHostToDevice,
					      ctx->stream));
		fs->plane[i] = d;
	}
	fs->linesize = (uint32_t)pitch;
	return SCOPE_OK;
}

/* Where the kernels should leave `bytes` of results, and how they reach the caller. */
int result_begin(scope_ctx *ctx, void *user_out, size_t bytes, uint32_t flags, uint8_t **dev_out)
{
	if (flags & SCOPE_OUT_DEVICE) {
		*dev_out = (uint8_t *)user_out;
		return SCOPE_OK;
	}
	int st = grow(ctx, &ctx->stage_out, &ctx->stage_out_bytes, round_up(bytes, 256));
	if (st)
		return st;
	*dev_out = ctx->stage_out;
	return SCOPE_OK;
}

int result_end(scope_ctx *ctx, void *user_out, size_t bytes, uint32_t flags, const uint8_t *dev_out)
{
	if (!(flags & SCOPE_OUT_DEVICE)) {
		HIP_TRY(ctx, hipMemcpyAsync(user_out, dev_out, bytes, hipMemcpyDeviceToHost, ctx->stream));
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
		return SCOPE_OK;
	}
	if (!(flags & SCOPE_ASYNC))
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return SCOPE_OK;
}

/* Resolves a scope_region against a width x height target: scale, origin and size of the staged plane. */
int resolve_region(uint32_t width, uint32_t height, const struct scope_region *r, uint32_t *scale, uint32_t *x,
		   uint32_t *y, uint32_t *cx, uint32_t *cy)
{
	const uint32_t s = (r && r->scale) ? r->scale : 1u;
	if (r && r->scale == 0 && (r->x | r->y | r->cx | r->cy))
		return SCOPE_EINVAL;
	const uint32_t sw = width / s, sh = height / s; /* src/common.c:249-250 */
	if (sw == 0 || sh == 0)
		return SCOPE_EINVAL; /* the reference skips the frame, :251-254 */
	*scale = s;
	if (!r || r->cx == 0 || r->cy == 0) {
		*x = *y = 0;
		*cx = sw;
		*cy = sh;
		return SCOPE_OK;
	}
	if (r->x >= sw || r->y >= sh || r->cx > sw - r->x || r->cy > sh - r->y)
		return SCOPE_EINVAL;
	*x = r->x;
	*y = r->y;
	*cx = r->cx;
	*cy = r->cy;
	return SCOPE_OK;
}

/* The context's stream waits for the side stream's outstanding reductions (device-side: one event). */
int side_fence(scope_ctx *ctx)
{
	if (!ctx->side_pending)
		return SCOPE_OK;
	const uint32_t k = ctx->side_slot & 1u; /* the most recent reduction; the side stream runs them in order */
	HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->side_done[k], 0));
	ctx->side_pending = false;
	return SCOPE_OK;
}

int check_flags(scope_ctx *ctx, uint32_t flags)
{
	if ((flags & SCOPE_ASYNC) && (flags & (SCOPE_IN_DEVICE | SCOPE_OUT_DEVICE)) != (SCOPE_IN_DEVICE | SCOPE_OUT_DEVICE))
		return fail(ctx, SCOPE_EINVAL, "SCOPE_ASYNC needs device input and output", hipSuccess);
	if ((flags & SCOPE_SIDE_REDUCE) && !(flags & SCOPE_ASYNC))
		return fail(ctx, SCOPE_EINVAL, "SCOPE_SIDE_REDUCE needs SCOPE_ASYNC", hipSuccess);
	if (!(flags & SCOPE_SIDE_REDUCE))
		return side_fence(ctx); /* a call without the flag sees every earlier table complete */
	return SCOPE_OK;
}

/* How many workgroups share one frame so that one launch covers the chip once. */
/* The streaming kernels address a slice with 32-bit byte offsets from its first row: a frame whose rows span
 * 2 GiB or more (huge pitch padding, or > 2^29 pixels) must be cut into at least this many slices.  0 = too big
 * for the 256 partial tables the reductions accept. */
uint32_t min_slices_for_span(uint32_t linesize, uint32_t height)
{
	const uint64_t span = (uint64_t)linesize * height;
	const uint64_t need = span / (1ull << 31) + 1;
	return need > 256 ? 0u : (uint32_t)need;
}

uint32_t pick_slices(const scope_ctx *ctx, uint32_t frames_in_launch, uint32_t units_per_frame)
{
	if (ctx->slices_override > 0)
		return (uint32_t)ctx->slices_override > 256u ? 256u : (uint32_t)ctx->slices_override;
	const uint32_t busy = frames_in_launch * units_per_frame;
	uint32_t s = busy ? (uint32_t)ctx->num_cu / busy : 1;
	if (s < 1)
		s = 1;
	if (s > 256)
		s = 256;
	return s;
}

/* waveform.c:228-234 == histogram.c:367-373.  Returns 1 and fills plane/fuse when a plane is
 * selectable, 0 when `components` selects none, SCOPE_ENODATA when the selected one is NULL. */
/* Inside scope_surface_begin(SCOPE_FUSE_K0) .. _end the YUV plane is never uploaded: scopes that want it read the
 * (cached) RGB plane through the fused K0 instead. */
bool surface_prefers_fused(const scope_ctx *ctx, const struct scope_surface_data *sd)
{
	return ctx->surf.active && ctx->surf.fuse_k0 && sd->rgb_data != NULL;
}

int pick_plane(const scope_ctx *ctx, const struct scope_surface_data *sd, uint32_t components, uint32_t flags,
	       const uint8_t **plane, int *fuse)
{
	*fuse = 0;
	if (components & 0x07u) {
		*plane = sd->rgb_data;
	} else if (components & 0x70u) {
		*plane = sd->yuv_data;
		if ((!*plane || surface_prefers_fused(ctx, sd)) && ((flags & SCOPE_FUSE_K0) || surface_prefers_fused(ctx, sd))) {
			*plane = sd->rgb_data;
			*fuse = 1;
		}
	} else {
		return 0;
	}
	return *plane ? 1 : SCOPE_ENODATA;
}

uint32_t channel_mask(uint32_t components)
{
	/* calc_b/g/r = components & 0x11/0x22/0x44 (waveform.c:236-238, histogram.c:375-377) */
	return (components | (components >> 4)) & 7u;
}

int run_vectorscope(scope_ctx *ctx, const uint8_t *const *planes, uint32_t n, uint32_t linesize, uint32_t width,
		    uint32_t height, int fuse, int colorspace, uint32_t flags, uint8_t *out)
{
	int st = check_flags(ctx, flags);
	if (!st)
		st = check_geometry(ctx, linesize, width, height);
	if (st)
		return st;
	/* SCOPE_OUT_PACKED16: tables leave as uint32[32768], two bins per word -- what the multi-GPU sum puts on the wire */
	const bool packed = (flags & SCOPE_OUT_PACKED16) != 0;
	if (packed && !(flags & SCOPE_OUT_DEVICE))
		return fail(ctx, SCOPE_EINVAL, "SCOPE_OUT_PACKED16 needs SCOPE_OUT_DEVICE", hipSuccess);
	const size_t table = packed ? (size_t)SCOPE_VSS_BINS * 2 : (size_t)SCOPE_VSS_BINS;
	uint8_t *dev_out = NULL;
	st = result_begin(ctx, out, (size_t)n * table, flags, &dev_out);
	if (st)
		return st;
	for (uint32_t f0 = 0; f0 < n; f0 += SCOPE_MAX_FRAMES) {
		const uint32_t nf = (n - f0 < SCOPE_MAX_FRAMES) ? n - f0 : SCOPE_MAX_FRAMES;
		ScopeFrameSet fs;
		st = stage_planes(ctx, planes + f0, nf, linesize, width, height, flags, &fs);
		if (st)
			return st;
		uint8_t *dst = dev_out + (size_t)f0 * table;
		uint32_t slices = pick_slices(ctx, nf, 1);
		/* ONE frame over staged planes: five eighths of the CUs (160 slices on 256) beat all of them -- 16.2 against
		 * 17.1 us per 4K call on noise, 15.0 against 15.8 on natural content, 152 and 168 in between
		 * (profiles/round2/latency_vss_slices_g11.jsonl): the pixel loop is latency-bound either way, and fewer slices
		 * are fewer partial tables to write out and to add up.  With K0 fused the loop is VALU-bound and wants every CU
		 * (256: 18.2 us, 160: 20.2); two and more frames are flat around the full chip. */
		if (ctx->slices_override <= 0 && nf == 1 && !fuse && slices > 16)
			slices = (slices * 5u / 8u + 15u) & ~15u;
		const uint32_t min_slices = min_slices_for_span(fs.linesize, height);
		if (min_slices == 0)
			return fail(ctx, SCOPE_EINVAL, "pitch * height too large", hipSuccess);
		if (slices < min_slices)
			slices = min_slices;
		const uint32_t mask_block = ctx->tn.vss_sparse ? scope_vss_mask_block(ctx->tn) : 0u;
		ctx->stat_vss_rows += scope_vss_rows_mode_host(fs, slices, ctx->tn) ? 1u : 0u;
		if (slices == 1 && !packed) {
			TIMED_LAUNCH(ctx, scope_launch_vss(fs, 1, fuse, colorspace, dst, NULL, ctx->tn, ctx->stream, tm_));
		} else if ((flags & SCOPE_SIDE_REDUCE) && mask_block) {
			/* the accumulation on the context's stream, its reduction on the side stream: the next call's accumulation does
			 * not wait for it (it fills the OTHER set of partial tables) */
			const uint32_t k = (ctx->side_slot += 1u) & 1u;
			if ((size_t)nf * slices * SCOPE_VSS_BINS > ctx->side_partial_bytes[k] || (size_t)nf * slices * mask_block > ctx->side_pmask_bytes[k])
				HIP_TRY(ctx, hipStreamSynchronize(ctx->side)); /* grow() frees: nothing of the side stream may still read the set */
			st = grow(ctx, &ctx->side_partial[k], &ctx->side_partial_bytes[k], (size_t)nf * slices * SCOPE_VSS_BINS);
			if (!st)
				st = grow(ctx, &ctx->side_pmask[k], &ctx->side_pmask_bytes[k], (size_t)nf * slices * mask_block);
			if (st)
				return st;
			if (ctx->side_busy[k]) /* the reduction that last read this set (two calls ago) must be through */
				HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->side_done[k], 0));
			TIMED_LAUNCH(ctx, scope_launch_vss(fs, slices, fuse, colorspace, ctx->side_partial[k], ctx->side_pmask[k], ctx->tn,
							   ctx->stream, tm_));
			HIP_TRY(ctx, hipEventRecord(ctx->acc_done[k], ctx->stream));
			HIP_TRY(ctx, hipStreamWaitEvent(ctx->side, ctx->acc_done[k], 0));
			HIP_TRY(ctx, scope_launch_sat_reduce_sparse(ctx->side_partial[k], ctx->side_pmask[k], dst, nf, slices, mask_block, packed,
								    ctx->side));
			HIP_TRY(ctx, hipEventRecord(ctx->side_done[k], ctx->side));
			ctx->side_busy[k] = true;
			ctx->side_pending = true;
		} else {
			st = grow(ctx, &ctx->partial, &ctx->partial_bytes, (size_t)nf * slices * SCOPE_VSS_BINS);
			if (!st && mask_block)
				st = grow(ctx, &ctx->pmask, &ctx->pmask_bytes, (size_t)nf * slices * mask_block);
			if (st)
				return st;
			/* a few slices per frame (the 64-frame batches: four): they meet inside the launch, no reduction kernel */
			const ScopeVssFoldSink sink = {dst, ctx->tickets + SCOPE_MAX_FRAMES, packed ? 1 : 0};
			const bool folded = mask_block && scope_vss_can_fold(slices, ctx->pmask, ctx->tn);
			TIMED_LAUNCH(ctx, scope_launch_vss(fs, slices, fuse, colorspace, ctx->partial, mask_block ? ctx->pmask : NULL,
							   ctx->tn, ctx->stream, tm_, folded ? &sink : NULL));
			ctx->stat_vss_fold += folded ? 1u : 0u;
			/* one partial per frame and a packed result: the "reduction" of one table is just the repacking */
			if (folded)
				;
			else if (mask_block)
				HIP_TRY(ctx, scope_launch_sat_reduce_sparse(ctx->partial, ctx->pmask, dst, nf, slices, mask_block, packed,
									    ctx->stream));
			else
				HIP_TRY(ctx, scope_launch_sat_reduce(ctx->partial, dst, nf, slices, SCOPE_VSS_BINS, packed, ctx->stream));
		}
		if (!(flags & SCOPE_IN_DEVICE) && f0 + nf < n)
			HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); /* stage_in is reused by the next chunk */
	}
	return result_end(ctx, out, (size_t)n * table, flags, dev_out);
}

/* his_counts != NULL: the same pass over the planes also leaves their K3 count tables (uint32[n][256][4], host or
 * device like `out`) for the channels of his_chmask that chmask covers -- the caller checks that it covers them all. */
int run_waveform(scope_ctx *ctx, const uint8_t *const *planes, uint32_t n, uint32_t linesize, uint32_t width,
		 uint32_t height, uint32_t chmask, int fuse, int colorspace, uint32_t flags, uint8_t *out,
		 uint32_t *his_counts = NULL, uint32_t his_chmask = 0)
{
	int st = check_flags(ctx, flags);
	if (!st)
		st = check_geometry(ctx, linesize, width, height);
	if (st)
		return st;
	if (width == 0)
		return SCOPE_OK; /* empty table */
	const size_t table = (size_t)SCOPE_WV_SIZE * width * 4;
	const size_t his_table = SCOPE_HIS_WORDS * sizeof(uint32_t);
	uint8_t *dev_out = out, *dev_his = (uint8_t *)his_counts;
	if (!(flags & SCOPE_OUT_DEVICE)) {
		const size_t his_at = round_up((size_t)n * table, 256);
		st = grow(ctx, &ctx->stage_out, &ctx->stage_out_bytes, his_at + (his_counts ? round_up((size_t)n * his_table, 256) : 0));
		if (st)
			return st;
		dev_out = ctx->stage_out;
		dev_his = his_counts ? ctx->stage_out + his_at : NULL;
	}
	const uint32_t strips = (width + SCOPE_WVS_STRIP - 1) / SCOPE_WVS_STRIP;
	for (uint32_t f0 = 0; f0 < n; f0 += SCOPE_MAX_FRAMES) {
		const uint32_t nf = (n - f0 < SCOPE_MAX_FRAMES) ? n - f0 : SCOPE_MAX_FRAMES;
		ScopeFrameSet fs;
		st = stage_planes(ctx, planes + f0, nf, linesize, width, height, flags, &fs);
		if (st)
			return st;
		uint8_t *dst = dev_out + (size_t)f0 * table;
		const ScopeHisSink sink = {ctx->his_acc, ctx->tickets, (uint32_t *)(dev_his ? dev_his + (size_t)f0 * his_table : NULL),
					   his_chmask};
		const ScopeHisSink *his = dev_his ? &sink : NULL;
		uint32_t bands = pick_slices(ctx, nf, strips);
		if (bands > 16)
			bands = 16;
		/* A launch that would need row bands to fill the chip (one 4K frame: 60 strips) takes 16-column strips instead
		 * when that gives one round of units (240): no partial tables, no second kernel -- 20.6 -> 15.1 us per 4K
		 * frame.  Two rounds of narrow units lose to row bands (two 4K frames: 27.1 vs 24.7 us).  Vector shape only. */
		const int narrow = ctx->tn.wvs_strip != 64 &&
				   (ctx->tn.wvs_strip == 16 || (bands > 1 && nf * scope_wvs_strips(width, 1) <= (uint32_t)ctx->num_cu)) &&
				   scope_wvs_vector_shape(fs, dst, ctx->tn);
		if (narrow)
			bands = 1;
		/* the partial tables are combined 16 bytes at a time */
		if (bands > 1 && ((table & 15u) || (((uintptr_t)dst) & 15u)))
			bands = 1;
		/* the kernel counts in 16 bits per column and band: a band holds at most 65535 rows */
		const uint32_t min_bands = (height + 65534u) / 65535u;
		if (min_bands > 256)
			return fail(ctx, SCOPE_EINVAL, "frame taller than 256 * 65535 rows", hipSuccess); /* the reduction sums <= 256 partials */
		if (bands < min_bands) {
			if (((uintptr_t)dst) & 15u)
				return fail(ctx, SCOPE_EINVAL, "frames taller than 65535 rows need a 16-byte aligned table", hipSuccess);
			bands = min_bands;
		}
		if (bands == 1) {
			TIMED_LAUNCH(ctx, scope_launch_wvs(fs, 1, chmask, fuse, colorspace, dst, ctx->num_cu, ctx->tn, ctx->stream, his, tm_,
							   narrow));
		} else {
			st = grow(ctx, &ctx->partial, &ctx->partial_bytes, (size_t)nf * bands * table);
			if (st)
				return st;
			TIMED_LAUNCH(ctx, scope_launch_wvs(fs, bands, chmask, fuse, colorspace, ctx->partial, ctx->num_cu, ctx->tn, ctx->stream,
							   his, tm_, narrow));
			HIP_TRY(ctx, scope_launch_sat_reduce(ctx->partial, dst, nf, bands, table, 0, ctx->stream));
		}
		if (!(flags & SCOPE_IN_DEVICE) && f0 + nf < n)
			HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	}
	if (!(flags & SCOPE_OUT_DEVICE)) {
		HIP_TRY(ctx, hipMemcpyAsync(out, dev_out, (size_t)n * table, hipMemcpyDeviceToHost, ctx->stream));
		if (his_counts)
			HIP_TRY(ctx, hipMemcpyAsync(his_counts, dev_his, (size_t)n * his_table, hipMemcpyDeviceToHost, ctx->stream));
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	} else if (!(flags & SCOPE_ASYNC)) {
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	}
	return SCOPE_OK;
}

/* The dock's three scopes from ONE read of device-resident RGB planes: 1 = done (enqueued, and waited for unless
 * SCOPE_ASYNC), 0 = this launch does not qualify (the caller runs the separate passes), < 0 = error.
 * cw / ch: the waveform's and the histogram's RGB channel masks, ch a subset of cw.
 * dock = 1 (default): scope_trio_kernel -- both tables resident in LDS, any number of frames per launch, one vectorscope
 * partial per slice; dock = 2: round 2's scope_dock_kernel where one round of strips covers the launch (else not done). */
int run_dock(scope_ctx *ctx, const uint8_t *const *planes, uint32_t n, uint32_t linesize, uint32_t width, uint32_t height,
	     uint32_t cw, uint32_t ch, int colorspace, uint32_t flags, uint8_t *vss_out, uint8_t *wvs_out, uint32_t *his_counts)
{
	const uint32_t dev = SCOPE_IN_DEVICE | SCOPE_OUT_DEVICE;
	if (!ctx->tn.dock || (flags & dev) != dev || (flags & SCOPE_OUT_PACKED16) || !cw || (ch & ~cw) || !width || !height || n == 0)
		return 0;
	int st = check_geometry(ctx, linesize, width, height);
	if (st)
		return st;
	const uint32_t mask_block = SCOPE_BLOCK;
	const size_t wvs_table = (size_t)SCOPE_WV_SIZE * width * 4;
	bool launched = false;
	for (uint32_t f0 = 0; f0 < n; f0 += SCOPE_MAX_FRAMES) {
		const uint32_t nf = (n - f0 < SCOPE_MAX_FRAMES) ? n - f0 : SCOPE_MAX_FRAMES;
		ScopeFrameSet fs;
		st = stage_planes(ctx, planes + f0, nf, linesize, width, height, flags, &fs);
		if (st)
			return st;
		uint8_t *wvs_dst = wvs_out + (size_t)f0 * wvs_table;
		uint8_t *vss_dst = vss_out + (size_t)f0 * SCOPE_VSS_BINS;
		uint32_t *his_dst = his_counts ? his_counts + (size_t)f0 * SCOPE_HIS_WORDS : NULL;
		const ScopeHisSink sink = {ctx->his_acc, ctx->tickets, his_dst, ch};
		const bool old_form = ctx->tn.dock == 2;
		if (old_form ? !scope_dock_fits(fs, wvs_dst, (uint32_t)ctx->num_cu, ctx->tn) : !scope_trio_fits(fs, wvs_dst, ctx->tn)) {
			if (launched) /* cannot happen: every chunk has the same geometry and table alignment */
				return fail(ctx, SCOPE_EINVAL, "three scopes: chunks of one call disagree on the launch shape", hipSuccess);
			return 0;
		}
		const uint32_t sw = scope_trio_strip_width(fs, (uint32_t)ctx->num_cu, ctx->tn);
		const uint32_t slices = old_form ? scope_wvs_strips(width, 1) : scope_trio_slices(fs, (uint32_t)ctx->num_cu, sw);
		const bool partials = slices > 1 || old_form; /* scope_dock_kernel always writes a sparse partial per strip */
		bool folded = false; /* scope_trio_kernel with 2 .. 4 slices per frame: they meet inside the launch */
		if (partials) {
			st = grow(ctx, &ctx->partial, &ctx->partial_bytes, (size_t)nf * slices * SCOPE_VSS_BINS);
			if (!st)
				st = grow(ctx, &ctx->pmask, &ctx->pmask_bytes, (size_t)nf * slices * mask_block);
			if (st)
				return st;
		}
		if (old_form) {
			TIMED_LAUNCH(ctx, scope_launch_dock(fs, cw, colorspace, wvs_dst, his_dst ? &sink : NULL, ctx->partial, ctx->pmask,
							    ctx->stream, tm_));
		} else {
			folded = slices > 1 && scope_vss_can_fold(slices, ctx->pmask, ctx->tn);
			const ScopeVssFoldSink vsink = {vss_dst, ctx->tickets + SCOPE_MAX_FRAMES, 0};
			TIMED_LAUNCH(ctx, scope_launch_trio(fs, slices, sw, cw, colorspace, wvs_dst, his_dst ? &sink : NULL,
							    slices > 1 ? ctx->partial : vss_dst, ctx->pmask, ctx->stream, tm_, (uint32_t)ctx->tn.vss_rle,
							    folded ? &vsink : NULL));
		}
		ctx->stat_trio_fold += folded ? 1u : 0u;
		if (partials && !folded)
			HIP_TRY(ctx, scope_launch_sat_reduce_sparse(ctx->partial, ctx->pmask, vss_dst, nf, slices, mask_block, 0, ctx->stream));
		launched = true;
	}
	if (!(flags & SCOPE_ASYNC))
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return 1;
}

int run_histogram(scope_ctx *ctx, const uint8_t *const *planes, uint32_t n, uint32_t linesize, uint32_t width,
		  uint32_t height, uint32_t chmask, int fuse, int colorspace, uint32_t flags, uint32_t *counts)
{
	int st = check_flags(ctx, flags);
	if (!st)
		st = check_geometry(ctx, linesize, width, height);
	if (st)
		return st;
	const size_t table = SCOPE_HIS_WORDS * sizeof(uint32_t);
	uint8_t *dev_out = NULL;
	st = result_begin(ctx, counts, (size_t)n * table, flags, &dev_out);
	if (st)
		return st;
	for (uint32_t f0 = 0; f0 < n; f0 += SCOPE_MAX_FRAMES) {
		const uint32_t nf = (n - f0 < SCOPE_MAX_FRAMES) ? n - f0 : SCOPE_MAX_FRAMES;
		ScopeFrameSet fs;
		st = stage_planes(ctx, planes + f0, nf, linesize, width, height, flags, &fs);
		if (st)
			return st;
		uint32_t *dst = (uint32_t *)(dev_out + (size_t)f0 * table);
		uint32_t slices = pick_slices(ctx, nf, 1);
		/* ONE frame: three quarters of the CUs with two loads per lane per batch (scope_launch_his) -- 13.8 against
		 * 14.6 us per 4K call (profiles/round2/latency_histogram_g14.jsonl); three and more frames want the full chip */
		if (ctx->slices_override <= 0 && nf == 1 && !fuse && slices > 16)
			slices = (slices * 3u / 4u + 15u) & ~15u;
		const uint32_t min_slices = min_slices_for_span(fs.linesize, height);
		if (min_slices == 0)
			return fail(ctx, SCOPE_EINVAL, "pitch * height too large", hipSuccess);
		if (slices < min_slices)
			slices = min_slices;
		ctx->stat_his_rows += scope_his_rows_mode_host(fs, slices) ? 1u : 0u;
		if (slices == 1) {
			TIMED_LAUNCH(ctx, scope_launch_his(fs, 1, chmask, fuse, colorspace, dst, NULL, NULL, ctx->tn, ctx->stream, tm_));
		} else if (ctx->tn.fold) {
			/* the slices of a frame meet in ctx->his_acc inside the launch; no partial tables, no second kernel */
			TIMED_LAUNCH(ctx, scope_launch_his(fs, slices, chmask, fuse, colorspace, dst, ctx->his_acc, ctx->tickets, ctx->tn,
							   ctx->stream, tm_));
		} else {
			st = grow(ctx, &ctx->partial, &ctx->partial_bytes, (size_t)nf * slices * table);
			if (st)
				return st;
			TIMED_LAUNCH(ctx, scope_launch_his(fs, slices, chmask, fuse, colorspace, (uint32_t *)ctx->partial, NULL, NULL,
							   ctx->tn, ctx->stream, tm_));
			HIP_TRY(ctx, scope_launch_sum_reduce((const uint32_t *)ctx->partial, dst, nf, slices, SCOPE_HIS_WORDS,
							     ctx->stream));
		}
		if (!(flags & SCOPE_IN_DEVICE) && f0 + nf < n)
			HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	}
	return result_end(ctx, counts, (size_t)n * table, flags, dev_out);
}

/* A table of zeros, for the cases where the reference zero-fills and returns. */
int zero_result(scope_ctx *ctx, void *out, size_t bytes, uint32_t flags)
{
	if (flags & SCOPE_OUT_DEVICE) {
		HIP_TRY(ctx, hipMemsetAsync(out, 0, bytes, ctx->stream));
		if (!(flags & SCOPE_ASYNC))
			HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	} else {
		memset(out, 0, bytes);
	}
	return SCOPE_OK;
}

/* Every entry point but one fences the side stream first (include/scope_hip.h, SCOPE_SIDE_REDUCE: "the next call on the
 * context without this flag fences first"): whatever it enqueues on the context's stream -- another scope, a table
 * pack / all-reduce, a display pass, a download -- then sees the tables of earlier SCOPE_SIDE_REDUCE calls complete.
 * Only scope_vectorscope_batch carrying the flag itself enters without the fence (SCOPE_ENTER_FLAGS). */
#define SCOPE_ENTER_FLAGS(ctx, flags)                                         \
	if (!(ctx))                                                           \
		return SCOPE_EINVAL;                                          \
	std::lock_guard<std::recursive_mutex> scope_lock_((ctx)->mu);         \
	int st = enter(ctx, flags)
#define SCOPE_ENTER(ctx) SCOPE_ENTER_FLAGS(ctx, 0u)

int enter(scope_ctx *ctx, uint32_t flags = 0u)
{
	if (!ctx)
		return SCOPE_EINVAL;
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	if (!(flags & SCOPE_SIDE_REDUCE))
		return side_fence(ctx);
	return SCOPE_OK;
}

uint32_t display_n_components(uint32_t components)
{
	/* n_components(), src/waveform.c:172-179 */
	uint32_t n = 0;
	for (uint32_t c = components & 0x77u; c; c >>= 1)
		n += c & 1u;
	return n;
}

/* shared plumbing: table in (host or device) -> kernel -> picture out (host or device) */
template <typename Launch>
int display_run(scope_ctx *ctx, const void *table, size_t table_bytes, uint8_t *rgba, size_t out_bytes,
		       uint32_t flags, Launch launch)
{
	SCOPE_ENTER(ctx);
	if (st)
		return st;
	if (!table || !rgba)
		return fail(ctx, SCOPE_EINVAL, "NULL argument", hipSuccess);
	const uint32_t dev = flags & (SCOPE_IN_DEVICE | SCOPE_OUT_DEVICE);
	if (dev != 0 && dev != (SCOPE_IN_DEVICE | SCOPE_OUT_DEVICE))
		return fail(ctx, SCOPE_EINVAL, "display passes take host or device buffers, not a mix", hipSuccess);
	if (out_bytes == 0)
		return SCOPE_OK;
	if (dev) {
		HIP_TRY(ctx, launch((const uint8_t *)table, rgba));
		if (!(flags & SCOPE_ASYNC))
			HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
		return SCOPE_OK;
	}
	st = grow(ctx, &ctx->stage_in, &ctx->stage_in_bytes, round_up(table_bytes, 256));
	if (!st)
		st = grow(ctx, &ctx->stage_out, &ctx->stage_out_bytes, round_up(out_bytes, 256));
	if (st)
		return st;
	HIP_TRY(ctx, hipMemcpyAsync(ctx->stage_in, table, table_bytes, hipMemcpyHostToDevice, ctx->stream));
	HIP_TRY(ctx, launch(ctx->stage_in, ctx->stage_out));
	HIP_TRY(ctx, hipMemcpyAsync(rgba, ctx->stage_out, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return SCOPE_OK;
}

} // namespace

extern "C" {

int scope_abi_version(void)
{
	return SCOPE_ABI_VERSION;
}

const char *scope_strerror(int status)
{
	switch (status) {
	case SCOPE_OK:
		return "ok";
	case SCOPE_EINVAL:
		return "invalid argument";
	case SCOPE_ENOMEM:
		return "out of memory";
	case SCOPE_ENODEV:
		return "no usable HIP device";
	case SCOPE_EIO:
		return "HIP runtime error";
	case SCOPE_ENODATA:
		return "required plane is NULL";
	case SCOPE_EAGAIN:
		return "pipeline full / empty";
	default:
		return "unknown status";
	}
}

const char *scope_last_error(scope_ctx *ctx)
{
	return ctx ? ctx->err : "no context";
}

int scope_ctx_create(int device, scope_ctx **out)
{
	if (!out)
		return SCOPE_EINVAL;
	*out = NULL;
	int count = 0;
	if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count)
		return SCOPE_ENODEV;
	if (hipSetDevice(device) != hipSuccess)
		return SCOPE_ENODEV;
	scope_ctx *ctx = new (std::nothrow) scope_ctx();
	if (!ctx)
		return SCOPE_ENOMEM;
	ctx->device = device;
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device) != hipSuccess) {
		delete ctx;
		return SCOPE_ENODEV;
	}
	ctx->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
	if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) {
		delete ctx;
		return SCOPE_EIO;
	}
	ctx->stream = ctx->own_stream;
	bool ok = scope_kernels_init() == hipSuccess;
	ok = ok && hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking) == hipSuccess;
	for (int k = 0; ok && k < 2; k++)
		ok = hipEventCreateWithFlags(&ctx->acc_done[k], hipEventDisableTiming) == hipSuccess &&
		     hipEventCreateWithFlags(&ctx->side_done[k], hipEventDisableTiming) == hipSuccess;
	const size_t acc_bytes = (size_t)SCOPE_MAX_FRAMES * SCOPE_HIS_WORDS * sizeof(uint32_t);
	/* two sets of arrival tickets: the histogram's in-launch combine, and (behind it) the vectorscope's fold */
	ok = ok && hipMalloc((void **)&ctx->tickets, 2 * SCOPE_MAX_FRAMES * sizeof(uint32_t)) == hipSuccess &&
	     hipMemsetAsync(ctx->tickets, 0, 2 * SCOPE_MAX_FRAMES * sizeof(uint32_t), ctx->stream) == hipSuccess &&
	     hipMalloc((void **)&ctx->his_acc, acc_bytes) == hipSuccess &&
	     hipMemsetAsync(ctx->his_acc, 0, acc_bytes, ctx->stream) == hipSuccess &&
	     hipStreamSynchronize(ctx->stream) == hipSuccess;
	for (uint32_t i = 0; ok && i < 2 * SCOPE_PROFILE_SLOTS; i++)
		ok = hipEventCreate(&ctx->ev[i]) == hipSuccess;
	if (!ok) {
		scope_ctx_destroy(ctx);
		return SCOPE_EIO;
	}
	ctx->tn.vss_merge = 3;
	ctx->tn.vss_unroll = 0;       /* auto: 4 loads per lane per batch for batches, 2 for launches of a few frames */
	ctx->tn.vss_fused_unroll = 0;
	ctx->tn.vss_hash = 0;
	ctx->tn.vss_fused_hash = -1;
	ctx->tn.vss_groups_log2 = -1; /* auto: 8 lane-group streams for batches, one stream for launches of a few frames */
	ctx->tn.his_unroll = 0; /* auto: 1 load per lane per batch, 2 for one-frame launches */
	ctx->tn.wvs_wide = 1;
	ctx->tn.wvs_strip = 0;
	ctx->tn.wvs_deep = 1;
	ctx->tn.dock = 1;
	ctx->tn.wvs_persist = 1;
	ctx->tn.wvs_per_cu = 1;
	ctx->tn.fold = 1;
	ctx->tn.vss_sparse = 1;
	ctx->tn.vss_fold = 1;
	ctx->tn.nt_loads = 1;
	ctx->tn.trio_strip = 0;
	ctx->tn.vss_rle = 0;
	ctx->err[0] = 0;
	*out = ctx;
	return SCOPE_OK;
}

void scope_ctx_destroy(scope_ctx *ctx)
{
	if (!ctx)
		return;
	(void)hipSetDevice(ctx->device);
	(void)hipStreamSynchronize(ctx->stream);
	if (ctx->partial)
		(void)hipFree(ctx->partial);
	if (ctx->stage_in)
		(void)hipFree(ctx->stage_in);
	if (ctx->stage_out)
		(void)hipFree(ctx->stage_out);
	if (ctx->tickets)
		(void)hipFree(ctx->tickets);
	if (ctx->his_acc)
		(void)hipFree(ctx->his_acc);
	if (ctx->pmask)
		(void)hipFree(ctx->pmask);
	if (ctx->side)
		(void)hipStreamSynchronize(ctx->side);
	for (int k = 0; k < 2; k++) {
		if (ctx->side_partial[k])
			(void)hipFree(ctx->side_partial[k]);
		if (ctx->side_pmask[k])
			(void)hipFree(ctx->side_pmask[k]);
		if (ctx->acc_done[k])
			(void)hipEventDestroy(ctx->acc_done[k]);
		if (ctx->side_done[k])
			(void)hipEventDestroy(ctx->side_done[k]);
	}
	if (ctx->side)
		(void)hipStreamDestroy(ctx->side);
	for (int k = 0; k < 2; k++)
		if (ctx->surf.dev[k])
			(void)hipFree(ctx->surf.dev[k]);
	for (uint32_t i = 0; i < 2 * SCOPE_PROFILE_SLOTS; i++)
		if (ctx->ev[i])
			(void)hipEventDestroy(ctx->ev[i]);
	(void)hipStreamDestroy(ctx->own_stream);
	delete ctx;
}

int scope_ctx_set_stream(scope_ctx *ctx, void *hip_stream)
{
	SCOPE_ENTER(ctx);
	if (st)
		return st;
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->side));
	ctx->side_pending = ctx->side_busy[0] = ctx->side_busy[1] = false;
	ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
	return SCOPE_OK;
}

void *scope_ctx_get_stream(scope_ctx *ctx)
{
	return ctx ? (void *)ctx->stream : NULL;
}

int scope_ctx_sync(scope_ctx *ctx)
{
	SCOPE_ENTER(ctx);
	if (st)
		return st;
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	if (ctx->side_pending || ctx->side_busy[0] || ctx->side_busy[1]) {
		HIP_TRY(ctx, hipStreamSynchronize(ctx->side));
		ctx->side_pending = ctx->side_busy[0] = ctx->side_busy[1] = false;
	}
	return SCOPE_OK;
}

int scope_ctx_fence(scope_ctx *ctx)
{
	SCOPE_ENTER(ctx);
	if (st)
		return st;
	return side_fence(ctx);
}

int scope_ctx_device_info(scope_ctx *ctx, struct scope_device_info *info)
{
	SCOPE_ENTER(ctx);
	if (st)
		return st;
	if (!info)
		return SCOPE_EINVAL;
	hipDeviceProp_t prop;
	HIP_TRY(ctx, hipGetDeviceProperties(&prop, ctx->device));
	memset(info, 0, sizeof *info);
	snprintf(info->name, sizeof info->name, "%s", prop.name);
	snprintf(info->arch, sizeof info->arch, "%s", prop.gcnArchName);
	info->device = ctx->device;
	info->compute_units = prop.multiProcessorCount;
	info->lds_bytes_per_cu = (int)prop.maxSharedMemoryPerMultiProcessor;
	info->clock_khz = prop.clockRate;
	info->hbm_bytes = prop.totalGlobalMem;
	return SCOPE_OK;
}

int scope_ctx_set_option(scope_ctx *ctx, const char *key, int value)
{
	if (!ctx || !key)
		return SCOPE_EINVAL;
	std::lock_guard<std::recursive_mutex> lock(ctx->mu);
	if (!strcmp(key, "vss.merge"))
		ctx->tn.vss_merge = value;
	else if (!strcmp(key, "vss.unroll"))
		ctx->tn.vss_unroll = value;
	else if (!strcmp(key, "vss.fused_unroll"))
		ctx->tn.vss_fused_unroll = value;
	else if (!strcmp(key, "vss.block"))
		ctx->tn.vss_block = value;
	else if (!strcmp(key, "his.block"))
		ctx->tn.his_block = value;
	else if (!strcmp(key, "vss.hash"))
		ctx->tn.vss_hash = value;
	else if (!strcmp(key, "vss.fused_hash"))
		ctx->tn.vss_fused_hash = value;
	else if (!strcmp(key, "vss.groups"))
		ctx->tn.vss_groups_log2 = value;
	else if (!strcmp(key, "nt"))
		ctx->tn.nt_loads = value;
	else if (!strcmp(key, "his.unroll"))
		ctx->tn.his_unroll = value;
	else if (!strcmp(key, "fold"))
		ctx->tn.fold = value != 0;
	else if (!strcmp(key, "vss.sparse"))
		ctx->tn.vss_sparse = value != 0;
	else if (!strcmp(key, "vss.fold"))
		ctx->tn.vss_fold = value != 0;
	else if (!strcmp(key, "wvs.per_cu"))
		ctx->tn.wvs_per_cu = value;
	else if (!strcmp(key, "wvs.persist"))
		ctx->tn.wvs_persist = value != 0;
	else if (!strcmp(key, "wvs.wide"))
		ctx->tn.wvs_wide = value != 0;
	else if (!strcmp(key, "wvs.strip"))
		ctx->tn.wvs_strip = value;
	else if (!strcmp(key, "wvs.deep"))
		ctx->tn.wvs_deep = value != 0;
	else if (!strcmp(key, "dock"))
		ctx->tn.dock = value < 0 ? 0 : (value > 2 ? 1 : value);
	else if (!strcmp(key, "trio.strip"))
		ctx->tn.trio_strip = value == 16 || value == 32 ? value : 0;
	else if (!strcmp(key, "vss.rle"))
		ctx->tn.vss_rle = value != 0;
	else if (!strcmp(key, "slices"))
		ctx->slices_override = value;
	else
		return fail(ctx, SCOPE_EINVAL, key, hipSuccess);
	return SCOPE_OK;
}

int scope_ctx_get_stat(scope_ctx *ctx, const char *key, uint64_t *value)
{
	if (!ctx || !key || !value)
		return SCOPE_EINVAL;
	std::lock_guard<std::recursive_mutex> lock(ctx->mu);
	if (!strcmp(key, "vss.rows_launches"))
		*value = ctx->stat_vss_rows;
	else if (!strcmp(key, "his.rows_launches"))
		*value = ctx->stat_his_rows;
	else if (!strcmp(key, "vss.fold_launches"))
		*value = ctx->stat_vss_fold;
	else if (!strcmp(key, "trio.fold_launches"))
		*value = ctx->stat_trio_fold;
	else if (!strcmp(key, "trio.chroma_guard_selftest"))
		*value = scope_trio_chroma_selftest();
	else
		return fail(ctx, SCOPE_EINVAL, key, hipSuccess);
	return SCOPE_OK;
}

int scope_convert_rgb_yuv(scope_ctx *ctx, const uint8_t *rgb, uint32_t rgb_linesize, uint8_t *yuv,
			  uint32_t yuv_linesize, uint32_t width, uint32_t height, int colorspace, uint32_t flags)
{
	SCOPE_ENTER(ctx);
	if (st)
		return st;
	if (!rgb || !yuv)
		return fail(ctx, SCOPE_EINVAL, "NULL plane", hipSuccess);
	st = check_geometry(ctx, rgb_linesize, width, height);
	if (!st)
		st = check_geometry(ctx, yuv_linesize, width, height);
	if (st)
		return st;
	if (width == 0 || height == 0)
		return SCOPE_OK;
	if (flags & SCOPE_IN_DEVICE) {
		if (((uintptr_t)rgb | (uintptr_t)yuv | rgb_linesize | yuv_linesize) & 3u)
			return fail(ctx, SCOPE_EINVAL, "device planes must be 4-byte aligned", hipSuccess);
		HIP_TRY(ctx, scope_launch_k0(rgb, rgb_linesize, yuv, yuv_linesize, width, height, colorspace, ctx->stream));
		if (!(flags & SCOPE_ASYNC))
			HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
		return SCOPE_OK;
	}
	ScopeFrameSet fs;
	const uint8_t *one[1] = {rgb};
	st = stage_planes(ctx, one, 1, rgb_linesize, width, height, 0, &fs);
	if (st)
		return st;
	const size_t bytes = (size_t)fs.linesize * height;
	st = grow(ctx, &ctx->stage_out, &ctx->stage_out_bytes, round_up(bytes, 256));
	if (st)
		return st;
	HIP_TRY(ctx, scope_launch_k0(fs.plane[0], fs.linesize, ctx->stage_out, fs.linesize, width, height, colorspace,
				     ctx->stream));
	HIP_TRY(ctx, hipMemcpy2DAsync(yuv, yuv_linesize, ctx->stage_out, fs.linesize, (size_t)width * 4, height,
				      hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return SCOPE_OK;
}

int scope_region_size(uint32_t width, uint32_t height, const struct scope_region *region, uint32_t *out_width,
		      uint32_t *out_height)
{
	uint32_t s, x, y, cx, cy;
	if (!region || !out_width || !out_height || region->scale == 0)
		return SCOPE_EINVAL;
	const int st = resolve_region(width, height, region, &s, &x, &y, &cx, &cy);
	if (st)
		return st;
	*out_width = cx;
	*out_height = cy;
	return SCOPE_OK;
}

int scope_stage_region(scope_ctx *ctx, const uint8_t *src, uint32_t src_linesize, uint32_t width, uint32_t height,
		       const struct scope_region *region, uint8_t *dst, uint32_t dst_linesize, uint32_t flags)
{
	SCOPE_ENTER(ctx);
	if (st)
		return st;
	if (!src || !dst || !region || region->scale == 0)
		return fail(ctx, SCOPE_EINVAL, "NULL argument or scale 0", hipSuccess);
	st = check_geometry(ctx, src_linesize, width, height);
	if (st)
		return st;
	uint32_t s, x, y, cx, cy;
	if (resolve_region(width, height, region, &s, &x, &y, &cx, &cy))
		return fail(ctx, SCOPE_EINVAL, "region outside the scaled frame", hipSuccess);
	if ((uint64_t)cx * 4 > dst_linesize)
		return fail(ctx, SCOPE_EINVAL, "dst_linesize < 4*cx", hipSuccess);
	const size_t first_row = (size_t)y * s + s / 2, first_col = (size_t)x * s + s / 2;
	if (flags & SCOPE_IN_DEVICE) {
		if (((uintptr_t)src | (uintptr_t)dst | src_linesize | dst_linesize) & 3u)
			return fail(ctx, SCOPE_EINVAL, "device planes must be 4-byte aligned", hipSuccess);
		HIP_TRY(ctx, scope_launch_stage(src + first_row * src_linesize + first_col * 4, (uint64_t)s * src_linesize, s, cx, cy,
						dst, dst_linesize, ctx->stream));
		if (!(flags & SCOPE_ASYNC))
			HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
		return SCOPE_OK;
	}
	/* host planes: only the sampled rows cross PCIe, and of those only the span of sampled columns */
	const size_t span = ((size_t)(cx - 1) * s + 1) * 4;
	const size_t pitch = round_up(span, 16);
	const size_t out_pitch = round_up((size_t)cx * 4, 16);
	st = grow(ctx, &ctx->stage_in, &ctx->stage_in_bytes, round_up(pitch * cy, 256));
	if (!st)
		st = grow(ctx, &ctx->stage_out, &ctx->stage_out_bytes, round_up(out_pitch * cy, 256));
	if (st)
		return st;
	HIP_TRY(ctx, hipMemcpy2DAsync(ctx->stage_in, pitch, src + first_row * src_linesize + first_col * 4,
				      (size_t)s * src_linesize, span, cy, hipMemcpyHostToDevice, ctx->stream));
	HIP_TRY(ctx, scope_launch_stage(ctx->stage_in, pitch, s, cx, cy, ctx->stage_out, (uint32_t)out_pitch, ctx->stream));
	HIP_TRY(ctx, hipMemcpy2DAsync(dst, dst_linesize, ctx->stage_out, out_pitch, (size_t)cx * 4, cy, hipMemcpyDeviceToHost,
				      ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return SCOPE_OK;
}

int scope_surface_begin(scope_ctx *ctx, uint32_t flags)
{
	if (!ctx)
		return SCOPE_EINVAL;
	ctx->mu.lock(); /* held until scope_surface_end(): the fan-out of one surface is one critical section */
	int st = enter(ctx);
	if (!st && ctx->surf.active)
		st = fail(ctx, SCOPE_EINVAL, "scope_surface_begin without scope_surface_end", hipSuccess);
	if (!st && (flags & ~(uint32_t)SCOPE_FUSE_K0))
		st = fail(ctx, SCOPE_EINVAL, "scope_surface_begin takes SCOPE_FUSE_K0 or 0", hipSuccess);
	if (st) {
		ctx->mu.unlock();
		return st;
	}
	ctx->surf.active = true;
	ctx->surf.fuse_k0 = (flags & SCOPE_FUSE_K0) != 0;
	ctx->surf.host[0] = ctx->surf.host[1] = NULL;
	ctx->surf.uploads = ctx->surf.hits = 0;
	ctx->surf.his_valid = false;
	return SCOPE_OK;
}

int scope_surface_end(scope_ctx *ctx, uint32_t *uploads, uint32_t *reuses)
{
	if (!ctx)
		return SCOPE_EINVAL;
	std::unique_lock<std::recursive_mutex> probe(ctx->mu); /* recursive: succeeds for the owner of a begin */
	if (!ctx->surf.active)
		return SCOPE_EINVAL;
	if (uploads)
		*uploads = ctx->surf.uploads;
	if (reuses)
		*reuses = ctx->surf.hits;
	ctx->surf.active = false;
	ctx->surf.his_valid = false;
	ctx->surf.host[0] = ctx->surf.host[1] = NULL;
	/* every call in between was synchronous (host planes), so nothing is in flight on the cached copies */
	probe.unlock();
	ctx->mu.unlock(); /* the lock scope_surface_begin() took */
	return SCOPE_OK;
}

int scope_vectorscope(scope_ctx *ctx, const struct scope_surface_data *sd, uint32_t flags, uint8_t *out)
{
	SCOPE_ENTER(ctx);
	if (st)
		return st;
	if (!sd || !out)
		return fail(ctx, SCOPE_EINVAL, "NULL argument", hipSuccess);
	const uint8_t *plane = sd->yuv_data;
	int fuse = 0;
	if ((!plane && (flags & SCOPE_FUSE_K0)) || surface_prefers_fused(ctx, sd)) {
		plane = sd->rgb_data;
		fuse = 1;
	}
	if (!plane)
		return SCOPE_ENODATA; /* vss_surface_cb returns without drawing, src/vectorscope.c:252 */
	if (sd->width == 0 || sd->height == 0)
		return zero_result(ctx, out, SCOPE_VSS_BINS, flags);
	const uint8_t *one[1] = {plane};
	return run_vectorscope(ctx, one, 1, sd->linesize, sd->width, sd->height, fuse, sd->colorspace, flags, out);
}

int scope_vectorscope_batch(scope_ctx *ctx, const struct scope_batch *b, uint32_t flags, uint8_t *out)
{
	SCOPE_ENTER_FLAGS(ctx, flags);
	if (st)
		return st;
	if (!b || !out || !b->planes || b->n_frames == 0)
		return fail(ctx, SCOPE_EINVAL, "NULL or empty batch", hipSuccess);
	for (uint32_t i = 0; i < b->n_frames; i++)
		if (!b->planes[i])
			return SCOPE_ENODATA;
	if (b->width == 0 || b->height == 0)
		return zero_result(ctx, out, (size_t)b->n_frames * SCOPE_VSS_BINS * ((flags & SCOPE_OUT_PACKED16) ? 2u : 1u), flags);
	return run_vectorscope(ctx, b->planes, b->n_frames, b->linesize, b->width, b->height, b->planes_are_rgb != 0,
			       b->colorspace, flags, out);
}

int scope_waveform(scope_ctx *ctx, const struct scope_surface_data *sd, uint32_t components, uint32_t flags,
		   uint8_t *out)
{
	SCOPE_ENTER(ctx);
	if (st)
		return st;
	if (!sd || !out)
		return fail(ctx, SCOPE_EINVAL, "NULL argument", hipSuccess);
	const size_t table = (size_t)SCOPE_WV_SIZE * sd->width * 4;
	const uint8_t *plane = NULL;
	int fuse = 0;
	const int sel = pick_plane(ctx, sd, components, flags, &plane, &fuse);
	if (sel < 0)
		return sel; /* wvs_surface_cb returns without drawing, src/waveform.c:276-281 */
	if (sel == 0 || sd->height == 0 || sd->width == 0)
		return table ? zero_result(ctx, out, table, flags) : SCOPE_OK; /* src/waveform.c:225-234 */
	const uint8_t *one[1] = {plane};
	if (ctx->surf.active && !(flags & (SCOPE_IN_DEVICE | SCOPE_OUT_DEVICE))) {
		/* the dock: the same surface goes to the histogram next (src/roi.c:329-341); its counts cost this pass nothing */
		ctx->surf.his_valid = false;
		st = run_waveform(ctx, one, 1, sd->linesize, sd->width, sd->height, channel_mask(components), fuse, sd->colorspace,
				  flags, out, ctx->surf.his_counts, channel_mask(components));
		if (st == SCOPE_OK) {
			ctx->surf.his_valid = true;
			ctx->surf.his_plane = plane;
			ctx->surf.his_fuse = fuse;
			ctx->surf.his_chmask = channel_mask(components);
			ctx->surf.his_linesize = sd->linesize;
			ctx->surf.his_width = sd->width;
			ctx->surf.his_height = sd->height;
			ctx->surf.his_colorspace = sd->colorspace;
		}
		return st;
	}
	return run_waveform(ctx, one, 1, sd->linesize, sd->width, sd->height, channel_mask(components), fuse,
			    sd->colorspace, flags, out);
}

int scope_waveform_batch(scope_ctx *ctx, const struct scope_batch *b, uint32_t components, uint32_t flags,
			 uint8_t *out)
{
	SCOPE_ENTER(ctx);
	if (st)
		return st;
	if (!b || !out || !b->planes || b->n_frames == 0)
		return fail(ctx, SCOPE_EINVAL, "NULL or empty batch", hipSuccess);
	for (uint32_t i = 0; i < b->n_frames; i++)
		if (!b->planes[i])
			return SCOPE_ENODATA;
	const size_t table = (size_t)SCOPE_WV_SIZE * b->width * 4;
	/* which plane kind the mask wants vs. what the batch holds */
	const bool want_rgb = (components & 0x07u) != 0, want_yuv = !want_rgb && (components & 0x70u) != 0;
	if ((!want_rgb && !want_yuv) || b->height == 0 || b->width == 0)
		return table ? zero_result(ctx, out, (size_t)b->n_frames * table, flags) : SCOPE_OK;
	if (want_rgb && !b->planes_are_rgb)
		return SCOPE_ENODATA;
	const int fuse = want_yuv && b->planes_are_rgb;
	return run_waveform(ctx, b->planes, b->n_frames, b->linesize, b->width, b->height, channel_mask(components), fuse,
			    b->colorspace, flags, out);
}

int scope_histogram_counts(scope_ctx *ctx, const struct scope_surface_data *sd, uint32_t components, uint32_t flags,
			   uint32_t *counts)
{
	SCOPE_ENTER(ctx);
	if (st)
		return st;
	if (!sd || !counts)
		return fail(ctx, SCOPE_EINVAL, "NULL argument", hipSuccess);
	const uint8_t *plane = NULL;
	int fuse = 0;
	const int sel = pick_plane(ctx, sd, components, flags, &plane, &fuse);
	if (sel < 0)
		return sel; /* his_surface_cb returns without drawing, src/histogram.c:436-441 */
	if (sel == 0 || sd->height == 0 || sd->width == 0)
		return zero_result(ctx, counts, SCOPE_HIS_WORDS * 4, flags); /* src/histogram.c:363-373 */
	const uint32_t chmask = channel_mask(components);
	if (ctx->surf.active && ctx->surf.his_valid && !(flags & (SCOPE_IN_DEVICE | SCOPE_OUT_DEVICE)) &&
	    ctx->surf.his_plane == plane && ctx->surf.his_fuse == fuse && (chmask & ~ctx->surf.his_chmask) == 0 &&
	    /* same pointer, other geometry or matrix: not the pass that left these counts -- fall through to run_histogram,
	     * where stage_planes refuses a geometry change inside scope_surface_begin/end with SCOPE_EINVAL */
	    ctx->surf.his_linesize == sd->linesize && ctx->surf.his_width == sd->width && ctx->surf.his_height == sd->height &&
	    (!fuse || ctx->surf.his_colorspace == sd->colorspace)) {
		for (uint32_t t = 0; t < SCOPE_HIS_WORDS; t++) {
			const uint32_t slot = t & 3u; /* slot 0 = R = byte 2, 1 = G, 2 = B */
			counts[t] = (slot < 3 && ((chmask >> (2u - slot)) & 1u)) ? ctx->surf.his_counts[t] : 0u;
		}
		ctx->surf.hits++;
		return SCOPE_OK;
	}
	const uint8_t *one[1] = {plane};
	return run_histogram(ctx, one, 1, sd->linesize, sd->width, sd->height, chmask, fuse, sd->colorspace, flags, counts);
}

int scope_histogram_counts_batch(scope_ctx *ctx, const struct scope_batch *b, uint32_t components, uint32_t flags,
				 uint32_t *counts)
{
	SCOPE_ENTER(ctx);
	if (st)
		return st;
	if (!b || !counts || !b->planes || b->n_frames == 0)
		return fail(ctx, SCOPE_EINVAL, "NULL or empty batch", hipSuccess);
	for (uint32_t i = 0; i < b->n_frames; i++)
		if (!b->planes[i])
			return SCOPE_ENODATA;
	const bool want_rgb = (components & 0x07u) != 0, want_yuv = !want_rgb && (components & 0x70u) != 0;
	if ((!want_rgb && !want_yuv) || b->height == 0 || b->width == 0)
		return zero_result(ctx, counts, (size_t)b->n_frames * SCOPE_HIS_WORDS * 4, flags);
	if (want_rgb && !b->planes_are_rgb)
		return SCOPE_ENODATA;
	const int fuse = want_yuv && b->planes_are_rgb;
	return run_histogram(ctx, b->planes, b->n_frames, b->linesize, b->width, b->height, channel_mask(components), fuse,
			     b->colorspace, flags, counts);
}

int scope_waveform_histogram(scope_ctx *ctx, const struct scope_surface_data *sd, uint32_t wvs_components,
			     uint32_t his_components, uint32_t flags, uint8_t *wvs_out, uint32_t *his_counts)
{
	SCOPE_ENTER(ctx);
	if (st)
		return st;
	if (!sd || !wvs_out || !his_counts)
		return fail(ctx, SCOPE_EINVAL, "NULL argument", hipSuccess);
	const uint8_t *pw = NULL, *ph = NULL;
	int fw = 0, fh = 0;
	const int sw = pick_plane(ctx, sd, wvs_components, flags, &pw, &fw);
	const int sh = pick_plane(ctx, sd, his_components, flags, &ph, &fh);
	const uint32_t cw = channel_mask(wvs_components), ch = channel_mask(his_components);
	if (sw == 1 && sh == 1 && pw == ph && fw == fh && (ch & ~cw) == 0 && sd->width && sd->height) {
		const uint8_t *one[1] = {pw};
		return run_waveform(ctx, one, 1, sd->linesize, sd->width, sd->height, cw, fw, sd->colorspace, flags, wvs_out,
				    his_counts, ch);
	}
	/* different planes or channels the waveform does not count: two passes, same results */
	st = scope_waveform(ctx, sd, wvs_components, flags, wvs_out);
	if (st)
		return st;
	return scope_histogram_counts(ctx, sd, his_components, flags, his_counts);
}

int scope_waveform_histogram_batch(scope_ctx *ctx, const struct scope_batch *b, uint32_t wvs_components,
				   uint32_t his_components, uint32_t flags, uint8_t *wvs_out, uint32_t *his_counts)
{
	SCOPE_ENTER(ctx);
	if (st)
		return st;
	if (!b || !wvs_out || !his_counts || !b->planes || b->n_frames == 0)
		return fail(ctx, SCOPE_EINVAL, "NULL or empty batch", hipSuccess);
	for (uint32_t i = 0; i < b->n_frames; i++)
		if (!b->planes[i])
			return SCOPE_ENODATA;
	const bool w_rgb = (wvs_components & 0x07u) != 0, w_yuv = !w_rgb && (wvs_components & 0x70u) != 0;
	const bool h_rgb = (his_components & 0x07u) != 0, h_yuv = !h_rgb && (his_components & 0x70u) != 0;
	const uint32_t cw = channel_mask(wvs_components), ch = channel_mask(his_components);
	const bool same_kind = (w_rgb && h_rgb && b->planes_are_rgb) || (w_yuv && h_yuv);
	if (same_kind && (ch & ~cw) == 0 && b->width && b->height)
		return run_waveform(ctx, b->planes, b->n_frames, b->linesize, b->width, b->height, cw, w_yuv && b->planes_are_rgb,
				    b->colorspace, flags, wvs_out, his_counts, ch);
	st = scope_waveform_batch(ctx, b, wvs_components, flags, wvs_out);
	if (st)
		return st;
	return scope_histogram_counts_batch(ctx, b, his_components, flags, his_counts);
}

int scope_three_scopes_batch(scope_ctx *ctx, const struct scope_batch *b, uint32_t wvs_components, uint32_t his_components,
			     uint32_t flags, uint8_t *vss_out, uint8_t *wvs_out, uint32_t *his_counts)
{
	SCOPE_ENTER(ctx);
	if (st)
		return st;
	if (!b || !vss_out || !wvs_out || !his_counts || !b->planes || b->n_frames == 0)
		return fail(ctx, SCOPE_EINVAL, "NULL or empty batch", hipSuccess);
	if (!b->planes_are_rgb)
		return fail(ctx, SCOPE_EINVAL, "scope_three_scopes_batch takes RGB planes (the vectorscope goes through the fused K0)", hipSuccess);
	if (flags & SCOPE_OUT_PACKED16)
		return fail(ctx, SCOPE_EINVAL, "SCOPE_OUT_PACKED16 is a vectorscope-only flag", hipSuccess);
	for (uint32_t i = 0; i < b->n_frames; i++)
		if (!b->planes[i])
			return SCOPE_ENODATA;
	/* One read for all three: device-resident planes and tables, RGB channels for waveform and histogram (the histogram a
	 * subset of the waveform's), and a launch small enough for one round of 16-column strips -- the dock's one frame. */
	if ((wvs_components & 0x07u) != 0 && (his_components & 0x07u) != 0) {
		const int done = run_dock(ctx, b->planes, b->n_frames, b->linesize, b->width, b->height, channel_mask(wvs_components),
					  channel_mask(his_components), b->colorspace, flags, vss_out, wvs_out, his_counts);
		if (done)
			return done < 0 ? done : SCOPE_OK;
	}
	/* everything else: the vectorscope's pass, then the waveform's with the histogram riding on it */
	st = scope_vectorscope_batch(ctx, b, flags, vss_out);
	if (st)
		return st;
	return scope_waveform_histogram_batch(ctx, b, wvs_components, his_components, flags, wvs_out, his_counts);
}

void scope_histogram_level(uint8_t *tex_buf, uint32_t *hi_max, uint32_t width, uint32_t height, uint32_t components,
			   int logscale, int level_fixed_value, int level_ratio_value)
{
	uint32_t n[SCOPE_HIS_WORDS];
	float f[SCOPE_HIS_WORDS];
	memcpy(n, tex_buf, sizeof n);

	/* level: fixed pixel count, else per-mille of the frame, else per-channel peak
	 * (src/histogram.c:397-402 with :330-355); slot j is enabled by mask 0x44 >> j */
	uint32_t level = 0;
	if (level_fixed_value > 0)
		level = (uint32_t)level_fixed_value;
	else if (level_ratio_value > 0)
		level = (uint32_t)((uint64_t)width * height * (uint64_t)level_ratio_value / 1000u);
	if (level_fixed_value > 0 || level_ratio_value > 0) {
		for (int j = 0; j < 3; j++)
			hi_max[j] = level ? level : 1u;
	} else {
		for (int j = 0; j < 3; j++) {
			uint32_t peak = 1;
			if (components & (0x44u >> j))
				for (int v = 0; v < SCOPE_HI_SIZE; v++)
					peak = n[4 * v + j] > peak ? n[4 * v + j] : peak;
			hi_max[j] = peak;
		}
	}

	/* float conversion in place (src/histogram.c:404-417) */
	memcpy(f, n, sizeof f);
	if (logscale) {
		for (int j = 0; j < 3; j++) {
			if (!(components & (0x44u >> j)))
				continue;
			const float scale = 1.0f / logf((float)(hi_max[j] + 1));
			for (int v = 0; v < SCOPE_HI_SIZE; v++)
				f[4 * v + j] = n[4 * v + j] ? logf((float)(n[4 * v + j] + 1)) * scale : 0.0f;
			hi_max[j] = 1;
		}
	} else {
		for (uint32_t i = 0; i < SCOPE_HIS_WORDS; i++)
			f[i] = (float)n[i];
	}
	memcpy(tex_buf, f, sizeof f);
}

int scope_histogram(scope_ctx *ctx, const struct scope_surface_data *sd, uint32_t components, int logscale,
		    int level_fixed_value, int level_ratio_value, uint32_t flags, uint8_t *tex_buf, uint32_t *hi_max)
{
	if (!ctx || !sd || !tex_buf || !hi_max || (flags & (SCOPE_OUT_DEVICE | SCOPE_ASYNC)))
		return SCOPE_EINVAL;
	uint32_t counts[SCOPE_HIS_WORDS];
	const int st = scope_histogram_counts(ctx, sd, components, flags, counts);
	if (st)
		return st;
	memcpy(tex_buf, counts, sizeof counts);
	/* no selectable plane: the reference returns before the level pass (src/histogram.c:372-373) */
	if (!(components & 0x77u))
		return SCOPE_OK;
	scope_histogram_level(tex_buf, hi_max, sd->width, sd->height, components, logscale, level_fixed_value,
			      level_ratio_value);
	return SCOPE_OK;
}

int scope_table_widen_u8_u32(scope_ctx *ctx, const uint8_t *in, uint32_t *out, uint64_t n)
{
	SCOPE_ENTER(ctx);
	if (st)
		return st;
	if (!in || !out)
		return SCOPE_EINVAL;
	HIP_TRY(ctx, scope_launch_widen(in, out, n, ctx->stream));
	return SCOPE_OK;
}

int scope_table_clamp_u32_u8(scope_ctx *ctx, const uint32_t *in, uint8_t *out, uint64_t n)
{
	SCOPE_ENTER(ctx);
	if (st)
		return st;
	if (!in || !out)
		return SCOPE_EINVAL;
	HIP_TRY(ctx, scope_launch_clamp(in, out, n, ctx->stream));
	return SCOPE_OK;
}

int scope_table_pack_u8_u16x2(scope_ctx *ctx, const uint8_t *in, uint32_t *out, uint64_t n_pairs)
{
	SCOPE_ENTER(ctx);
	if (st)
		return st;
	if (!in || !out)
		return SCOPE_EINVAL;
	HIP_TRY(ctx, scope_launch_pack16(in, out, n_pairs, ctx->stream));
	return SCOPE_OK;
}

int scope_table_unpack_clamp_u16x2_u8(scope_ctx *ctx, const uint32_t *in, uint8_t *out, uint64_t n_pairs)
{
	SCOPE_ENTER(ctx);
	if (st)
		return st;
	if (!in || !out)
		return SCOPE_EINVAL;
	HIP_TRY(ctx, scope_launch_unpack16_clamp(in, out, n_pairs, ctx->stream));
	return SCOPE_OK;
}

/* ---- frame pools: WHICH ADDRESS RANGE a long-lived pool lies at decides a few per cent of every streaming kernel's rate ------------
 * Round 6's probe (tools/placement_vmm_probe.hip, profiles/round6/placement_vmm_probe_c6.txt): on a box whose pools spread over 8 %
 * the rate FOLLOWS THE VIRTUAL ADDRESS RANGE, not the physical memory -- two hipMemCreate handles that swap their addresses swap
 * their rates (0.856 <-> 0.809 of peak), an offset of 2 or 64 MiB inside a range changes nothing, the latency of dependent loads
 * is the same on fast and slow ranges at every stride, one 24 GiB mapping reads the same in every 2 GiB window.  What a search
 * needs, then, is candidates at DIFFERENT ADDRESS RANGES -- which plain allocations held at the same time are: `candidates`
 * hipMalloc'ed pools, each read by the vectorscope accumulation kernel in THREE INTERLEAVED ROUNDS (the kernel's own begin / end
 * stamps, the fastest launch of a round counts, the median round decides: a ranking by one pass each, rounds 3-5's, is within
 * the noise of a box whose ranges read alike), the best kept, the rest freed.  candidates x bytes of HBM are held for a few
 * milliseconds.
 * Two forms on HIP's virtual-memory calls -- ONE physical allocation tried at several ranges, nothing held twice -- were built in
 * round 6 and withdrawn: mapped at one range at a time and mapped anew at the best, the caller's first WRITE into the pool
 * right after that hipMemMap took a GPU memory access fault in one call of five (tools/pool_stress.py; the default bench line
 * aborted once in an evidence visit); aliased at all ranges at once, with the losers unmapped at the end, it never faulted but the
 * surviving mapping read 3.5 % slower than a plain allocation (0.81 against 0.84 in three of three runs,
 * profiles/round6/pool_search_s6_aliased.txt).  The virtual-memory calls stay in the probe tool. */
namespace {

struct PoolProbe {
	ScopeFrameSet fs;
	uint64_t frame_bytes, probe_bytes;
	uint32_t slices, mask_block;
};

int pool_probe_setup(scope_ctx *ctx, uint64_t bytes, PoolProbe *pp)
{
	memset(&pp->fs, 0, sizeof pp->fs);
	pp->fs.linesize = 4096;
	pp->fs.width = 1024;
	pp->fs.n_frames = bytes >= ((uint64_t)SCOPE_MAX_FRAMES << 20) ? SCOPE_MAX_FRAMES : (uint32_t)((bytes >> 20) ? (bytes >> 20) : 1);
	pp->frame_bytes = (bytes / pp->fs.n_frames) & ~(uint64_t)4095;
	if (pp->frame_bytes >= (1ull << 31))
		pp->frame_bytes = (1ull << 31) - 4096; /* 32-bit offsets inside a slice's span: a huge pool is probed on its first 128 GiB */
	pp->fs.height = (uint32_t)(pp->frame_bytes / 4096);
	pp->probe_bytes = (uint64_t)pp->fs.n_frames * pp->fs.height * 4096;
	pp->slices = pick_slices(ctx, pp->fs.n_frames, 1);
	pp->mask_block = ctx->tn.vss_sparse ? scope_vss_mask_block(ctx->tn) : 0u;
	if (!pp->probe_bytes)
		return SCOPE_OK;
	int rc = grow(ctx, &ctx->partial, &ctx->partial_bytes, (size_t)pp->fs.n_frames * pp->slices * SCOPE_VSS_BINS);
	if (rc == SCOPE_OK && pp->mask_block)
		rc = grow(ctx, &ctx->pmask, &ctx->pmask_bytes, (size_t)pp->fs.n_frames * pp->slices * pp->mask_block);
	return rc;
}

/* GB/s of the vectorscope accumulation kernel over the pool at `base` taken as frames: `launches` launches, the fastest counts */
int pool_probe(scope_ctx *ctx, PoolProbe *pp, const void *base, hipEvent_t ev[2], int warm, int launches, float *gbps)
{
	for (uint32_t f = 0; f < pp->fs.n_frames; f++)
		pp->fs.plane[f] = (const uint8_t *)base + (size_t)f * pp->frame_bytes;
	const ScopeTiming tm = {ev[0], ev[1]};
	float best_ms = 0.0f;
	for (int rep = 0; rep < warm + launches; rep++) {
		if (scope_launch_vss(pp->fs, pp->slices, 0, 2, ctx->partial, (pp->slices > 1 && pp->mask_block) ? ctx->pmask : NULL, ctx->tn,
				     ctx->stream, &tm) != hipSuccess ||
		    hipEventSynchronize(ev[1]) != hipSuccess)
			return fail(ctx, SCOPE_EIO, "frame pool probe", hipGetLastError());
		float ms = 0.0f;
		(void)hipEventElapsedTime(&ms, ev[0], ev[1]);
		if (rep >= warm && ms > 0.0f && (best_ms == 0.0f || ms < best_ms))
			best_ms = ms;
	}
	*gbps = best_ms > 0.0f ? (float)((double)pp->probe_bytes / ((double)best_ms * 1e-3) / 1e9) : 0.0f;
	return SCOPE_OK;
}

float median3(float a, float b, float c)
{
	return a > b ? (b > c ? b : (a > c ? c : a)) : (a > c ? a : (b > c ? c : b));
}

} // namespace

int scope_frame_pool_alloc(scope_ctx *ctx, uint64_t bytes, uint32_t candidates, void **pool, float *read_gbps,
			   uint32_t *tried)
{
	SCOPE_ENTER(ctx);
	if (st)
		return st;
	if (!pool || bytes == 0)
		return fail(ctx, SCOPE_EINVAL, "scope_frame_pool_alloc: NULL pool or zero bytes", hipSuccess);
	*pool = NULL;
	if (tried)
		*tried = 0;
	if (candidates == 0)
		candidates = 1;
	if (candidates > SCOPE_POOL_MAX_CANDIDATES)
		candidates = SCOPE_POOL_MAX_CANDIDATES;
	void *cand[SCOPE_POOL_MAX_CANDIDATES] = {NULL};
	float rate[SCOPE_POOL_MAX_CANDIDATES] = {0}, r[3][SCOPE_POOL_MAX_CANDIDATES] = {{0}};
	hipEvent_t ev[2] = {NULL, NULL};
	PoolProbe pp;
	memset(&pp, 0, sizeof pp);
	int rc = candidates > 1 ? pool_probe_setup(ctx, bytes, &pp) : SCOPE_OK;
	if (rc == SCOPE_OK && candidates > 1 && (hipEventCreate(&ev[0]) != hipSuccess || hipEventCreate(&ev[1]) != hipSuccess))
		rc = fail(ctx, SCOPE_EIO, "hipEventCreate", hipGetLastError());
	/* all candidates are held at once: freeing one and allocating again hands out the same address range again */
	uint32_t n = 0, best = 0;
	for (; rc == SCOPE_OK && n < candidates; n++) {
		const hipError_t e = hipMalloc(&cand[n], (size_t)bytes);
		if (e != hipSuccess) {
			cand[n] = NULL;
			(void)hipGetLastError(); /* the launchers end with hipGetLastError(): a stale out-of-memory must not fail the next one */
			if (n == 0)
				rc = fail(ctx, SCOPE_ENOMEM, "hipMalloc (frame pool)", e);
			break; /* fewer candidates than asked for: the best of those there are */
		}
		/* a known content in every candidate (what fresh memory holds is anybody's, and the kernel's rate depends on it) */
		if (candidates > 1 && pp.probe_bytes && hipMemsetAsync(cand[n], 0x5a, (size_t)bytes, ctx->stream) != hipSuccess)
			rc = fail(ctx, SCOPE_EIO, "hipMemsetAsync (frame pool)", hipGetLastError());
	}
	if (rc == SCOPE_OK && n > 1 && pp.probe_bytes) {
		for (int round = 0; round < 3 && rc == SCOPE_OK; round++)
			for (uint32_t i = 0; i < n && rc == SCOPE_OK; i++)
				rc = pool_probe(ctx, &pp, cand[i], ev, round == 0 ? 2 : 1, 4, &r[round][i]);
		for (uint32_t i = 0; i < n; i++) {
			rate[i] = median3(r[0][i], r[1][i], r[2][i]);
			if (rate[i] > rate[best])
				best = i;
		}
	}
	(void)hipStreamSynchronize(ctx->stream);
	if (rc == SCOPE_OK) {
		*pool = cand[best];
		cand[best] = NULL;
		if (tried)
			*tried = n;
		if (read_gbps) {
			/* the kept one first, then the others in the order they were tried */
			read_gbps[0] = rate[best];
			for (uint32_t i = 0, k = 1; i < n; i++)
				if (i != best)
					read_gbps[k++] = rate[i];
		}
	}
	for (uint32_t i = 0; i < SCOPE_POOL_MAX_CANDIDATES; i++)
		if (cand[i])
			(void)hipFree(cand[i]);
	if (ev[0])
		(void)hipEventDestroy(ev[0]);
	if (ev[1])
		(void)hipEventDestroy(ev[1]);
	return rc;
}

int scope_frame_pool_free(scope_ctx *ctx, void *pool)
{
	SCOPE_ENTER(ctx);
	if (st)
		return st;
	if (pool && hipFree(pool) != hipSuccess)
		return fail(ctx, SCOPE_EIO, "hipFree (frame pool)", hipGetLastError());
	return SCOPE_OK;
}

int scope_ctx_profile(scope_ctx *ctx, int enable)
{
	if (!ctx)
		return SCOPE_EINVAL;
	std::lock_guard<std::recursive_mutex> lock(ctx->mu);
	ctx->profile = enable != 0;
	ctx->timed = 0;
	return SCOPE_OK;
}

int scope_ctx_kernel_ms(scope_ctx *ctx, uint32_t *launches, float *total_ms)
{
	SCOPE_ENTER(ctx);
	if (st)
		return st;
	if (!launches || !total_ms)
		return fail(ctx, SCOPE_EINVAL, "NULL argument", hipSuccess);
	const uint32_t n = ctx->timed < SCOPE_PROFILE_SLOTS ? ctx->timed : SCOPE_PROFILE_SLOTS;
	float sum = 0.0f;
	for (uint32_t i = 0; i < n; i++) {
		float ms = 0.0f;
		HIP_TRY(ctx, hipEventSynchronize(ctx->ev[2 * i + 1]));
		HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev[2 * i], ctx->ev[2 * i + 1]));
		sum += ms;
	}
	*launches = n;
	*total_ms = sum;
	return SCOPE_OK;
}

/* ---- D1-D3 display post-passes ------------------------------------------------------------------ */

void scope_display_size(uint32_t base_w, uint32_t base_h, uint32_t components, int display, uint32_t *out_w,
			uint32_t *out_h)
{
	const uint32_t n = display_n_components(components);
	if (out_w)
		*out_w = display == 2 ? base_w * n : base_w;
	if (out_h)
		*out_h = display == 1 ? base_h * n : base_h;
}

int scope_display_vectorscope_zoom(scope_ctx *ctx, const uint8_t *table, int intensity, int color_type, int colorspace,
				   float zoom, uint32_t flags, uint8_t *rgba)
{
	if (!ctx)
		return SCOPE_EINVAL;
	if (intensity < 1)
		intensity = 1; /* src/vectorscope.c:129-131 */
	if (!(zoom >= 1.0f))
		zoom = 1.0f; /* :480-481 */
	return display_run(ctx, table, SCOPE_VSS_BINS, rgba, SCOPE_VSS_BINS * 4, flags, [&](const uint8_t *t, uint8_t *o) {
		return scope_launch_display_vss(t, o, intensity, color_type, colorspace, zoom, ctx->stream);
	});
}

int scope_display_vectorscope(scope_ctx *ctx, const uint8_t *table, int intensity, int color_type, int colorspace,
			      uint32_t flags, uint8_t *rgba)
{
	return scope_display_vectorscope_zoom(ctx, table, intensity, color_type, colorspace, 1.0f, flags, rgba);
}

int scope_display_waveform(scope_ctx *ctx, const uint8_t *table, uint32_t width, uint32_t components, int display,
			   int intensity, uint32_t flags, uint8_t *rgba)
{
	if (!ctx)
		return SCOPE_EINVAL;
	if (width == 0 || (uint64_t)width * 256 * 3 >= (1ull << 30))
		return fail(ctx, SCOPE_EINVAL, "bad waveform width", hipSuccess);
	uint32_t ow, oh;
	scope_display_size(width, SCOPE_WV_SIZE, components, display, &ow, &oh);
	const uint32_t n = display_n_components(components);
	return display_run(ctx, table, (size_t)SCOPE_WV_SIZE * width * 4, rgba, (size_t)ow * oh * 4, flags,
			   [&](const uint8_t *t, uint8_t *o) {
				   return scope_launch_display_wvs(t, o, width, n, display, intensity, ow, oh, ctx->stream);
			   });
}

int scope_display_histogram(scope_ctx *ctx, const float *tex, const uint32_t *hi_max, uint32_t components, int display,
			    uint32_t level_height, uint32_t flags, uint8_t *rgba)
{
	if (!ctx)
		return SCOPE_EINVAL;
	if (!hi_max || level_height == 0 || level_height > 65536)
		return fail(ctx, SCOPE_EINVAL, "bad histogram display arguments", hipSuccess);
	uint32_t ow, oh;
	scope_display_size(SCOPE_HI_SIZE, level_height, components, display, &ow, &oh);
	const uint32_t n = display_n_components(components);
	/* his_set_image: the maxima travel as floats (src/histogram.c:428-429) */
	const float hm0 = (float)hi_max[0], hm1 = (float)hi_max[1], hm2 = (float)hi_max[2];
	return display_run(ctx, tex, SCOPE_HIS_WORDS * sizeof(float), rgba, (size_t)ow * oh * 4, flags,
			   [&](const uint8_t *t, uint8_t *o) {
				   return scope_launch_display_his((const float *)t, o, hm0, hm1, hm2, n, display, ow, oh, ctx->stream);
			   });
}

/* ---- graticule over a finished picture ------------------------------------------------------------------------- */

void scope_graticule_label_strip(uint8_t *alpha)
{
	if (alpha)
		scope_graticule_strip(alpha);
}

int scope_display_graticule(scope_ctx *ctx, uint8_t *rgba, uint32_t width, uint32_t height,
			    const struct scope_graticule_label *labels, uint32_t n_labels,
			    const struct scope_graticule_vertex *lines, uint32_t n_line_vertices, uint32_t argb, uint32_t flags)
{
	SCOPE_ENTER(ctx);
	if (st)
		return st;
	if (!rgba || (n_labels && !labels) || (n_line_vertices && !lines) || (n_line_vertices & 1u))
		return fail(ctx, SCOPE_EINVAL, "scope_display_graticule: NULL list or an odd number of line vertices", hipSuccess);
	const uint32_t dev = flags & (SCOPE_IN_DEVICE | SCOPE_OUT_DEVICE);
	if (dev != 0 && dev != (SCOPE_IN_DEVICE | SCOPE_OUT_DEVICE))
		return fail(ctx, SCOPE_EINVAL, "the picture is a host or a device buffer, not a mix", hipSuccess);
	if ((uint64_t)width * height >= (1ull << 30) || n_labels > 4096 || n_line_vertices > 65536)
		return fail(ctx, SCOPE_EINVAL, "scope_display_graticule: picture or list too large", hipSuccess);
	const size_t pic_bytes = (size_t)width * height * 4;
	if (!pic_bytes || (!n_labels && !n_line_vertices))
		return SCOPE_OK;
	/* the lists and the glyph strip travel together: [quads | segments | strip] */
	const size_t quad_bytes = (size_t)n_labels * sizeof(struct scope_graticule_label);
	const size_t seg_bytes = (size_t)n_line_vertices * sizeof(struct scope_graticule_vertex);
	const size_t strip_at = round_up(quad_bytes + seg_bytes, 16), strip_bytes = 96 * 16;
	const size_t list_bytes = strip_at + strip_bytes;
	uint8_t *host = (uint8_t *)malloc(list_bytes);
	if (!host)
		return SCOPE_ENOMEM;
	if (quad_bytes)
		memcpy(host, labels, quad_bytes);
	if (seg_bytes)
		memcpy(host + quad_bytes, lines, seg_bytes);
	scope_graticule_strip(host + strip_at);
	st = grow(ctx, &ctx->stage_in, &ctx->stage_in_bytes, round_up(list_bytes, 256));
	if (!st && !dev)
		st = grow(ctx, &ctx->stage_out, &ctx->stage_out_bytes, round_up(pic_bytes, 256));
	if (st) {
		free(host);
		return st;
	}
	hipError_t e = hipMemcpyAsync(ctx->stage_in, host, list_bytes, hipMemcpyHostToDevice, ctx->stream);
	uint8_t *pic = dev ? rgba : ctx->stage_out;
	if (e == hipSuccess && !dev)
		e = hipMemcpyAsync(pic, rgba, pic_bytes, hipMemcpyHostToDevice, ctx->stream);
	if (e == hipSuccess)
		e = scope_launch_graticule(pic, width, height, (const float *)ctx->stage_in, n_labels,
					   (const float *)(ctx->stage_in + quad_bytes), n_line_vertices / 2, ctx->stage_in + strip_at, argb,
					   ctx->stream);
	if (e == hipSuccess && !dev)
		e = hipMemcpyAsync(rgba, pic, pic_bytes, hipMemcpyDeviceToHost, ctx->stream);
	/* `host` is pageable: the upload has consumed it only once the stream has drained */
	const hipError_t e2 = hipStreamSynchronize(ctx->stream);
	free(host);
	if (e != hipSuccess || e2 != hipSuccess)
		return fail(ctx, SCOPE_EIO, "scope_display_graticule", e != hipSuccess ? e : e2);
	return SCOPE_OK;
}

/* ---- frame pipeline ---------------------------------------------------------------------------- */

#define SCOPE_PIPE_MAX_DEPTH 16

struct scope_pipe_slot {
	uint8_t *dev_rgb;   /* what crossed PCIe: the frame, or just its sampled rows */
	uint8_t *dev_stage; /* the scaled / cropped plane (NULL when the region is the whole frame at scale 1) */
	uint8_t *dev_vss, *dev_wvs;
	uint32_t *dev_his;
	uint32_t host_index;  /* which of the pipe's pinned table buffers this frame's download goes to */
	hipEvent_t uploaded, computed, downloaded;
};

struct scope_pipe {
	scope_ctx *ctx;
	struct scope_pipe_config cfg;
	size_t pitch, vss_bytes, wvs_bytes, his_bytes;
	uint32_t scale, rx, ry, rw, rh; /* resolved region: the scopes see an rw x rh plane */
	size_t span, stage_pitch;       /* bytes of a source row that are uploaded; pitch of dev_stage */
	bool staged;
	hipStream_t up, down;
	scope_pipe_slot slot[SCOPE_PIPE_MAX_DEPTH];
	/* depth + 1 pinned buffers (vss | wvs | his back to back), used round robin: the tables handed out by
	 * scope_pipe_collect stay untouched until the next collect, also when the ring was full (at most `depth` frames can
	 * be submitted between two collects, and they go to the other `depth` buffers) */
	uint8_t *host_tables[SCOPE_PIPE_MAX_DEPTH + 1];
	uint64_t submitted, collected;
};

void scope_pipe_destroy(scope_pipe *p)
{
	if (!p)
		return;
	(void)hipSetDevice(p->ctx->device);
	(void)hipStreamSynchronize(p->ctx->stream);
	if (p->up)
		(void)hipStreamSynchronize(p->up);
	if (p->down)
		(void)hipStreamSynchronize(p->down);
	for (uint32_t i = 0; i < p->cfg.depth; i++) {
		scope_pipe_slot *s = &p->slot[i];
		if (s->dev_rgb)
			(void)hipFree(s->dev_rgb);
		if (s->dev_stage)
			(void)hipFree(s->dev_stage);
		if (s->dev_vss)
			(void)hipFree(s->dev_vss);
		if (s->dev_wvs)
			(void)hipFree(s->dev_wvs);
		if (s->dev_his)
			(void)hipFree(s->dev_his);
		if (s->uploaded)
			(void)hipEventDestroy(s->uploaded);
		if (s->computed)
			(void)hipEventDestroy(s->computed);
		if (s->downloaded)
			(void)hipEventDestroy(s->downloaded);
	}
	for (uint32_t i = 0; i <= p->cfg.depth && i <= SCOPE_PIPE_MAX_DEPTH; i++)
		if (p->host_tables[i])
			(void)hipHostFree(p->host_tables[i]);
	if (p->up)
		(void)hipStreamDestroy(p->up);
	if (p->down)
		(void)hipStreamDestroy(p->down);
	free(p);
}

int scope_pipe_create(scope_ctx *ctx, const struct scope_pipe_config *cfg, scope_pipe **out)
{
	SCOPE_ENTER(ctx);
	if (st)
		return st;
	if (!cfg || !out || cfg->width == 0 || cfg->height == 0 || cfg->depth == 0 || cfg->depth > SCOPE_PIPE_MAX_DEPTH)
		return fail(ctx, SCOPE_EINVAL, "bad pipeline configuration", hipSuccess);
	st = check_geometry(ctx, cfg->width * 4, cfg->width, cfg->height);
	if (st)
		return st;
	uint32_t rs, rx, ry, rw, rh;
	if (resolve_region(cfg->width, cfg->height, &cfg->region, &rs, &rx, &ry, &rw, &rh))
		return fail(ctx, SCOPE_EINVAL, "pipeline region outside the scaled frame", hipSuccess);
	scope_pipe *p = (scope_pipe *)calloc(1, sizeof *p);
	if (!p)
		return SCOPE_ENOMEM;
	p->ctx = ctx;
	p->cfg = *cfg;
	p->scale = rs;
	p->rx = rx;
	p->ry = ry;
	p->rw = rw;
	p->rh = rh;
	p->staged = rs != 1; /* a pure crop is done by the 2-D upload itself */
	p->span = ((size_t)(rw - 1) * rs + 1) * 4; /* first to last sampled texel of a row */
	p->pitch = round_up(p->span, 16);
	p->stage_pitch = round_up((size_t)rw * 4, 16);
	p->vss_bytes = cfg->want_vectorscope ? SCOPE_VSS_BINS : 0;
	p->wvs_bytes = (cfg->wvs_components & 0x77u) ? (size_t)SCOPE_WV_SIZE * rw * 4 : 0;
	p->his_bytes = (cfg->his_components & 0x77u) ? SCOPE_HIS_WORDS * sizeof(uint32_t) : 0;
	bool ok = hipStreamCreateWithFlags(&p->up, hipStreamNonBlocking) == hipSuccess &&
		  hipStreamCreateWithFlags(&p->down, hipStreamNonBlocking) == hipSuccess;
	for (uint32_t i = 0; ok && i < cfg->depth; i++) {
		scope_pipe_slot *s = &p->slot[i];
		ok = hipMalloc((void **)&s->dev_rgb, round_up(p->pitch * rh, 256)) == hipSuccess;
		if (ok && p->staged)
			ok = hipMalloc((void **)&s->dev_stage, round_up(p->stage_pitch * rh, 256)) == hipSuccess;
		if (ok && p->vss_bytes)
			ok = hipMalloc((void **)&s->dev_vss, p->vss_bytes) == hipSuccess;
		if (ok && p->wvs_bytes)
			ok = hipMalloc((void **)&s->dev_wvs, p->wvs_bytes) == hipSuccess;
		if (ok && p->his_bytes)
			ok = hipMalloc((void **)&s->dev_his, p->his_bytes) == hipSuccess;
		ok = ok && hipEventCreate(&s->uploaded) == hipSuccess && hipEventCreate(&s->computed) == hipSuccess &&
		     hipEventCreate(&s->downloaded) == hipSuccess;
	}
	for (uint32_t i = 0; ok && i <= cfg->depth; i++)
		ok = hipHostMalloc((void **)&p->host_tables[i], p->vss_bytes + p->wvs_bytes + p->his_bytes + 64, 0) == hipSuccess;
	if (!ok) {
		scope_pipe_destroy(p);
		return fail(ctx, SCOPE_ENOMEM, "pipeline allocation", hipSuccess);
	}
	*out = p;
	return SCOPE_OK;
}

int scope_pipe_size(const scope_pipe *p, uint32_t *width, uint32_t *height)
{
	if (!p || !width || !height)
		return SCOPE_EINVAL;
	*width = p->rw;
	*height = p->rh;
	return SCOPE_OK;
}

namespace {

/* Everything scope_pipe_submit enqueues for one frame.  A failure half-way leaves work queued on up to three streams. */
int pipe_enqueue(scope_pipe *p, scope_pipe_slot *s, const uint8_t *rgb_host, uint32_t linesize)
{
	scope_ctx *ctx = p->ctx;
	int st = SCOPE_OK;
	const uint32_t w = p->rw, h = p->rh;
	const uint32_t dev_flags = SCOPE_IN_DEVICE | SCOPE_OUT_DEVICE | SCOPE_ASYNC;

	/* upload stream: the RGB plane only -- half the bytes of the reference's RGB+YUV staging surface -- and of
	 * it only the rows (and the span of columns) that target_scale / the ROI sample */
	const uint8_t *first = rgb_host + ((size_t)p->ry * p->scale + p->scale / 2) * linesize +
			       ((size_t)p->rx * p->scale + p->scale / 2) * 4;
	HIP_TRY(ctx, hipMemcpy2DAsync(s->dev_rgb, p->pitch, first, (size_t)p->scale * linesize, p->span, h,
				      hipMemcpyHostToDevice, p->up));
	HIP_TRY(ctx, hipEventRecord(s->uploaded, p->up));

	/* compute stream: the scopes of the same HBM-resident plane, K0 fused where Y'CbCr is needed: the vectorscope
	 * reads it once, waveform and histogram share a second read (the histogram rides on the waveform's pass) */
	HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, s->uploaded, 0));
	const uint8_t *plane[1] = {s->dev_rgb};
	uint32_t pitch = (uint32_t)p->pitch;
	if (p->staged) {
		HIP_TRY(ctx, scope_launch_stage(s->dev_rgb, p->pitch, p->scale, w, h, s->dev_stage, (uint32_t)p->stage_pitch,
						ctx->stream));
		plane[0] = s->dev_stage;
		pitch = (uint32_t)p->stage_pitch;
	}
	bool his_done = false, all_done = false;
	if (p->vss_bytes && p->wvs_bytes && (p->cfg.wvs_components & 0x07u) && (!p->his_bytes || (p->cfg.his_components & 0x07u))) {
		/* all three from ONE read where the frame qualifies (scope_dock_kernel: one round of 16-column strips) */
		const uint32_t cw = channel_mask(p->cfg.wvs_components), ch = p->his_bytes ? channel_mask(p->cfg.his_components) : 0u;
		st = run_dock(ctx, plane, 1, pitch, w, h, cw, ch, p->cfg.colorspace, dev_flags, s->dev_vss, s->dev_wvs,
			      p->his_bytes ? (uint32_t *)s->dev_his : NULL);
		if (st < 0)
			return st;
		all_done = his_done = st == 1;
		st = SCOPE_OK;
	}
	if (p->vss_bytes && !all_done) {
		st = run_vectorscope(ctx, plane, 1, pitch, w, h, 1, p->cfg.colorspace, dev_flags, s->dev_vss);
		if (st)
			return st;
	}
	if (p->wvs_bytes && !all_done) {
		const uint32_t c = p->cfg.wvs_components, hc = p->cfg.his_components;
		/* the histogram rides on the waveform's pass when it counts the same kind of plane and no other channels */
		his_done = p->his_bytes && ((c & 0x07u) != 0) == ((hc & 0x07u) != 0) && (channel_mask(hc) & ~channel_mask(c)) == 0;
		st = run_waveform(ctx, plane, 1, pitch, w, h, channel_mask(c), !(c & 0x07u), p->cfg.colorspace, dev_flags, s->dev_wvs,
				  his_done ? s->dev_his : NULL, channel_mask(hc));
		if (st)
			return st;
	}
	if (p->his_bytes && !his_done) {
		const uint32_t c = p->cfg.his_components;
		st = run_histogram(ctx, plane, 1, pitch, w, h, channel_mask(c), !(c & 0x07u), p->cfg.colorspace,
				   dev_flags, s->dev_his);
		if (st)
			return st;
	}
	HIP_TRY(ctx, hipEventRecord(s->computed, ctx->stream));

	/* download stream: the small tables into pinned memory */
	HIP_TRY(ctx, hipStreamWaitEvent(p->down, s->computed, 0));
	uint8_t *dst = p->host_tables[s->host_index];
	if (p->vss_bytes)
		HIP_TRY(ctx, hipMemcpyAsync(dst, s->dev_vss, p->vss_bytes, hipMemcpyDeviceToHost, p->down));
	if (p->wvs_bytes)
		HIP_TRY(ctx, hipMemcpyAsync(dst + p->vss_bytes, s->dev_wvs, p->wvs_bytes, hipMemcpyDeviceToHost, p->down));
	if (p->his_bytes)
		HIP_TRY(ctx, hipMemcpyAsync(dst + p->vss_bytes + p->wvs_bytes, s->dev_his, p->his_bytes, hipMemcpyDeviceToHost,
					    p->down));
	HIP_TRY(ctx, hipEventRecord(s->downloaded, p->down));
	return SCOPE_OK;
}

} // namespace

extern "C" int scope_pipe_submit(scope_pipe *p, const uint8_t *rgb_host, uint32_t linesize)
{
	if (!p || !rgb_host)
		return SCOPE_EINVAL;
	scope_ctx *ctx = p->ctx;
	SCOPE_ENTER(ctx);
	if (st)
		return st;
	if ((uint64_t)p->cfg.width * 4 > linesize)
		return fail(ctx, SCOPE_EINVAL, "linesize < 4*width", hipSuccess);
	if (p->submitted - p->collected >= p->cfg.depth)
		return SCOPE_EAGAIN;
	scope_pipe_slot *s = &p->slot[p->submitted % p->cfg.depth];
	s->host_index = (uint32_t)(p->submitted % (p->cfg.depth + 1));
	st = pipe_enqueue(p, s, rgb_host, linesize);
	if (st) {
		/* roll back: whatever was queued for this frame drains (its events may be half recorded, its buffers half
		 * written), and the slot is NOT taken -- the next submit starts it from scratch and collect never sees it */
		char keep[sizeof ctx->err];
		memcpy(keep, ctx->err, sizeof keep);
		(void)hipStreamSynchronize(p->up);
		(void)hipStreamSynchronize(ctx->stream);
		(void)hipStreamSynchronize(p->down);
		memcpy(ctx->err, keep, sizeof keep);
		return st;
	}
	p->submitted++;
	return SCOPE_OK;
}

int scope_pipe_collect(scope_pipe *p, const uint8_t **vectorscope, const uint8_t **waveform, const uint32_t **his_counts)
{
	if (!p)
		return SCOPE_EINVAL;
	scope_ctx *ctx = p->ctx;
	SCOPE_ENTER(ctx);
	if (st)
		return st;
	if (p->submitted == p->collected)
		return SCOPE_EAGAIN;
	scope_pipe_slot *s = &p->slot[p->collected % p->cfg.depth];
	HIP_TRY(ctx, hipEventSynchronize(s->downloaded));
	const uint8_t *tables = p->host_tables[s->host_index];
	if (vectorscope)
		*vectorscope = p->vss_bytes ? tables : NULL;
	if (waveform)
		*waveform = p->wvs_bytes ? tables + p->vss_bytes : NULL;
	if (his_counts)
		*his_counts = p->his_bytes ? (const uint32_t *)(tables + p->vss_bytes + p->wvs_bytes) : NULL;
	p->collected++;
	return SCOPE_OK;
}

} /* extern "C" */

/* ---- multi-GPU: the table sum as a C call -------------------------------------------------------------------------
 * RCCL is bound at run time (dlopen of the librccl already in the process, else the ROCm one), so a host that never
 * sums tables across GPUs does not load it and the library has no link-time dependency on it.  The five entry points
 * are declared here exactly as <rccl/rccl.h> does (ncclUniqueId is a 128-byte struct passed by value). */
namespace {

struct rccl_id {
	char internal[128];
};
struct rccl_api {
	void *handle;
	int (*get_unique_id)(rccl_id *);
	int (*comm_init_rank)(void **comm, int nranks, rccl_id id, int rank);
	int (*comm_destroy)(void *comm);
	int (*all_reduce)(const void *send, void *recv, size_t count, int datatype, int op, void *comm, hipStream_t stream);
	const char *(*error_string)(int);
	int (*comm_count)(const void *comm, int *count);    /* optional: scope_rccl_comm_info */
	int (*comm_user_rank)(const void *comm, int *rank); /* optional */
	int (*get_version)(int *version);                   /* optional */
};

const rccl_api *rccl()
{
	static std::mutex mu;
	static rccl_api api;
	static bool tried = false;
	std::lock_guard<std::mutex> lock(mu);
	if (!tried) {
		tried = true;
		/* SCOPE_RCCL_LIBRARY names the collective library of the host (a path or a soname exporting the nccl* entry
		 * points below); when it is set nothing else is tried, so a host never gets another RCCL than the one it asked for */
		const char *forced = getenv("SCOPE_RCCL_LIBRARY");
		const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
		if (forced && *forced)
			api.handle = dlopen(forced, RTLD_NOW | RTLD_GLOBAL);
		else
			for (const char *n : names) {
				api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
				if (api.handle)
					break;
			}
		if (api.handle) {
			api.get_unique_id = (int (*)(rccl_id *))dlsym(api.handle, "ncclGetUniqueId");
			api.comm_init_rank = (int (*)(void **, int, rccl_id, int))dlsym(api.handle, "ncclCommInitRank");
			api.comm_destroy = (int (*)(void *))dlsym(api.handle, "ncclCommDestroy");
			api.all_reduce = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))dlsym(api.handle, "ncclAllReduce");
			api.error_string = (const char *(*)(int))dlsym(api.handle, "ncclGetErrorString");
			api.comm_count = (int (*)(const void *, int *))dlsym(api.handle, "ncclCommCount");
			api.comm_user_rank = (int (*)(const void *, int *))dlsym(api.handle, "ncclCommUserRank");
			api.get_version = (int (*)(int *))dlsym(api.handle, "ncclGetVersion");
			if (!api.get_unique_id || !api.comm_init_rank || !api.comm_destroy || !api.all_reduce)
				api.handle = NULL;
		}
	}
	return api.handle ? &api : NULL;
}

int rccl_fail(scope_ctx *ctx, const char *what, int rc)
{
	const rccl_api *r = rccl();
	if (ctx)
		snprintf(ctx->err, sizeof ctx->err, "%s: %s", what, (r && r->error_string) ? r->error_string(rc) : "RCCL not loadable");
	return SCOPE_EIO;
}

} // namespace

extern "C" {

int scope_rccl_unique_id(uint8_t id[SCOPE_RCCL_ID_BYTES])
{
	const rccl_api *r = rccl();
	if (!r || !id)
		return r ? SCOPE_EINVAL : SCOPE_ENODEV;
	rccl_id u;
	if (r->get_unique_id(&u) != 0)
		return SCOPE_EIO;
	memcpy(id, u.internal, sizeof u.internal);
	return SCOPE_OK;
}

int scope_rccl_comm_create(scope_ctx *ctx, const uint8_t id[SCOPE_RCCL_ID_BYTES], int n_ranks, int rank, void **comm)
{
	SCOPE_ENTER(ctx);
	if (st)
		return st;
	if (!id || !comm || n_ranks < 1 || rank < 0 || rank >= n_ranks)
		return fail(ctx, SCOPE_EINVAL, "scope_rccl_comm_create", hipSuccess);
	if (n_ranks > 257)
		return fail(ctx, SCOPE_EINVAL, "more than 257 ranks: the packed 16-bit halves of the table sum could carry", hipSuccess);
	const rccl_api *r = rccl();
	if (!r)
		return rccl_fail(ctx, "dlopen(librccl)", 0);
	rccl_id u;
	memcpy(u.internal, id, sizeof u.internal);
	*comm = NULL;
	const int rc = r->comm_init_rank(comm, n_ranks, u, rank); /* binds to the current device = the context's */
	return rc ? rccl_fail(ctx, "ncclCommInitRank", rc) : SCOPE_OK;
}

int scope_rccl_comm_destroy(void *comm)
{
	const rccl_api *r = rccl();
	if (!r || !comm)
		return SCOPE_EINVAL;
	return r->comm_destroy(comm) ? SCOPE_EIO : SCOPE_OK;
}

int scope_rccl_comm_info(void *rccl_comm, int *n_ranks, int *rank, int *version)
{
	if (n_ranks)
		*n_ranks = -1;
	if (rank)
		*rank = -1;
	if (version)
		*version = -1;
	if (!rccl_comm && (n_ranks || rank))
		return SCOPE_EINVAL; /* before anything is filled in */
	const rccl_api *r = rccl();
	if (!r)
		return SCOPE_ENODEV;
	if (version && r->get_version && r->get_version(version) != 0)
		return SCOPE_EIO;
	if (!rccl_comm)
		return SCOPE_OK;
	if (n_ranks && r->comm_count && r->comm_count(rccl_comm, n_ranks) != 0)
		return SCOPE_EIO;
	if (rank && r->comm_user_rank && r->comm_user_rank(rccl_comm, rank) != 0)
		return SCOPE_EIO;
	return SCOPE_OK;
}

int scope_table_allreduce(scope_ctx *ctx, void *rccl_comm, uint32_t *words, uint64_t n_words)
{
	SCOPE_ENTER(ctx);
	if (st)
		return st;
	if (!rccl_comm || !words)
		return fail(ctx, SCOPE_EINVAL, "scope_table_allreduce: NULL argument", hipSuccess);
	if (n_words == 0)
		return SCOPE_OK;
	const rccl_api *r = rccl();
	if (!r)
		return rccl_fail(ctx, "dlopen(librccl)", 0);
	/* ncclUint32 = 3, ncclSum = 0; in place, on the context's stream (ordered behind the kernels that filled the table) */
	const int rc = r->all_reduce(words, words, (size_t)n_words, 3, 0, rccl_comm, ctx->stream);
	return rc ? rccl_fail(ctx, "ncclAllReduce", rc) : SCOPE_OK;
}

} /* extern "C" */
