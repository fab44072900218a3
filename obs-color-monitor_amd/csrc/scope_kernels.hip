/*
 * scope_kernels.hip -- hand-written gfx950 (CDNA4 / MI355X) kernels for obs-color-monitor's
 * scope accumulation: K0 RGB->Y'CbCr, K1 vectorscope, K2 waveform, K3 histogram.
 *
 * Design (DESIGN.md has the full story):
 *   * HBM-bound byte/integer work, no MFMA.  Every pixel is read from HBM exactly once with
 *     coalesced streaming loads (16 B/lane for K0/K1/K3, 8 B/lane = two 256-byte row segments per wave for K2),
 *     about 32-128 KiB of them in flight per CU (fewer is faster: DESIGN.md section 5).
 *   * One 1024-thread workgroup per CU owns a private table in LDS (128 KiB u16-packed
 *     256x256 for K1; 96 KiB lane-replicated 3x256 for K3; 32 KiB per channel column-strip
 *     table for K2) and accumulates with LDS atomics; bank placement is chosen so that the
 *     K2/K3 atomics are bank-conflict free by construction and K1's spread over all 32 banks
 *     for clustered chroma.
 *   * A table leaves the CU once, as a coalesced 16 B/lane store of clamped u8 (K1/K2) or
 *     u32 (K3) partials; partial tables of one frame are combined by a tiny second kernel
 *     (saturating add is exact: min(min(a,255)+min(b,255),255) == min(a+b,255)) or, for the histogram's small
 *     tables, by the last workgroup of the frame.
 *   * K0 inside the scopes runs in 24-bit fixed point with an exhaustively derived guard band and an fp32 redo
 *     near a step: bit-exact with the fp32 definition, a third of its instructions.
 *   * One-frame launches have shapes of their own: the waveform in 16-column strips (240 units per 4K frame, no row
 *     bands, no partial tables), and scope_dock_kernel, which keeps a strip's texels in REGISTERS and walks them for
 *     waveform + histogram and again for the vectorscope: the dock's three tables from one read of the frame.
 *
 * Reference semantics (relative to /root/reference): data/common.effect:23-43,
 * src/vectorscope.c:217-238, src/waveform.c:220-257, src/histogram.c:357-395.
 *
 * Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (K0 must not be FMA-contracted).
 */
#include "scope_device.h"


#ifdef SCOPE_TRACE_WG
__device__ unsigned long long scope_dbg_times[1024 * 4];
extern "C" int scope_debug_read(unsigned long long *out)
{
	return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(scope_dbg_times), sizeof(unsigned long long) * 1024 * 4);
}
#define SCOPE_STAMP(k) do { if (threadIdx.x == 0) scope_dbg_times[blockIdx.x * 4 + (k)] = wall_clock64(); } while (0)
#else
#define SCOPE_STAMP(k) do { } while (0)
#endif

static void k0_fixed_row(const float c[3], double offset, int out[4])
{
	for (int i = 0; i < 3; i++)
		out[i] = (int)lrint((double)c[i] * (double)(1 << SCOPE_K0_FX_SHIFT)); /* 24 bits: signed (U, V) or unsigned (Y) */
	out[3] = (int)(uint32_t)(llrint((255.0 * offset + 0.5) * (double)(1 << SCOPE_K0_FX_SHIFT)) + SCOPE_K0_FX_GUARD);
}

ScopeK0 scope_k0_matrix(int colorspace)
{
	/* src/common.c:207-208: colorspace == 1 -> 601, anything else -> 709 */
	ScopeK0 k;
	if (colorspace == 1) {
		k.ur = -0.147643f, k.ug = -0.289855f, k.ub = +0.437500f;
		k.yr = +0.299000f, k.yg = +0.587000f, k.yb = +0.114000f;
		k.vr = +0.437500f, k.vg = -0.366351f, k.vb = -0.071147f;
	} else {
		k.ur = -0.100643f, k.ug = -0.338571f, k.ub = +0.439216f;
		k.yr = +0.212600f, k.yg = +0.715200f, k.yb = +0.072200f;
		k.vr = +0.439216f, k.vg = -0.398941f, k.vb = -0.040273f;
	}
	const float cu[3] = {k.ur, k.ug, k.ub}, cy[3] = {k.yr, k.yg, k.yb}, cv[3] = {k.vr, k.vg, k.vb};
	k0_fixed_row(cu, (double)0.5f - (double)0.00390625f, k.fu);
	k0_fixed_row(cy, 0.0, k.fy);
	k0_fixed_row(cv, (double)0.5f, k.fv);
	return k;
}

/* The same rows with FULL-RANGE chroma (x 255 / 224): a matrix the library does not offer.  It exists for one caller, the self-test of
 * scope_trio_kernel's launch-time chroma guard, which must refuse it (its U and V reach 0 and 255, outside the compact table). */
ScopeK0 scope_k0_matrix_full_range(int colorspace)
{
	ScopeK0 k = scope_k0_matrix(colorspace);
	const float sc = 255.0f / 224.0f;
	k.ur *= sc, k.ug *= sc, k.ub *= sc, k.vr *= sc, k.vg *= sc, k.vb *= sc;
	const float cu[3] = {k.ur, k.ug, k.ub}, cv[3] = {k.vr, k.vg, k.vb};
	k0_fixed_row(cu, (double)0.5f - (double)0.00390625f, k.fu);
	k0_fixed_row(cv, (double)0.5f, k.fv);
	return k;
}

/* ------------------------------------------------------------------------------------------
 * Pixel streaming
 *
 * A workgroup walks cells [c0, c1) of a (rows x ncols) grid of VECW-pixel cells.  The 64 lanes of a
 * wave are split into G = 2^glog2 lane groups and the cell range into G contiguous streams; lane
 * group g of every wave walks stream g, so one wave instruction touches G image regions that lie
 * far apart (each still a contiguous 1024/G-byte run, i.e. whole 128-byte lines for G <= 8).
 * Neighbouring pixels of natural video share their chroma, and LDS atomics on one address
 * serialise at ~4 clk per lane (profiles/round1/lds_atomic_microbench.txt); spreading a wave over
 * G regions divides that multiplicity without giving up coalescing.
 *
 * Loads are double-buffered: the U loads of batch b+1 are in flight while batch b is binned, and
 * `before_batch` (uniform; may contain barriers) runs between issue and use, so the vectorscope's
 * clamp passes do not drain the memory pipeline.  `after_first_issue` (uniform; may contain barriers) runs once, with
 * the first batch in flight: the kernels zero their LDS table there, under the HBM latency of the first loads instead
 * of in front of it (0.7 us of a one-frame launch).  `fn(v, ok)` gets a whole batch: U cells per lane
 * and which of them are real.  Row pitch is honoured.
 * ---------------------------------------------------------------------------------------- */


template <int VECW> struct ScopeVec;
template <> struct ScopeVec<4> {
	typedef scope_v4 type;
};
template <> struct ScopeVec<1> {
	typedef u32 type;
};

template <bool NT, typename V> __device__ __forceinline__ V scope_load(const V *p)
{
	if (NT)
		return __builtin_nontemporal_load(p);
	return *p;
}

/* min(sum over `slices` partial tables, 255) for 16 byte-wide bins; slices <= 256, so the packed 16-bit lanes
 * (max 256*255 = 65280) cannot overflow.  p points at the bins in the first partial, `stride` uint4s apart. */
__device__ __forceinline__ uint4 scope_sat_sum16(const uint4 *p, u64 stride, u32 slices)
{
	u32 ev[4] = {0, 0, 0, 0}, od[4] = {0, 0, 0, 0};
	for (u32 s = 0; s < slices; s++) {
		const uint4 q = p[(u64)s * stride];
		ev[0] += q.x & 0x00ff00ffu;
		od[0] += (q.x >> 8) & 0x00ff00ffu;
		ev[1] += q.y & 0x00ff00ffu;
		od[1] += (q.y >> 8) & 0x00ff00ffu;
		ev[2] += q.z & 0x00ff00ffu;
		od[2] += (q.z >> 8) & 0x00ff00ffu;
		ev[3] += q.w & 0x00ff00ffu;
		od[3] += (q.w >> 8) & 0x00ff00ffu;
	}
	u32 r[4];
	for (int j = 0; j < 4; j++)
		r[j] = min(ev[j] & 0xffffu, 255u) | (min(od[j] & 0xffffu, 255u) << 8) | (min(ev[j] >> 16, 255u) << 16) |
		       (min(od[j] >> 16, 255u) << 24);
	return make_uint4(r[0], r[1], r[2], r[3]);
}

#ifndef SCOPE_SLICE_ALIGN
#define SCOPE_SLICE_ALIGN 32 /* 16-byte cells: 512 bytes */
#endif
/* The same with the inner boundaries on multiples of SCOPE_SLICE_ALIGN cells (a power of two; 0 = as they fall): slices
 * that start on 512-byte boundaries read a little faster than slices that start anywhere -- a three-frame 4K launch
 * (85 slices per frame, 24 395 cells each) 29.9 -> 29.0 us per call, a one-frame launch (8100 cells each) 17.1 -> 17.0;
 * 4 KiB boundaries do the same. */
__device__ __forceinline__ u32 scope_slice_begin_cells(u32 total, u32 s, u32 S)
{
	const u32 c = scope_slice_begin(total, s, S);
	if (SCOPE_SLICE_ALIGN == 0 || s == 0 || s >= S)
		return c;
	return c & ~((u32)SCOPE_SLICE_ALIGN - 1u);
}

#ifndef WVS_PRIO
#ifndef WVS_STORE_NT
#define WVS_STORE_NT 4 /* the batch form's table stores.  4: by channel count -- write-through (sc1) for the one-channel masks, streaming (nt)
			* for the others.  Round 6's write-side A/B on three boxes (profiles/round6/ab_waveform_stores_*.txt):
			* against nt (1, rounds 1-5), sc1 (2) gives the one-channel modes +3.0 ... +4.1 % and RGB / luma -1.6 ... +0.2 %;
			* plain stores (0) lose 0 ... 10 %, sc1 + nt (3) 1 ... 2 % */
#endif
#ifndef WVS_NARROW_STORE
#define WVS_NARROW_STORE 2 /* the narrow-strip (one-frame) form's table stores: 2 write-through (+1 ... +3.6 % at one and three frames per call),
			    * 1 streaming (rounds 2-5; A/B) */
#endif
#define WVS_PRIO 1 /* the same inside scope_wvs_kernel's units, narrow-strip form only (see there); 0: A/B */
#endif
#ifndef SCOPE_STREAM_PRIO
#define SCOPE_STREAM_PRIO 1 /* the waves' issue priority goes round, a step per pair of batches (scope_prio_round); 0: A/B */
#endif
#ifndef SCOPE_STREAM_MAIN
#define SCOPE_STREAM_MAIN 1 /* 0: every batch through the checked form (A/B) */
#endif
/* ROWS (padded planes, linesize > ncols * cell size: what a D3D11 staging surface can look like, src/vectorscope.c:225,236): the
 * streams are whole ROWS -- [c0, c1) is a range of rows, lane group g walks its share of them -- and a row is a multiple of
 * a lane group's cells (scope_rows_mode), so every lane of a wave is in the same column phase: whether a load's successor lies
 * in the next row is a WAVE-UNIFORM question, the column cursor lives in a scalar register, and the per-lane address costs
 * one add of a scalar increment (the pitch padding folded in when the cursor wraps), as in the CONTIG form.  The general
 * form below keeps the column per lane: eleven operations per load, 9 % of the fused vectorscope kernel on a padded 4K plane. */
template <int VECW, int U, bool NT, bool CONTIG, bool MAIN = CONTIG, bool ROUNDS = false, bool ROWS = false, typename Pro, typename Hook,
	  typename F>
__device__ __forceinline__ void scope_stream(const uint8_t *plane, u32 linesize, u32 xbyte0, u32 ncols, u32 c0, u32 c1,
					     u32 glog2, Pro &&after_first_issue, Hook &&before_batch, F &&fn)
{
	typedef typename ScopeVec<VECW>::type vec_t;
	if (c0 >= c1)
		return;
	const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	const u32 wave_u = (u32)__builtin_amdgcn_readfirstlane((int)wave);
	(void)wave_u;
	const u32 lg = 64u >> glog2;            /* lanes per group */
	const u32 g = lane >> (6u - glog2);     /* my group = my stream */
	const u32 tg = blockDim.x >> glog2;     /* lanes of the workgroup on one stream */
	const u32 len = c1 - c0;
	u32 s0, s1, longest, shortest;
	if (ROWS && !CONTIG) {
		const u32 r0 = c0 / ncols, rl = len / ncols; /* c0 and c1 are row boundaries */
		s0 = (r0 + (u32)(((u64)rl * g) >> glog2)) * ncols;
		s1 = (r0 + (u32)(((u64)rl * (g + 1)) >> glog2)) * ncols;
		longest = ((rl + (1u << glog2) - 1u) >> glog2) * ncols;
		shortest = (rl >> glog2) * ncols;
	} else {
		s0 = c0 + (u32)(((u64)len * g) >> glog2);
		s1 = c0 + (u32)(((u64)len * (g + 1)) >> glog2);
		longest = (len + (1u << glog2) - 1u) >> glog2;
		shortest = len >> glog2;
	}
	const u32 step = (u32)U * tg;
	const u32 batches = (longest + step - 1u) / step;
	/* priority rounds (scope_prio_round): the histogram's kernels only -- RGB +0.8 ... 1.6 %, the fused Y'CbCr form +5 %; the
	 * vectorscope's forms are indifferent to them or lose (plain +0.2 % on noise, -0.7 % on ramps; fused -0.8 % on noise) -- and
	 * only where a slice is many batches long: launches of a few frames run two to six batches per slice and lost 1 % to them
	 * (visits p1 - p3, profiles/round4/ab_prio_rounds.txt) */
	const bool PRIO = ROUNDS && SCOPE_STREAM_PRIO != 0 && batches >= 16u;

	u32 c = s0 + wave * lg + (lane & (lg - 1u));
	const u32 dq = tg / ncols;
	const u32 dr = tg - dq * ncols;

	/* Addresses: a wave-uniform base (the slice's first row) plus a 32-bit byte offset per lane that is
	 * advanced incrementally -- d_off per load, plus the pitch padding whenever the column wraps -- so a load
	 * costs a handful of 32-bit operations instead of 64-bit multiply-adds.  (The launchers keep a slice's
	 * span below 4 GiB.)
	 * Loads are unconditional (lanes past their stream re-read the slice's first cell) so that the
	 * compiler can count them, and the two register sets A/B are used ping-pong without copies:
	 * the wait before binning a batch is vmcnt(U) and the next batch stays in flight. */
	const u32 cell_bytes = 4u * VECW;
	const u32 row0 = c0 / ncols;
	const uint8_t *base = plane + (size_t)row0 * linesize + xbyte0;
	const u32 safe_off = (c0 - row0 * ncols) * cell_bytes;
	const u32 row = c / ncols;
	u32 col = c - row * ncols;
	u32 off = (row - row0) * linesize + col * cell_bytes;
	const u32 d_off = dq * linesize + dr * cell_bytes;
	const u32 wrap_off = linesize - ncols * cell_bytes; /* what a column wrap adds on top: the pitch padding */
	/* CONTIG: the rows follow each other without padding (linesize == ncols * cell_bytes, the usual case), so a
	 * cell's offset is its index times the cell size: no column bookkeeping, three operations per load instead of
	 * eleven.  Same-process A/B: histogram +1 %, fused-K0 vectorscope +4 %, plain vectorscope +0.6 % on noise and
	 * +3.7 % on screen content (once its per-pixel arithmetic had been trimmed; before that it lost 3 % there). */
	const u32 end_off = ROWS && !CONTIG ? (s1 / ncols - row0) * linesize : (s1 - row0 * ncols) * cell_bytes;
	/* ROWS: the wave's column cursor (the lane's column is col_s + its place in the lane group) and what a wrap adds */
	u32 col_s = ROWS ? (wave_u * lg) % ncols : 0u;
	const u32 d_off_wrap = d_off + wrap_off;
	(void)col_s;
	(void)d_off_wrap;
	vec_t va[U], vb[U];
	bool oka[U], okb[U];

#define SCOPE_ISSUE(v, ok)                                                                             \
	do {                                                                                           \
		_Pragma("unroll") for (int k = 0; k < U; k++)                                          \
		{                                                                                      \
			if (CONTIG) {                                                                  \
				ok[k] = off < end_off;                                                 \
				v[k] = scope_load_base<NT, vec_t>(base, ok[k] ? off : safe_off);       \
				off += tg * cell_bytes;                                                \
			} else if (ROWS) {                                                             \
				ok[k] = off < end_off;                                                 \
				v[k] = scope_load_base<NT, vec_t>(base, ok[k] ? off : safe_off);       \
				col_s += dr;                                                           \
				const bool wrap_ = col_s >= ncols;                                     \
				col_s = wrap_ ? col_s - ncols : col_s;                                 \
				off += wrap_ ? d_off_wrap : d_off;                                     \
			} else {                                                                       \
				ok[k] = (c + (u32)k * tg) < s1;                                        \
				v[k] = scope_load_base<NT, vec_t>(base, ok[k] ? off : safe_off);       \
				off += d_off;                                                          \
				col += dr;                                                             \
				if (col >= ncols) {                                                    \
					col -= ncols;                                                  \
					off += wrap_off;                                               \
				}                                                                      \
			}                                                                              \
		}                                                                                      \
		c += step;                                                                             \
	} while (0)
#define SCOPE_BIN(v, ok)                                                  \
	do {                                                              \
		before_batch((u32)U * blockDim.x * VECW);                 \
		fn(v, ok);                                                \
	} while (0)

	/* Every ISSUE is unconditional -- past the end it only re-reads the safe cell with ok = false --
	 * so on every path exactly U younger loads are outstanding when a batch is binned and the
	 * compiler can wait with vmcnt(U) instead of draining the queue. */
	SCOPE_ISSUE(va, oka);
	after_first_issue();
	u32 b = 0;
	if ((CONTIG || ROWS) && MAIN && SCOPE_STREAM_MAIN) {
		/* The first `full` batches are inside the stream for every lane of every group (the shortest group has
		 * len >> glog2 cells): no range test, no address select and no predication there -- two VALU instructions per
		 * load and an exec-mask branch per load less (what the fused-K0 kernels are short of); the rest of the stream,
		 * one or two iterations, runs the checked form below.  oka stays all true across this loop: it was set by the
		 * checked ISSUE of batch 0, and the loop only runs when that batch is full.
		 * Same-process A/B on 64 x 4K: fused-K0 vectorscope +1.9 %, fused Y'CbCr histogram +4 %, plain vectorscope
		 * +0.3 %; the plain RGB histogram -2 .. -5 % (its callers pass MAIN = false).  (A form in which the vectorscope's
		 * main loop knew only the plain adds and left for the general loop when the vote said absorb -- to get the vote's
		 * 1.4 % back -- gained nothing on noise and lost 6 % on flat and 13 % on screen content: both bodies in one loop.) */
		const u32 full = shortest / step;
		bool yes[U];
#pragma unroll
		for (int k = 0; k < U; k++)
			yes[k] = true;
#define SCOPE_ISSUE_FULL(v)                                                                        \
	do {                                                                                       \
		_Pragma("unroll") for (int k = 0; k < U; k++)                                      \
		{                                                                                  \
			v[k] = scope_load_base<NT, vec_t>(base, off);                              \
			if (CONTIG) {                                                              \
				off += tg * cell_bytes;                                            \
			} else {                                                                   \
				col_s += dr;                                                       \
				const bool wrap_ = col_s >= ncols;                                 \
				col_s = wrap_ ? col_s - ncols : col_s;                             \
				off += wrap_ ? d_off_wrap : d_off;                                 \
			}                                                                          \
		}                                                                                  \
		c += step;                                                                         \
	} while (0)
		for (; b + 2 < full; b += 2) {
			if (PRIO)
				scope_prio_round(wave_u, b >> 1);
			SCOPE_ISSUE_FULL(vb);
			SCOPE_BIN(va, yes);
			SCOPE_ISSUE_FULL(va);
			SCOPE_BIN(vb, yes);
		}
#undef SCOPE_ISSUE_FULL
	}
	for (; b < batches; b += 2) {
		if (PRIO)
			scope_prio_round(wave_u, b >> 1);
		SCOPE_ISSUE(vb, okb);
		SCOPE_BIN(va, oka);
		SCOPE_ISSUE(va, oka);
		if (b + 1 < batches)
			SCOPE_BIN(vb, okb);
	}
	if (PRIO)
		__builtin_amdgcn_s_setprio(0); /* the epilogues (flush, reduction, tickets) and whatever shares the SIMD run at the usual priority */
#undef SCOPE_ISSUE
#undef SCOPE_BIN
}

#ifndef SCOPE_ROWS
#define SCOPE_ROWS 1 /* 0: padded planes through the general form (A/B) */
#endif
/* Does a padded plane's slice run the ROWS form of scope_stream?  A row must be whole lane groups (every lane of a wave in one
 * column phase) and a lane group's share at least sixteen rows (streams are whole rows: a share of 8.4 rows -- one 4K frame
 * over 256 workgroups -- would be 8 rows here and 9 there). */
__host__ __device__ __forceinline__ bool scope_rows_mode(u32 linesize, u32 ncols, u32 cell_bytes, u32 rows, u32 slices, u32 glog2)
{
	return SCOPE_ROWS && linesize != ncols * cell_bytes && ncols % (64u >> glog2) == 0u && rows >= ((16u * slices) << glog2);
}

#ifndef VSS_RLE
#define VSS_RLE 1 /* 0: the fused vectorscope without its run-length mode (round 3's kernel; A/B with `make variant`) */
#endif
/* The mode lives in the batch form of the kernel only (four loads per lane per batch).  The form small launches run (two loads:
 * one frame per call, the ring of three) is dominated by its start-up, and carrying the mode's code cost it 2.5-4 % on noise and
 * natural content for +4 % (one frame) / +14 % (three) on screen content (visit c4/c5, profiles/round4/ab_fused_vectorscope.txt). */
#define VSS_RLE_IN(U) (VSS_RLE && (U) == 4)
#ifndef VSS_PACKED_STATE
#define VSS_PACKED_STATE 1 /* the fused kernel's absorb state in one register (vss_absorb_pairs1); 0: four (rounds 3-5; A/B) */
#endif
#ifndef VSS_RLE_LEAN
#define VSS_RLE_LEAN 1 /* 0: round 4's instruction sequence of the run-length modes (A/B) */
#endif
#ifndef VSS_RLE_KEYS
#define VSS_RLE_KEYS 1 /* 0: the plain (staged-plane) vectorscope without the run-length form of its absorb mode (A/B) */
#endif
#ifndef VSS_RK_LO
#define VSS_RK_LO 16u /* flat lanes of 64 at the vote: from here the run-length form ... */
#define VSS_RK_HI 60u /* ... up to here (above: the per-lane absorb mode, whose colour then rarely changes) */
#endif
#define VSS_RLE_KEYS_IN(U, FUSE) (VSS_RLE_KEYS && (U) == 4 && !(FUSE))
/* run-length mode's scratch behind the fused kernel's table: per wave, texels 1-3 of up to 64 lanes */
#define VSS_RLE_SCRATCH_WORDS_PER_WAVE 192u
#define VSS_RLE_SCRATCH_BYTES (16u * VSS_RLE_SCRATCH_WORDS_PER_WAVE * 4u)
template <int VECW, int U, bool NT, int MERGE, bool FUSE, int HASH, typename Pro>
__device__ __forceinline__ void vss_region(u32 *tab, const uint8_t *plane, u32 linesize, u32 xbyte0, u32 ncols, u32 rows,
					   u32 slice, u32 slices, u32 glog2, u32 rle_force, const ScopeK0 &k0, u32 &since_clamp,
					   VssLane &st, Pro &&prologue)
{
	if (ncols == 0)
		return;
	const u32 total = rows * ncols;
	const bool rows_mode = VECW == 4 && scope_rows_mode(linesize, ncols, 16u, rows, slices, glog2); /* uniform */
	const u32 c0 = rows_mode ? scope_slice_begin(rows, slice, slices) * ncols : scope_slice_begin_cells(total, slice, slices);
	const u32 c1 = rows_mode ? scope_slice_begin(rows, slice + 1, slices) * ncols : scope_slice_begin_cells(total, slice + 1, slices);
	auto epoch = [&](u32 px) {
		/* uniform across the workgroup: everybody takes the same branch */
		if (since_clamp + px > VSS_EPOCH_PIXELS) {
			vss_flush_lane<HASH>(tab, st); /* pending pixels belong to the epoch that ends here */
			__syncthreads();
			vss_clamp_pass(tab, VssTable<HASH>::words);
			__syncthreads();
			since_clamp = 0;
		}
		since_clamp += px;
	};
	u32 vote_clock = 0;          /* wave-uniform: every lane runs the same batches */
	u32 absorb_verdict = 0;      /* wave-uniform and kept in a scalar register (readfirstlane): a bool carried round the
				      * loop becomes a lane mask and every `if (absorb)` an exec-mask dance */
	if (VECW == 4 && FUSE && HASH == 2) {
		/* the pair form: chroma travels two pixels to a register from the matrix rows to the LDS operands */
		/* the lane's colour as U | U << 16 and V | V << 16; 0xffffffff (no pair looks like that) while it has none */
		u32 lane_u = st.colour == 0xffffffffu ? 0xffffffffu : scope_perm(0u, st.colour, 0x0c010c01u);
		u32 lane_v = st.colour == 0xffffffffu ? 0xffffffffu : scope_perm(0u, st.colour, 0x0c020c02u);
		/* VSS_PACKED_STATE (round 6): the same state in ONE register inside this region, S = count | U << 16 | V << 24
		 * (vss_absorb_pairs1, scope_device.h: four registers were copied at every join of the binning paths) */
		u32 S = VSS_PACKED_STATE && st.colour != 0xffffffffu ? (st.count | ((st.colour << 8) & 0xffff0000u)) : 0u;
		auto epoch_s = [&](u32 px) { /* `epoch` with the packed state */
			if (since_clamp + px > VSS_EPOCH_PIXELS) {
				vss_flush_packed<HASH>(tab, S);
				__syncthreads();
				vss_clamp_pass(tab, VssTable<HASH>::words);
				__syncthreads();
				since_clamp = 0;
			}
			since_clamp += px;
		};
		(void)lane_u;
		(void)lane_v;
		/* ---- RUN-LENGTH MODE: the wave's window as runs of flat cells --------------------------------------------------
		 * Screen capture -- OBS's daily bread -- is flat areas with a little single-texel detail.  The per-lane absorb mode
		 * below suits content on which a LANE's stream is one colour for long; but a lane's consecutive cells lie 2 KiB apart
		 * (a wave's loads are contiguous, its batches are not), so there a lane changes colour with almost every cell, some
		 * lane of the 64 holds detail in almost every cell, and the wave runs K0 on all four texels, the fast path, the
		 * colour change AND the mixed path for every cell: 26-28 VALU instructions per texel, 0.50 of the HBM peak
		 * (profiles/round3/pmc_sq_screen_p9.txt).  What IS coherent is the neighbourhood across lanes: the lanes of a lane
		 * group hold ADJACENT cells.  So, with no per-lane state at all:
		 *   - a cell is FLAT when its four texels agree in raw RGB (the vectorscope ignores alpha, src/vectorscope.c:217-238);
		 *     K0 runs on texel 0 of every lane only (one vote, 11 instructions);
		 *   - flat cells that continue the cell of the lane below (same key, same lane group: one DPP move and a compare)
		 *     add nothing; the HEAD of each run adds 4 x the run's length, which it reads off a ballot of the run breaks
		 *     (a 64-bit shift and a find-first-bit);
		 *   - the lanes whose cell is not flat add their texel 0 themselves and write texels 1-3 to a scratch row of the wave
		 *     in the LDS, packed (v_mbcnt gives the slot); the wave then converts and adds the packed texels 64 at a time --
		 *     any lane may add any texel -- so the detail of a whole wave-cell (about 7 lanes x 3 texels on the synthetic
		 *     screen content) costs ONE more K0 round instead of three.
		 * Exact: every texel is counted once, in the bin K0 gives it (same K0, same votes).  Two per-lane forms of the same
		 * idea were measured and dropped this round (profiles/round4/ab_fused_vectorscope.txt). */
		const u32 lane_id = threadIdx.x & 63u;
		const u32 run_lanes = min(64u >> glog2, 16u); /* a run lives inside a lane group and inside a DPP row */
		/* this lane's place among the wave's lanes and the wave's width, written so that the CPU emulation -- whose waves are ONE
		 * lane wide, with that lane's ballot bit at its thread's position -- gets 0 and 1 */
		const u32 lane_w = scope_lanes_below(~0ull), wave_w = (u32)__popcll(__ballot(1));
		u32 *rle_scr = tab + VssTable<HASH>::words + (threadIdx.x >> 6) * VSS_RLE_SCRATCH_WORDS_PER_WAVE;
		const u32 lane_tag = 0x40000000u | lane_id; /* no key looks like it (keys: bytes 0 and 3 zero, U >= 15: never zero either) */
		auto bin_rle = [&](const scope_v4(&q)[U], const bool(&ok)[U]) {
#pragma unroll
			for (int k = 0; k < U; k++) {
				const u32 p0 = q[k][0], p1 = q[k][1], p2 = q[k][2], p3 = q[k][3];
				/* flat: the four texels agree in B, G, R (bits that differ between p0 and any of the others, alpha masked off) */
				const bool flat = ok[k] && scope_bitop3<0xa8>(scope_bitop3<0x7e>(p0, p1, p2), p0 ^ p3, 0x00ffffffu) == 0u;
				u32 key0[1] = {p0};
				scope_k0_cell<1, true>(key0, k0, true, false); /* wave votes inside: every lane goes through */
				/* a flat cell's tag is its key, any other cell's the lane's own: equal tags <=> flat cells of one bin.  Round 6 (the
				 * three-scope kernel's run-length form carried back, `VSS_RLE_LEAN`): the tag is one select, the lane below comes with
				 * zero for the first lane of a DPP row (bound_ctrl: no register to prepare; a run may cross the lane groups of a row --
				 * equal bins merge whatever their places in the picture), the run's length is a 32-bit count of trailing zeros (a run
				 * ends within its sixteen lanes), and heads and rough cells add their texel 0 through ONE address computation. */
				const u32 tag = flat ? key0[0] : lane_tag;
				const u32 below = VSS_RLE_LEAN ? scope_lane_below0(tag) : scope_lane_below(tag, 0u); /* every lane takes part (the DPP move's contract) */
				const bool cont = tag == below && (VSS_RLE_LEAN || (flat && (lane_id & (run_lanes - 1u)) != 0u)); /* (round 4: runs inside a lane group) */
				const u64 breaks = ~__ballot(cont); /* every lane that continues nothing (and every lane there is none of) */
				const bool nf = ok[k] && !flat;
				const u64 rough = __ballot(nf);
				if (VSS_RLE_LEAN) {
					if (ok[k] && !cont) { /* the head of a run, or a rough cell's texel 0 */
						u32 n = 1u;
						if (flat) {
							const u64 above = ((breaks >> 1) | (1ull << 63)) >> lane_id; /* bit i: lane + 1 + i continues nothing */
							n = 4u * (u32)__builtin_ctz((u32)above) + 4u;
						}
						vss_add<HASH>(tab, key0[0], n);
					}
				} else if (flat && !cont) { /* the head of a run: the next break above this lane ends it */
					const u64 above = ((breaks >> 1) | (1ull << 63)) >> lane_id;
					vss_add<HASH>(tab, key0[0], 4u * (u32)__ffsll((long long)above));
				}
				if (rough != 0) { /* wave-uniform */
					const u32 n3 = 3u * (u32)__popcll(rough);
					if (nf) {
						if (!VSS_RLE_LEAN)
							vss_add<HASH>(tab, key0[0], 1);
						u32 *slot = rle_scr + 3u * scope_lanes_below(rough);
						slot[0] = p1;
						slot[1] = p2;
						slot[2] = p3;
					}
					/* a wave's LDS operations are performed in order, so the reads below see the writes above; the fence and
					 * the wave barrier (no instructions on gfx9) say so to the compiler, which may otherwise move or forward the
					 * plain stores and loads of lanes that only MAY alias */
					scope_wave_fence();
					for (u32 base = 0; base < n3; base += wave_w) {
						const u32 idx = base + lane_w;
						u32 key[1] = {idx < n3 ? rle_scr[idx] : 0u};
						scope_k0_cell<1, true>(key, k0, true, false);
						if (idx < n3)
							vss_add<HASH>(tab, key[0], 1);
					}
					scope_wave_fence(); /* ... and the next cell's stores stay behind these reads */
				}
			}
		};
		/* Which mode a wave runs.  The run-length mode is entered from the absorb vote below (chroma, after K0, every 64 texels
		 * per lane -- noise and natural video never pass it, so they never pay for the second question) when the same batch
		 * also passes on RAW texels: at least a quarter of the wave holds nothing but cells that are flat in RGB.  Content whose
		 * chroma is flat while its RGB is not (a black-and-white film) passes the first vote only and stays in the per-lane
		 * absorb mode.  Once in, the wave re-asks the raw question every 64 texels per lane and leaves when it fails. */
		u32 rle_on = rle_force, rle_clock = 0; /* wave-uniform, kept in scalar registers */
		auto rgb_vote = [&](const scope_v4(&q)[U], const bool(&ok)[U]) -> u32 {
			bool all_flat = true;
#pragma unroll
			for (int k = 0; k < U; k++)
				all_flat &= !ok[k] || (scope_bitop3<0xf6>(q[k][0] ^ q[k][1], q[k][0], q[k][2]) | (q[k][0] ^ q[k][3])) << 8 == 0u;
			return (u32)__builtin_amdgcn_readfirstlane((int)(__popcll(__ballot(all_flat)) >= 16));
		};
		auto bin_pairs = [&](const scope_v4(&q)[U], const bool(&ok)[U]) {
			if (VSS_RLE_IN(U) && MERGE == 3 && __builtin_expect(__builtin_amdgcn_readfirstlane(rle_on) != 0, false)) {
				if ((rle_clock++ & (16u / U - 1u)) == 0 && !rle_force)
					rle_on = rgb_vote(q, ok);
				if (__builtin_amdgcn_readfirstlane(rle_on) != 0) {
					bin_rle(q, ok);
					return;
				}
				/* left the mode: it keeps nothing in the lanes, so the forms below go on where they were */
			}
			u32 pu[U][2], pv[U][2];
#pragma unroll
			for (int k = 0; k < U; k++) {
				const u32 p[4] = {q[k][0], q[k][1], q[k][2], q[k][3]};
				scope_k0_cell_uv<2>(p, k0, pu[k], pv[k]);
			}
			bool absorb = MERGE == 1 || MERGE == 2;
			if (MERGE == 3) { /* the same vote as below */
				if ((vote_clock++ & (16u / U - 1u)) == 0) {
					bool flat = true;
#pragma unroll
					for (int k = 0; k < U; k++)
						flat &= !ok[k] || ((pu[k][0] == pu[k][1]) & (pv[k][0] == pv[k][1]) &
								   (pu[k][0] == scope_perm(0u, pu[k][0], 0x01000302u)) &
								   (pv[k][0] == scope_perm(0u, pv[k][0], 0x01000302u)));
					absorb_verdict = (u32)__builtin_amdgcn_readfirstlane((int)(__popcll(__ballot(flat)) >= 16));
					if (VSS_RLE_IN(U) && absorb_verdict != 0 && rgb_vote(q, ok) != 0) {
						rle_on = 2u; /* from the next batch on: run-length mode (this batch is still the absorb mode's, then it hands over) */
					}
				}
				absorb = __builtin_amdgcn_readfirstlane(absorb_verdict) != 0;
			}
			if (__builtin_expect(absorb, false)) {
#pragma unroll
				for (int k = 0; k < U; k++)
					if (ok[k]) {
						if (VSS_PACKED_STATE)
							vss_absorb_pairs1(tab, S, pu[k][0], pv[k][0], pu[k][1], pv[k][1]);
						else
							vss_absorb_pairs(tab, st, lane_u, lane_v, pu[k][0], pv[k][0], pu[k][1], pv[k][1]);
					}
				if (VSS_RLE_IN(U) && MERGE == 3 && __builtin_amdgcn_readfirstlane(rle_on) == 2u) {
					/* entering the run-length mode with the next batch: it keeps nothing in the lanes, so what the absorb mode
					 * holds goes to the table now (rare; every lane of the wave takes part) */
					if (VSS_PACKED_STATE)
						vss_flush_packed<HASH>(tab, S);
					else
						vss_flush_lane<HASH>(tab, st);
					rle_on = 1u;
					rle_clock = 1u;
				}
			} else {
#pragma unroll
				for (int k = 0; k < U; k++)
					if (ok[k]) {
						vss_add_pair(tab, pu[k][0], pv[k][0]);
						vss_add_pair(tab, pu[k][1], pv[k][1]);
					}
			}
		};
		if (VSS_PACKED_STATE) {
			if (linesize == ncols * 16u)
				scope_stream<4, U, NT, true>(plane, linesize, xbyte0, ncols, c0, c1, glog2, prologue, epoch_s, bin_pairs);
			else if (rows_mode)
				scope_stream<4, U, NT, false, true, false, true>(plane, linesize, xbyte0, ncols, c0, c1, glog2, prologue, epoch_s, bin_pairs);
			else
				scope_stream<4, U, NT, false>(plane, linesize, xbyte0, ncols, c0, c1, glog2, prologue, epoch_s, bin_pairs);
			/* back into the two-register form the kernel's other region and its last flush know */
			st.count = S & 0xffffu;
			st.colour = S ? ((S >> 16) << 8) : 0xffffffffu;
		} else if (linesize == ncols * 16u)
			scope_stream<4, U, NT, true>(plane, linesize, xbyte0, ncols, c0, c1, glog2, prologue, epoch, bin_pairs);
		else if (rows_mode)
			scope_stream<4, U, NT, false, true, false, true>(plane, linesize, xbyte0, ncols, c0, c1, glog2, prologue, epoch, bin_pairs);
		else
			scope_stream<4, U, NT, false>(plane, linesize, xbyte0, ncols, c0, c1, glog2, prologue, epoch, bin_pairs);
	} else if (VECW == 4) {
		u32 keys_rle = 0; /* wave-uniform */
		const u32 rk_lane = threadIdx.x & 63u, rk_run = min(64u >> glog2, 16u);
		(void)rk_lane;
		(void)rk_run;
		auto bin_cells = [&](const scope_v4(&q)[U], const bool(&ok)[U]) {
					       u32 idx[U][4];
#pragma unroll
					       for (int k = 0; k < U; k++) {
						       u32 p[4] = {q[k][0], q[k][1], q[k][2], q[k][3]};
						       if (FUSE)
							       scope_k0_cell<4, true>(p, k0, true, false); /* keys come out */
#pragma unroll
						       for (int j = 0; j < 4; j++)
							       idx[k][j] = FUSE ? p[j] : vss_bin(p[j]);
					       }
					       /* MERGE 3: a wave vote -- does at least a quarter of the wave hold nothing but
						* single-colour cells in this batch (16 pixels per lane)?  Then it sits on flat content
						* and runs in absorb mode; noisy video never passes the vote and only pays the plain
						* adds.  Content is spatially coherent, so the vote is taken on every fourth batch and
						* the verdict kept in between (either mode is exact; the choice is about speed only,
						* and a vote per cell or even per batch costs more than it saves). */
					       bool absorb = MERGE == 1 || MERGE == 2;
					       if (MERGE == 3) {
						       if ((vote_clock++ & (16u / U - 1u)) == 0) { /* every 64 pixels per lane */
							       bool flat = true;
#pragma unroll
							       for (int k = 0; k < U; k++)
								       flat &= !ok[k] || ((idx[k][0] == idx[k][1]) & (idx[k][1] == idx[k][2]) &
											  (idx[k][2] == idx[k][3]));
							       const u32 nflat = (u32)__popcll(__ballot(flat));
							       absorb_verdict = (u32)__builtin_amdgcn_readfirstlane((int)(nflat >= 16));
							       /* MIXED windows -- some lanes flat, some not: screen capture -- take the run-length form of the
								* absorb mode (below): no per-lane colour, which on such content changes with every cell */
							       if (VSS_RLE_KEYS_IN(U, FUSE)) {
								       const u32 mixed = (u32)__builtin_amdgcn_readfirstlane((int)((nflat >= VSS_RK_LO && nflat < VSS_RK_HI) || rle_force));
								       if (mixed && !keys_rle) /* entering: the lanes' counts go to the table (rare) */
									       vss_flush_lane<HASH>(tab, st);
								       keys_rle = mixed;
							       }
						       }
						       absorb = __builtin_amdgcn_readfirstlane(absorb_verdict) != 0;
					       }
					       if (VSS_RLE_KEYS_IN(U, FUSE) && MERGE == 3 && __builtin_expect(__builtin_amdgcn_readfirstlane(keys_rle) != 0, false)) {
						       /* Run-length form on keys (the fused kernel's bin_rle without K0): flat cells that continue the
							* cell of the lane below add nothing, the head of a run adds 4 x its length, the other cells add
							* their four texels. */
#pragma unroll
						       for (int k = 0; k < U; k++) {
							       const u32 i0 = idx[k][0];
							       const bool flat = ok[k] && ((i0 == idx[k][1]) & (i0 == idx[k][2]) & (i0 == idx[k][3]));
							       if (VSS_RLE_LEAN) { /* round 6: the lean sequence of bin_rle above */
								       const u32 tag = flat ? i0 : (0x40000000u | rk_lane); /* staged planes: a key may be zero -- so may a
													      * DPP row's first lane's fill, hence the lane test below */
								       const bool cont = tag == scope_lane_below0(tag) && (rk_lane & 15u) != 0u;
								       const u64 breaks = ~__ballot(cont);
								       if (ok[k] && !cont) {
									       u32 n = 1u;
									       if (flat) {
										       const u64 above = ((breaks >> 1) | (1ull << 63)) >> rk_lane;
										       n = 4u * (u32)__builtin_ctz((u32)above) + 4u;
									       }
									       vss_add<HASH>(tab, i0, n);
									       if (!flat) {
										       vss_add<HASH>(tab, idx[k][1], 1);
										       vss_add<HASH>(tab, idx[k][2], 1);
										       vss_add<HASH>(tab, idx[k][3], 1);
									       }
								       }
								       continue;
							       }
							       const u32 tag = flat ? (i0 | 0x80000000u) : (0x40000000u | rk_lane);
							       const u32 below = scope_lane_below(tag, 0u); /* every lane takes part (the DPP move's contract) */
							       const bool cont = flat && tag == below && (rk_lane & (rk_run - 1u)) != 0u;
							       const u64 breaks = ~__ballot(cont);
							       if (flat && !cont) {
								       const u64 above = ((breaks >> 1) | (1ull << 63)) >> rk_lane;
								       vss_add<HASH>(tab, i0, 4u * (u32)__ffsll((long long)above));
							       } else if (ok[k] && !flat) {
								       vss_add<HASH>(tab, i0, 1);
								       vss_add<HASH>(tab, idx[k][1], 1);
								       vss_add<HASH>(tab, idx[k][2], 1);
								       vss_add<HASH>(tab, idx[k][3], 1);
							       }
						       }
					       } else if (__builtin_expect(absorb, false)) {
#pragma unroll
						       for (int k = 0; k < U; k++)
							       if (ok[k])
								       vss_absorb4<HASH>(tab, st, idx[k][0], idx[k][1], idx[k][2], idx[k][3]);
					       } else {
#pragma unroll
						       for (int k = 0; k < U; k++)
							       if (ok[k]) {
								       vss_add<HASH>(tab, idx[k][0], 1);
								       vss_add<HASH>(tab, idx[k][1], 1);
								       vss_add<HASH>(tab, idx[k][2], 1);
								       vss_add<HASH>(tab, idx[k][3], 1);
							       }
					       }
		};
		if (linesize == ncols * 16u) /* uniform: planes whose rows carry no padding */
			scope_stream<4, U, NT, true>(plane, linesize, xbyte0, ncols, c0, c1, glog2, prologue, epoch, bin_cells);
		else if (rows_mode)
			scope_stream<4, U, NT, false, true, false, true>(plane, linesize, xbyte0, ncols, c0, c1, glog2, prologue, epoch, bin_cells);
		else
			scope_stream<4, U, NT, false>(plane, linesize, xbyte0, ncols, c0, c1, glog2, prologue, epoch, bin_cells);
	} else {
		scope_stream<1, U, NT, false>(plane, linesize, xbyte0, ncols, c0, c1, glog2, prologue, epoch,
				       [&](const u32(&q)[U], const bool(&ok)[U]) {
#pragma unroll
					       for (int k = 0; k < U; k++) {
						       u32 p0[1] = {q[k]};
						       if (FUSE)
							       scope_k0_cell<1, true>(p0, k0, true, false);
						       if (ok[k])
							       vss_add<HASH>(tab, FUSE ? p0[0] : vss_bin(p0[0]), 1);
					       }
				       });
	}
}

/* grid = n_frames * slices workgroups; unit = frame * slices + slice.
 * Region A: ncols_a cells of 4 pixels from byte 0 of each row (16-byte aligned rows only);
 * region B: ncols_b single pixels from byte xbyte_b (row tails, or whole rows when unaligned).
 *
 * pmask == NULL: dst[unit] is a complete clamped table (the final one when slices == 1).
 * pmask != NULL: a SPARSE partial -- only the 16-bin groups that hold a count are written, and thread t records which
 * of its groups g = t + k * blockDim.x (k < 8: blockDim.x >= 512) it wrote as bit k of its mask byte (layout below).
 * The slices of a one-frame launch see 1/256 of the picture each: on real video most of a slice's table is empty, and
 * even uniform noise reaches only the 57 % of the CbCr square that RGB maps to (scope_sat_reduce_sparse_kernel
 * reads back only what was written).  Measured alternatives -- every workgroup adding its table to one accumulator
 * with device-scope atomics (35 us for 256 dense tables: memory-side atomics run at ~7.5 G cache lines/s and count
 * twice the partials' bytes) -- are in profiles/round2/flush_microbench_r2a.txt. */
template <int U, bool NT, int MERGE, bool FUSE, int HASH>
__global__ __launch_bounds__(SCOPE_BLOCK) void scope_vss_kernel(ScopeFrameSet fs, u32 slices, u32 ncols_a, u32 ncols_b,
							       u32 xbyte_b, u32 glog2, u32 rle_force, ScopeK0 k0_arg, uint8_t *dst,
							       uint8_t *pmask, VssFold fold)
{
	const ScopeK0 k0 = FUSE ? scope_k0_pinned(k0_arg, true, false) : k0_arg;
	u32 *tab = scope_lds;
	scope_lds_is_at_zero(tab);
	const u32 unit = blockIdx.x;
	const u32 frame = unit / slices;
	const u32 slice = unit - frame * slices;
	const uint8_t *plane = fs.plane[frame];

	/* the table is zeroed with the first batch of loads already in flight (scope_stream) */
	bool zeroed = false; /* uniform */
	auto zero_table = [&]() {
		if (zeroed)
			return;
		zeroed = true;
		const uint4 zero = make_uint4(0, 0, 0, 0);
		for (u32 i = threadIdx.x * 4; i < VssTable<HASH>::words; i += blockDim.x * 4)
			*(uint4 *)&tab[i] = zero;
		__syncthreads();
	};

	SCOPE_STAMP(0);
	u32 since_clamp = 0;
	VssLane st = {0xffffffffu, 0};
	vss_region<4, U, NT, MERGE, FUSE, HASH>(tab, plane, fs.linesize, 0, ncols_a, fs.height, slice, slices, glog2, rle_force, k0,
						 since_clamp, st, zero_table);
	vss_region<1, U, NT, MERGE, FUSE, HASH>(tab, plane, fs.linesize, xbyte_b, ncols_b, fs.height, slice, slices, glog2,
						 rle_force, k0, since_clamp, st, zero_table);
	zero_table(); /* a slice without pixels */
	vss_flush_lane<HASH>(tab, st);
	__syncthreads();
	SCOPE_STAMP(1);

	/* flush: 16 consecutive bins of one row per lane per step = 8 table words, which the swizzle (a multiple of eight
	 * words) leaves contiguous: two 16-byte LDS reads, one coalesced 16-byte store.  Final table, sparse partial (blocked by table
	 * row: what one workgroup of the reduction reads -- a row of every slice -- is one contiguous run), or the slices of the
	 * frame meeting inside the launch: vss_write_out (scope_device.h).  The arrival flag is the LDS word behind the table. */
	vss_write_out(
		[&](u32 g, uint4 &lo, uint4 &hi) {
			const u32 row = 255u - (g >> 4); /* table row v of output row g >> 4 */
			const u32 w0 = HASH == 2 ? row * VSS_PITCH2 + ((g & 15u) << 3)
						 : (row << 7) + (((g & 15u) << 3) ^ (HASH == 1 ? 0u : ((row & 15u) << 3)));
			lo = *(const uint4 *)&tab[w0];
			hi = *(const uint4 *)&tab[w0 + 4];
		},
		frame, slice, slices, dst, pmask, fold, &tab[VssTable<HASH>::words]);
	SCOPE_STAMP(2);
}

/* ------------------------------------------------------------------------------------------
 * K3 histogram
 *
 * LDS table: u32 [3 channels][256 values][32 replicas]; a lane always uses replica lane&31, so
 * the 32 lanes of an LDS lane group hit 32 different banks whatever the pixel values are
 * (flat frames included).  Counters are 32 bit and never saturate (src/histogram.c:388-393).
 * ---------------------------------------------------------------------------------------- */

#define HIS_WORDS (3u * 256u * 32u)

template <bool FUSE> __device__ __forceinline__ void his_add(u32 *tab, u32 px, u32 chmask, u32 rep)
{
	if (!FUSE && (px >> 24) == 0)
		return; /* src/histogram.c:386-387; the K0 pass writes alpha 255 whatever the source holds */
	if (chmask & 1u)
		atomicAdd(&tab[((0u * 256u + (px & 0xffu)) << 5) + rep], 1u);
	if (chmask & 2u)
		atomicAdd(&tab[((1u * 256u + ((px >> 8) & 0xffu)) << 5) + rep], 1u);
	if (chmask & 4u)
		atomicAdd(&tab[((2u * 256u + ((px >> 16) & 0xffu)) << 5) + rep], 1u);
}

template <int VECW, int U, bool NT, bool FUSE, typename Pro>
__device__ __forceinline__ void his_region(u32 *tab, const uint8_t *plane, u32 linesize, u32 xbyte0, u32 ncols, u32 rows,
					   u32 slice, u32 slices, u32 chmask, const ScopeK0 &k0, Pro &&prologue)
{
	if (ncols == 0)
		return;
	const u32 total = rows * ncols;
	const bool rows_mode = VECW == 4 && scope_rows_mode(linesize, ncols, 16u, rows, slices, 0); /* uniform */
	const u32 c0 = rows_mode ? scope_slice_begin(rows, slice, slices) * ncols : scope_slice_begin_cells(total, slice, slices);
	const u32 c1 = rows_mode ? scope_slice_begin(rows, slice + 1, slices) * ncols : scope_slice_begin_cells(total, slice + 1, slices);
	const u32 rep = threadIdx.x & 31u;
	const bool need_uv = (chmask & 5u) != 0, need_y = (chmask & 2u) != 0;
	auto nothing = [](u32) {};
	if (VECW == 4) {
		auto bin_cells = [&](const scope_v4(&q)[U], const bool(&ok)[U]) {
#pragma unroll
					       for (int k = 0; k < U; k++) {
						       u32 p[4] = {q[k][0], q[k][1], q[k][2], q[k][3]};
						       if (FUSE)
							       scope_k0_cell<4>(p, k0, need_uv, need_y);
#pragma unroll
						       for (int j = 0; j < 4; j++)
							       if (ok[k])
								       his_add<FUSE>(tab, p[j], chmask, rep);
					       }
		};
		if (linesize == ncols * 16u) /* uniform: rows without padding */
			scope_stream<4, U, NT, true, FUSE, true>(plane, linesize, xbyte0, ncols, c0, c1, 0, prologue, nothing, bin_cells);
		else if (rows_mode)
			scope_stream<4, U, NT, false, FUSE, true, true>(plane, linesize, xbyte0, ncols, c0, c1, 0, prologue, nothing, bin_cells);
		else
			scope_stream<4, U, NT, false, false, true>(plane, linesize, xbyte0, ncols, c0, c1, 0, prologue, nothing, bin_cells);
	} else {
		scope_stream<1, U, NT, false>(plane, linesize, xbyte0, ncols, c0, c1, 0, prologue, nothing,
				       [&](const u32(&q)[U], const bool(&ok)[U]) {
#pragma unroll
					       for (int k = 0; k < U; k++) {
						       u32 p0[1] = {q[k]};
						       if (FUSE)
							       scope_k0_cell<1>(p0, k0, need_uv, need_y);
						       if (ok[k])
							       his_add<FUSE>(tab, p0[0], chmask, rep);
					       }
				       });
	}
}

/* acc == NULL: dst[unit] receives this workgroup's counts (the final table when slices == 1, else a partial for
 * scope_sum_reduce_kernel).  acc != NULL (slices > 1): the slices of a frame are combined inside the launch -- every
 * workgroup adds its non-zero counts to acc[frame] with device-scope atomics (at most 768 per workgroup: 12 wave
 * instructions, 0.2 us for a whole one-frame launch of 256 workgroups), and the one that arrives last moves the sums
 * to dst[frame] with atomic exchanges, which also leaves the accumulator zero for the next launch
 * (scope_last_arrival).  One launch, no partial tables, whatever the number of slices. */
template <int U, bool NT, bool FUSE>
__global__ __launch_bounds__(SCOPE_BLOCK) void scope_his_kernel(ScopeFrameSet fs, u32 slices, u32 ncols_a, u32 ncols_b,
							       u32 xbyte_b, u32 chmask, ScopeK0 k0_arg, u32 *dst, u32 *acc, u32 *tickets)
{
	const ScopeK0 k0 = FUSE ? scope_k0_pinned(k0_arg, (chmask & 5u) != 0, (chmask & 2u) != 0) : k0_arg;
	u32 *tab = scope_lds;
	const u32 unit = blockIdx.x;
	const u32 frame = unit / slices;
	const u32 slice = unit - frame * slices;
	const uint8_t *plane = fs.plane[frame];

	bool zeroed = false; /* uniform */
	auto zero_table = [&]() {
		if (zeroed)
			return;
		zeroed = true;
		const uint4 zero = make_uint4(0, 0, 0, 0);
		for (u32 i = threadIdx.x * 4; i < HIS_WORDS; i += blockDim.x * 4)
			*(uint4 *)&tab[i] = zero;
		__syncthreads();
	};

	his_region<4, U, NT, FUSE>(tab, plane, fs.linesize, 0, ncols_a, fs.height, slice, slices, chmask, k0, zero_table);
	his_region<1, U, NT, FUSE>(tab, plane, fs.linesize, xbyte_b, ncols_b, fs.height, slice, slices, chmask, k0, zero_table);
	zero_table();
	__syncthreads();

	/* fold the 32 replicas; lane t starts at replica t so a lane group reads 32 banks */
	u32 *out = acc ? acc + (size_t)frame * SCOPE_HIS_WORDS : dst + (size_t)unit * SCOPE_HIS_WORDS;
	for (u32 t = threadIdx.x; t < SCOPE_HIS_WORDS; t += blockDim.x) {
		const u32 value = t >> 2, slot = t & 3u;
		u32 sum = 0;
		if (slot < 3) {
			const u32 ch = 2u - slot; /* slot 0 = R = byte2, 1 = G = byte1, 2 = B = byte0 */
			const u32 base = (ch * 256u + value) << 5;
			for (u32 j = 0; j < 32; j++)
				sum += tab[base + ((j + value) & 31u)];
		}
		if (!acc)
			out[t] = sum;
		else if (sum)
			__hip_atomic_fetch_add(&out[t], sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}

	if (acc && scope_last_arrival(&tickets[frame], slices, &tab[0])) {
		for (u32 t = threadIdx.x; t < SCOPE_HIS_WORDS; t += blockDim.x)
			dst[(size_t)frame * SCOPE_HIS_WORDS + t] =
				__hip_atomic_exchange(&out[t], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (threadIdx.x == 0)
			__hip_atomic_store(&tickets[frame], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
}

/* ------------------------------------------------------------------------------------------
 * K2 waveform
 *
 * A unit is a strip of 64 columns over a band of rows of one frame.  LDS table per enabled channel: 8192 words =
 * 16-bit counters [256 values][64 columns]; a column can receive at most `rows in band` (< 65536, the C-ABI cuts
 * taller frames into bands) increments, so 16 bits never overflow.  One persistent workgroup per CU walks units
 * g, g + G, ... with one continuous load stream (details at the kernel).
 * ---------------------------------------------------------------------------------------- */

#define WVS_CH_WORDS 8192u
#ifndef WVS_U
#define WVS_U 2 /* row pairs per wave per batch; two batches in flight (measured: 2 beats 4 by 4 %, 1 loses 17 %) */
#endif
/* Strip widths (SW).  64 columns is the batch shape.  A launch of one or two frames has too few 64-column strips to
 * fill the chip (3840 / 64 = 60) and used to be cut into row bands whose dense partial tables a second kernel summed
 * (1.85 x the algorithmic traffic).  With SW = 16 a 4K frame is 240 units and needs neither: a wave then reads EIGHT
 * rows of 64 bytes at a time, and row r of the eight counts in replica r & 3 of the strip's table, which makes the
 * LDS side -- word value*32 + (lane & 31), 32 different banks per half wave -- the very same as for SW = 64; the
 * flush adds the four replicas up. */
#ifndef WVS_DEEP_SETS
#define WVS_DEEP_SETS 4 /* batches issued ahead of a flush on top of the two of the steady state (DEEP) */
#endif
#define WVS_NARROW 16
#ifndef WVS_NARROW_U
#define WVS_NARROW_U 1 /* row groups per wave per batch.  A unit is 270 groups on 16 waves = 17 each, and a wave's batches are
			* rounded up to an even number: 18 at 1, 20 at 2, 24 at 4 -- 14.1 / 14.4 / 15.1 us per 4K frame (6: 15.4, 9: 16.3) */
#endif

/* Counter of (value, column c of the strip): 16-bit half (c & 1) of word value*32 + (c >> 1).
 *
 * A wave reads TWO rows at a time: lanes 0-31 the first, lanes 32-63 the second, each lane two adjacent
 * texels (one 8-byte load when WIDE).  One atomic instruction therefore touches, per 32-lane half, 32
 * different words of one 32-word group = 32 different banks whatever the pixel values are; the two
 * halves go through the LDS in separate passes.  Conflict-free by construction.
 *
 * CH = which bytes of the texel are counted (bit c -> byte c), a compile-time constant so that the
 * per-pixel code has no channel branches.  Row-pair pointers and validity are wave-uniform (SGPRs,
 * scalar branches); per lane there is one constant 32-bit offset, the alpha tests and the atomics.
 * WIDE needs width % 4 == 0, 8-byte aligned rows and a 16-byte aligned table; otherwise texels are
 * loaded and stored one by one. */
#ifndef WVS_DOT4
#define WVS_DOT4 1 /* the table address of an add by v_dot4_u32_u8, one instruction instead of two (scope_trio_kernel's gain: +4.7 %).
		    * Here: the luma mode (fused K0, VALU-bound) +1.7 %, one channel +0.6 %, RGB 0 %; one frame -0.7 %.  The histogram's
		    * kernels LOSE 1.3 ... 1.7 % with it and do not have it (visit t2, profiles/round4/ab_dot4_addresses.txt) */
#endif
template <int CH> __device__ __forceinline__ void wvs_count(u32 *tab, u32 px, u32 wbyte, u32 inc)
{
	constexpr u32 slot1 = (CH & 1), slot2 = slot1 + ((CH >> 1) & 1);
	uint8_t *t = (uint8_t *)tab + wbyte;
	if (WVS_DOT4) { /* value * 128 + the lane's offset in one v_dot4_u32_u8 (scope_udot4) */
		uint8_t *t0 = (uint8_t *)tab;
		if (CH & 1)
			atomicAdd((u32 *)(t0 + scope_udot4(px, 128u, wbyte)), inc);
		if (CH & 2)
			atomicAdd((u32 *)(t0 + slot1 * WVS_CH_WORDS * 4u + scope_udot4(px, 128u << 8, wbyte)), inc);
		if (CH & 4)
			atomicAdd((u32 *)(t0 + slot2 * WVS_CH_WORDS * 4u + scope_udot4(px, 128u << 16, wbyte)), inc);
		return;
	}
	if (CH & 1)
		atomicAdd((u32 *)(t + ((px & 0xffu) << 7)), inc);
	if (CH & 2)
		atomicAdd((u32 *)(t + slot1 * WVS_CH_WORDS * 4u + (((px >> 8) & 0xffu) << 7)), inc);
	if (CH & 4)
		atomicAdd((u32 *)(t + slot2 * WVS_CH_WORDS * 4u + (((px >> 16) & 0xffu) << 7)), inc);
}

/* One (frame, band, strip) unit as a wave sees it; everything here is wave-uniform. */
struct WvsUnit {
	const uint8_t *rows; /* first row of the band (always a readable address) */
	u32 frame, band, strip;
	u32 nrows;  /* rows in the band */
	u32 npairs; /* row groups = nrows / (rows per wave instruction: 2, or 8 for narrow strips); the rows left over are handled apart */
	u32 slots;  /* batches per wave, rounded up to an even number >= 2 (slots past the data load nothing useful) */
};

template <int SW> __device__ __forceinline__ WvsUnit wvs_unit(const ScopeFrameSet &fs, u32 unit, u32 total, u32 strips, u32 bands,
					    u32 nwaves)
{
	constexpr u32 RW = 128u / SW, U = SW == WVS_NARROW ? WVS_NARROW_U : WVS_U;
	WvsUnit u;
	if (unit >= total) { /* past the end: two empty slots */
		u.rows = fs.plane[0];
		u.frame = u.band = u.strip = u.nrows = u.npairs = 0;
		u.slots = 2;
		return u;
	}
	/* Narrow strips read half a cache line per row: the workgroup that reads the other half should sit on the same
	 * XCD (same L2) and run at the same time.  Workgroups b and b + 8 do (observed placement: XCD = b % 8; a matter of
	 * speed only), so they take neighbouring strips. */
	if (SW == WVS_NARROW && unit < (total & ~15u))
		unit = (unit & ~15u) + ((unit & 7u) << 1) + ((unit >> 3) & 1u);
	const u32 per_frame = strips * bands;
	u.frame = unit / per_frame;
	const u32 rem = unit - u.frame * per_frame;
	u.band = rem / strips;
	u.strip = rem - u.band * strips;
	const u32 r0 = scope_slice_begin(fs.height, u.band, bands);
	u.nrows = scope_slice_begin(fs.height, u.band + 1, bands) - r0;
	u.rows = fs.plane[u.frame] + (size_t)r0 * fs.linesize;
	u.npairs = u.nrows / RW;
	const u32 nbatch = (u.npairs + U - 1) / U;
	u.slots = max(2u, ((nbatch + nwaves - 1) / nwaves + 1u) & ~1u);
	return u;
}

/* Persistent workgroups: workgroup g takes units g, g + gridDim.x, ...  The loads of a wave form ONE
 * stream over all its units -- the first batches of the next unit are issued before the current unit's
 * table is flushed and stay in flight across the flush and the zero-fill -- so the memory pipeline does
 * not drain at every unit boundary (a unit is only ~0.5 MB).  Batch n of a unit = row pairs 4n..4n+3;
 * wave w takes batches w, w + nwaves, ... */

/* DEEP (batches: every unit has the same, sufficiently many, slots): a unit's flush -- barrier, 96 KiB of LDS read, 64 KiB
 * stored, zero-fill, barrier -- takes the workgroup 3-4 us, and the two batches of the steady state (32 KiB per CU, the
 * depth HBM likes best) have arrived after one: the CU's share of the memory pipeline then idles for the rest (measured
 * with the flush compiled out: 5.08 instead of 6.08 us per 4K frame, of which the table's bytes explain 0.6).  So
 * WVS_DEEP_SETS more batches of the next unit are issued just before the flush, into register sets of their own, and
 * binned first thing after it without re-issuing: the queue is deep across the flush and decays to two again.
 * Same-process A/B, 64 x 4K RGB: 6078 -> 6210 GB/s on noise, 6171 -> 6215 on natural content; two extra sets do the
 * same as four, six and eight lose (6033, 5894).  What is left of the flush (measured by compiling parts of it out):
 * its LDS reads, packing and stores 2.3 %, its two barriers 2.2 %, the zero-fill 0.8 % -- the table's own bytes are
 * a tenth of the traffic and stay. */
template <int CH, bool FUSE, bool WIDE, bool HIS, int SW, bool DEEP>
__global__ __launch_bounds__(SCOPE_BLOCK) void scope_wvs_kernel(ScopeFrameSet fs, u32 strips, u32 bands, ScopeK0 k0_arg,
							       uint8_t *dst, ScopeWvsHis his)
{
	const ScopeK0 k0 = FUSE ? scope_k0_pinned(k0_arg, (CH & 5) != 0, (CH & 2) != 0) : k0_arg;
	static_assert(SW == 64 || (SW == WVS_NARROW && WIDE), "narrow strips exist in the vector shape only");
	constexpr int U = SW == WVS_NARROW ? WVS_NARROW_U : WVS_U;
	constexpr u32 LR = SW / 2u;   /* lanes per row (two texels each) */
	constexpr u32 RW = 64u / LR;  /* rows per wave instruction */
	constexpr u32 slot0 = 0, slot1 = (CH & 1), slot2 = slot1 + ((CH >> 1) & 1), nch = slot2 + ((CH >> 2) & 1);
	constexpr bool need_uv = (CH & 5) != 0, need_y = (CH & 2) != 0;

	u32 *tab = scope_lds;
	const u32 W = fs.width;
	const u32 total = fs.n_frames * strips * bands;
	const u32 lane = threadIdx.x & 63u;
	const u32 pair = lane & (LR - 1u); /* columns 2*pair, 2*pair+1 of the strip */
	const u32 second = lane / LR;      /* which row of the row group */
	const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const u32 nwaves = blockDim.x >> 6;
	/* table column of the lane: the strip's column pair for SW = 64; for narrow strips replica (row & 3) of it, which
	 * is again lane & 31 -- each half wave covers one 32-word value group whatever the strip width */
	const u32 wbyte = (lane & 31u) << 2;
	const u32 second_off = second * fs.linesize;
	const size_t pair_bytes = (size_t)RW * fs.linesize;

	/* ---- issue side: where the wave's load stream is ------------------------------------------ */
	u32 unit_i = blockIdx.x, slot_i = 0;
	WvsUnit ui = wvs_unit<SW>(fs, unit_i, total, strips, bands, nwaves);
	u32 lo0 = 0, lo1 = 0; /* per-lane byte offsets inside a row pair of the issue-side unit */
#define WVS_LANE_OFFSETS()                                                                                    \
	do {                                                                                                  \
		const u32 xs_ = ui.strip * (u32)SW, x0_ = xs_ + 2u * pair;                                    \
		lo0 = ((x0_ < W) ? x0_ : xs_) * 4u + second_off;                                              \
		lo1 = ((x0_ + 1u < W) ? x0_ + 1u : xs_) * 4u + second_off;                                    \
	} while (0)
	WVS_LANE_OFFSETS();

	u32 va[U][2], vb[U][2];
	bool oka[U], okb[U];
	constexpr int NX = DEEP ? WVS_DEEP_SETS : 1;
	u32 vx[NX][U][2];
	bool okx[NX][U];

	/* Unconditional loads keep the vmcnt bookkeeping exact (see scope_stream); a slot past the data reads
	 * the 8 bytes at the band's first texel with every lane, i.e. one cache line. */
#define WVS_ISSUE(v, ok)                                                                                      \
	do {                                                                                                  \
		if (slot_i == ui.slots) {                                                                     \
			unit_i += gridDim.x;                                                                  \
			slot_i = 0;                                                                           \
			ui = wvs_unit<SW>(fs, unit_i, total, strips, bands, nwaves);                          \
			WVS_LANE_OFFSETS();                                                                   \
		}                                                                                             \
		const u32 first_ = (wave + slot_i * nwaves) * (u32)U;                                         \
		_Pragma("unroll") for (int k = 0; k < U; k++)                                                 \
		{                                                                                             \
			ok[k] = first_ + (u32)k < ui.npairs;                                                  \
			const uint8_t *p_ = ui.rows + (ok[k] ? (size_t)(first_ + (u32)k) * pair_bytes : (size_t)0); \
			if (WIDE) {                                                                           \
				const scope_u2 t_ = scope_load_row2_nt(p_, ok[k] ? lo0 : 0u);                 \
				v[k][0] = t_.x;                                                               \
				v[k][1] = t_.y;                                                               \
			} else {                                                                              \
				v[k][0] = scope_load_row_nt(p_, ok[k] ? lo0 : 0u);                            \
				v[k][1] = scope_load_row_nt(p_, ok[k] ? lo1 : 0u);                            \
			}                                                                                     \
		}                                                                                             \
		slot_i++;                                                                                     \
	} while (0)
#define WVS_BIN(v, ok)                                                                                        \
	do {                                                                                                  \
		_Pragma("unroll") for (int k = 0; k < U; k++)                                                 \
		{                                                                                             \
			if (!ok[k])                                                                           \
				continue;                                                                     \
			u32 pp_[2] = {v[k][0], v[k][1]};                                                      \
			if (FUSE)                                                                             \
				scope_k0_cell<2>(pp_, k0, need_uv, need_y);                                   \
			const u32 p0 = pp_[0], p1 = pp_[1];                                                   \
			if (live0 && (FUSE || (p0 >> 24) != 0)) /* src/waveform.c:247-248; K0 writes alpha 255 */ \
				wvs_count<CH>(tab, p0, wbyte, 1u);                                            \
			if (live1 && (FUSE || (p1 >> 24) != 0))                                               \
				wvs_count<CH>(tab, p1, wbyte, 65536u);                                        \
		}                                                                                             \
	} while (0)

	WVS_ISSUE(va, oka);
	WVS_ISSUE(vb, okb);

	/* ---- bin side: one unit at a time, two slots behind the load stream ------------------------
	 * `steady` = a flush precedes this unit.  Its stores were issued AFTER the two batches now in flight, so
	 * with the first slot pair peeled the compiler can count them (vmcnt(11..8) instead of (7..4)) and the
	 * wave does not wait for the table write-out before it goes on binning. */
	auto run_unit = [&](const u32 unit_b, auto steady) {
		const WvsUnit ub = wvs_unit<SW>(fs, unit_b, total, strips, bands, nwaves);
		const u32 xs = ub.strip * (u32)SW; /* always < W */
		const u32 x0 = xs + 2u * pair;
		const bool live0 = x0 < W, live1 = x0 + 1u < W;

		const uint4 zero = make_uint4(0, 0, 0, 0);
		for (u32 i = threadIdx.x * 4; i < nch * WVS_CH_WORDS + (HIS ? 768u : 0u); i += blockDim.x * 4)
			*(uint4 *)&tab[i] = zero;
		__syncthreads();

		u32 sl = 0;
		if (decltype(steady)::value) {
			if (DEEP) { /* slots 2 .. 2 + NX - 1 of this unit, issued before the previous flush (rows commute) */
#pragma unroll
				for (int e = 0; e < NX; e++)
					WVS_BIN(vx[e], okx[e]);
				sl = NX;
			}
			WVS_BIN(va, oka);
			WVS_ISSUE(va, oka);
			WVS_BIN(vb, okb);
			WVS_ISSUE(vb, okb);
			sl += 2;
		}
		for (; sl < ub.slots; sl += 2) {
			if (WVS_PRIO && SW == 16) /* the narrow-strip form only (one frame per call: +1.7 %); batches of 64-column strips LOSE 2 % to it (visit p2) */
				scope_prio_round((u32)__builtin_amdgcn_readfirstlane((int)wave), sl >> 1); /* fair shares of the SIMD: scope_device.h */
			WVS_BIN(va, oka);
			WVS_ISSUE(va, oka);
			WVS_BIN(vb, okb);
			WVS_ISSUE(vb, okb);
		}
		if (WVS_PRIO && SW == 16)
			__builtin_amdgcn_s_setprio(0); /* the flush and the next unit's start at the usual priority */
		if (DEEP) { /* va, vb hold slots 0, 1 of the next unit; these are its slots 2 .. */
#pragma unroll
			for (int e = 0; e < NX; e++)
				WVS_ISSUE(vx[e], okx[e]);
		}
		/* the rows a band has beyond whole row groups (an odd last row; up to seven for narrow strips): wave 0 */
		const u32 left = ub.nrows - ub.npairs * RW;
		if (left && wave == 0 && second < left) {
			const uint8_t *p = ub.rows + (size_t)(ub.npairs * RW + second) * fs.linesize;
			u32 p0 = *(const u32 *)(p + (live0 ? x0 : xs) * 4u), p1 = *(const u32 *)(p + (live1 ? x0 + 1u : xs) * 4u);
			if (FUSE) {
				p0 = scope_k0_texel(p0, k0, need_uv, need_y);
				p1 = scope_k0_texel(p1, k0, need_uv, need_y);
			}
			if (live0 && (FUSE || (p0 >> 24) != 0))
				wvs_count<CH>(tab, p0, wbyte, 1u);
			if (live1 && (FUSE || (p1 >> 24) != 0))
				wvs_count<CH>(tab, p1, wbyte, 65536u);
		}
		__syncthreads();

		/* flush: lane -> 4 adjacent columns of one output row (255 - value), 16 B per store.
		 * Output texel bytes B,G,R,X = channels 0,1,2 and 0 (src/waveform.c:250-254; X never written).
		 * Streaming (nt) stores: the table is not read again here and must not displace the frames in L2. */
		uint8_t *table = dst + ((size_t)ub.frame * bands + ub.band) * ((size_t)SCOPE_WVS_ROWS * W * 4);
		const u32 xg = threadIdx.x & 15u; /* columns 4*xg .. 4*xg+3 of the strip = words 2*xg, 2*xg+1 */
		u32 *hsum = tab + nch * WVS_CH_WORDS; /* HIS: [3 slots][256 values] totals of this unit */
		/* n[c] = this lane's four columns of channel slot c added up (HIS) */
		auto gather = [&](u32 value, u32 g, u32(&o)[4], u32(&n)[3]) {
			o[0] = o[1] = o[2] = o[3] = 0;
			n[0] = n[1] = n[2] = 0;
			if (CH & 1) {
				const uint2 c = *(const uint2 *)&tab[slot0 * WVS_CH_WORDS + (value << 5) + 2u * g];
				o[0] |= min(c.x & 0xffffu, 255u);
				o[1] |= min(c.x >> 16, 255u);
				o[2] |= min(c.y & 0xffffu, 255u);
				o[3] |= min(c.y >> 16, 255u);
				n[0] = (c.x & 0xffffu) + (c.x >> 16) + (c.y & 0xffffu) + (c.y >> 16);
			}
			if (CH & 2) {
				const uint2 c = *(const uint2 *)&tab[slot1 * WVS_CH_WORDS + (value << 5) + 2u * g];
				o[0] |= min(c.x & 0xffffu, 255u) << 8;
				o[1] |= min(c.x >> 16, 255u) << 8;
				o[2] |= min(c.y & 0xffffu, 255u) << 8;
				o[3] |= min(c.y >> 16, 255u) << 8;
				n[1] = (c.x & 0xffffu) + (c.x >> 16) + (c.y & 0xffffu) + (c.y >> 16);
			}
			if (CH & 4) {
				const uint2 c = *(const uint2 *)&tab[slot2 * WVS_CH_WORDS + (value << 5) + 2u * g];
				o[0] |= min(c.x & 0xffffu, 255u) << 16;
				o[1] |= min(c.x >> 16, 255u) << 16;
				o[2] |= min(c.y & 0xffffu, 255u) << 16;
				o[3] |= min(c.y >> 16, 255u) << 16;
				n[2] = (c.x & 0xffffu) + (c.x >> 16) + (c.y & 0xffffu) + (c.y >> 16);
			}
		};
		/* the 16 lanes that share `value` (one DPP row) add their column sums; one lane per row and channel hands the
		 * total to LDS (a value has one owner row per unit, the LDS add is an atomic only so that the one-lane waves
		 * of the CPU emulation, where every lane is its own row, add up as well) */
		auto tally = [&](u32 value, bool real, const u32(&n)[3]) {
			if (CH & 1)
				scope_row16_add(&hsum[0u * 256u + value], real ? n[0] : 0u);
			if (CH & 2)
				scope_row16_add(&hsum[1u * 256u + value], real ? n[1] : 0u);
			if (CH & 4)
				scope_row16_add(&hsum[2u * 256u + value], real ? n[2] : 0u);
		};
		if (SW == WVS_NARROW) {
			/* Narrow strip: 256 values x 4 column groups = one (value, group) per thread.  The sixteen lanes of a
			 * DPP row take four consecutive values x four groups and read the four replicas in rotated order --
			 * lane (rho, g) reads replica (rho + j) & 3 at step j -- so at every step the sixteen lanes cover all 32
			 * banks.  The replica counts of a column add up to at most the band's rows (< 65536): the packed
			 * 16-bit halves cannot carry. */
			const u32 groups = (min(W - xs, (u32)SW)) >> 2;
			const u32 g4 = threadIdx.x & 3u, rho = (threadIdx.x >> 2) & 3u;
			const u32 value = ((threadIdx.x >> 4) << 2) + rho;
			const u32 g = min(g4, groups - 1u);
			uint2 acc[3] = {make_uint2(0, 0), make_uint2(0, 0), make_uint2(0, 0)};
#pragma unroll
			for (u32 j = 0; j < 4; j++) {
				const u32 w = (value << 5) + (((rho + j) & 3u) << 3) + 2u * g;
				if (CH & 1) {
					const uint2 c = *(const uint2 *)&tab[slot0 * WVS_CH_WORDS + w];
					acc[0].x += c.x;
					acc[0].y += c.y;
				}
				if (CH & 2) {
					const uint2 c = *(const uint2 *)&tab[slot1 * WVS_CH_WORDS + w];
					acc[1].x += c.x;
					acc[1].y += c.y;
				}
				if (CH & 4) {
					const uint2 c = *(const uint2 *)&tab[slot2 * WVS_CH_WORDS + w];
					acc[2].x += c.x;
					acc[2].y += c.y;
				}
			}
			u32 o[4] = {0, 0, 0, 0};
#pragma unroll
			for (u32 c = 0; c < 3; c++) {
				if (!((CH >> c) & 1))
					continue;
				o[0] |= min(acc[c].x & 0xffffu, 255u) << (8u * c);
				o[1] |= min(acc[c].x >> 16, 255u) << (8u * c);
				o[2] |= min(acc[c].y & 0xffffu, 255u) << (8u * c);
				o[3] |= min(acc[c].y >> 16, 255u) << (8u * c);
				if (HIS && g4 < groups) /* lanes past the right edge repeat a column group */
					atomicAdd(&hsum[c * 256u + value],
						  (acc[c].x & 0xffffu) + (acc[c].x >> 16) + (acc[c].y & 0xffffu) + (acc[c].y >> 16));
			}
			const scope_v4 t = {o[0], o[1], o[2], o[3]};
			if (WVS_NARROW_STORE == 2)
				scope_llvm_raw_buffer_store_v4i32(scope_i4{(int)o[0], (int)o[1], (int)o[2], (int)o[3]}, scope_buf_make(table, SCOPE_WVS_ROWS * W * 4u),
								  (int)(((xs + (g << 2)) + (255u - value) * W) * 4u), 0, 16);
			else
				__builtin_nontemporal_store(t, (scope_v4 *)(table + (size_t)(xs + (g << 2)) * 4) + (size_t)(255u - value) * (W >> 2));
		} else if (WIDE) {
			/* width % 4 == 0 and a 16-byte aligned table: a fixed number of unconditional vector stores per
			 * lane (lanes past the right edge repeat the strip's last column group), so the waits of the
			 * next unit's first batch can be counted exactly and do not include these stores. */
			const u32 groups = (min(W - xs, SCOPE_WVS_STRIP)) >> 2;
			const u32 g = min(xg, groups - 1u);
			scope_v4 *ocol = (scope_v4 *)(table + (size_t)(xs + (g << 2)) * 4);
#pragma unroll
			for (u32 i = 0; i < 256u / (SCOPE_BLOCK / 16u); i++) {
				const u32 value = (threadIdx.x >> 4) + i * (SCOPE_BLOCK / 16u);
				u32 o[4], n[3];
				gather(value, g, o, n);
				const scope_v4 t = {o[0], o[1], o[2], o[3]};
				constexpr int STORE = WVS_STORE_NT == 4 ? (nch == 1 ? 2 : 1) : WVS_STORE_NT;
				if (STORE == 1) {
					__builtin_nontemporal_store(t, ocol + (size_t)(255u - value) * (W >> 2));
				} else if (STORE == 0) {
					ocol[(size_t)(255u - value) * (W >> 2)] = t; /* plain (cached) stores: measurement only, round 6 */
				} else { /* 2: write-through (sc1: the table's bytes leave the L2 at once and take no line from the pixel stream --
					  * the one-channel modes, which do the least arithmetic between their loads, gain 4 %); 3: sc1 + nt */
					const scope_buf ob = scope_buf_make(table, SCOPE_WVS_ROWS * W * 4u);
					scope_llvm_raw_buffer_store_v4i32(scope_i4{(int)o[0], (int)o[1], (int)o[2], (int)o[3]}, ob,
									  (int)(((xs + (g << 2)) + (255u - value) * W) * 4u), 0, STORE == 2 ? 16 : 18);
				}
				if (HIS)
					tally(value, xg < groups, n); /* lanes past the right edge repeat a column group */
			}
		} else {
			const u32 xo = xs + (xg << 2);
			for (u32 value = threadIdx.x >> 4; value < 256u; value += blockDim.x >> 4) {
				u32 o[4], n[3];
				gather(value, xg, o, n);
				u32 *orow = (u32 *)(table + ((size_t)(255u - value) * W + xo) * 4);
				for (u32 j = 0; j < 4; j++)
					if (xo + j < W)
						orow[j] = o[j];
				if (HIS)
					tally(value, true, n); /* columns past the right edge hold no counts */
			}
		}
		__syncthreads(); /* the table is zeroed again next */
		if (HIS) {
			/* this unit's totals -> the frame's accumulator: slot-major, so that a wave instruction covers 64
			 * consecutive words (two cache lines); channels the caller did not ask for are dropped here */
			u32 *acc = his.acc + (size_t)ub.frame * SCOPE_HIS_WORDS;
			for (u32 t = threadIdx.x; t < 768u; t += blockDim.x) {
				const u32 byte = t >> 8; /* hsum is [texel byte][value]; histogram slot = 2 - byte (R,G,B) */
				const u32 v = hsum[t];
				if (v && ((CH & his.chmask) >> byte) & 1u)
					__hip_atomic_fetch_add(&acc[(2u - byte) * 256u + (t & 255u)], v, __ATOMIC_RELAXED,
							       __HIP_MEMORY_SCOPE_AGENT);
			}
			if (scope_last_arrival(&his.tickets[ub.frame], strips * bands, &hsum[0])) {
				for (u32 t = threadIdx.x; t < SCOPE_HIS_WORDS; t += blockDim.x) {
					const u32 value = t >> 2, slot = t & 3u; /* slot 0 = R = byte 2, 1 = G, 2 = B, 3 unused */
					his.out[(size_t)ub.frame * SCOPE_HIS_WORDS + t] =
						__hip_atomic_exchange(&acc[slot * 256u + value], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				}
				if (threadIdx.x == 0)
					__hip_atomic_store(&his.tickets[ub.frame], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
			__syncthreads(); /* hsum[0] carried the verdict */
		}
	};
	u32 unit_b = blockIdx.x; /* < total: the grid never exceeds the unit count */
	run_unit(unit_b, std::false_type());
	for (unit_b += gridDim.x; unit_b < total; unit_b += gridDim.x)
		run_unit(unit_b, std::true_type());
#undef WVS_ISSUE
#undef WVS_BIN
#undef WVS_LANE_OFFSETS
}

/* ------------------------------------------------------------------------------------------
 * The dock's three scopes of ONE frame from ONE read (src/roi.c:329-341 hands one staged surface to every registered
 * scope; the reference walks it three times).
 *
 * All three tables do not fit one LDS (vectorscope 128 KiB + waveform 96 KiB > 160 KiB), and re-reading the frame
 * costs a second pass over HBM.  But the pixels of a narrow strip fit the REGISTER FILE: a 4K frame in 16-column strips
 * is 240 units of 16 x 2160 texels = 138 KB, i.e. 17 eight-byte loads per lane of a 1024-thread workgroup = 34 VGPRs.
 * So a workgroup loads its strip once (eight loads ahead of their use) and walks the registers twice:
 *   1. waveform counters (four replicas, as scope_wvs_kernel<..., 16>) -> flush, the histogram's totals riding on it;
 *   2. the same LDS, zeroed again, as the vectorscope's 16-bit table: K0 in fixed point on the texel pairs, absorb
 *      mode by wave vote -> a sparse partial table, which scope_sat_reduce_sparse_kernel adds up over the strips.
 * Lane -> texels: pair = lane & 7 (columns 2 pair, 2 pair + 1 of the strip), rslot = lane >> 3 picks one of EIGHT
 * row ranges that lie H/8 rows apart -- a wave instruction touches eight distant image regions, which keeps
 * neighbouring pixels' equal chroma from meeting in one LDS instruction (scope_stream's lane groups do the same) -- and
 * replica rslot & 3 of the waveform table.  One strip holds at most 16 * 4080 = 65 280 texels: the vectorscope's 16-bit
 * counters need no clamp pass.  Vector shape only (W % 4 == 0, 8-byte aligned rows, 16-byte aligned tables).
 * ---------------------------------------------------------------------------------------- */
#define DOCK_G 17 /* loads per lane: 16 waves x 17 x 8 row ranges cover 2176 rows */

template <int CH, bool HIS>
__global__ __launch_bounds__(SCOPE_BLOCK) void scope_dock_kernel(ScopeFrameSet fs, u32 strips, ScopeK0 k0, uint8_t *wvs_dst,
								ScopeWvsHis his, uint8_t *vss_part, uint8_t *vss_mask)
{
	constexpr u32 slot0 = 0, slot1 = (CH & 1), slot2 = slot1 + ((CH >> 1) & 1), nch = slot2 + ((CH >> 2) & 1);
	u32 *tab = scope_lds;
	scope_lds_is_at_zero(tab);
	const u32 W = fs.width, H = fs.height;
	const u32 lane = threadIdx.x & 63u;
	const u32 pair = lane & 7u, rslot = lane >> 3;
	const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const u32 nwaves = blockDim.x >> 6;
	const u32 wbyte = (lane & 31u) << 2; /* replica (rslot & 3) x column pair, as scope_wvs_kernel<..., 16> */

	u32 unit = blockIdx.x;
	const u32 total = fs.n_frames * strips;
	if (unit < (total & ~15u)) /* workgroups b and b + 8 share an XCD: neighbouring strips, the two halves of a cache line */
		unit = (unit & ~15u) + ((unit & 7u) << 1) + ((unit >> 3) & 1u);
	const u32 frame = unit / strips, strip = unit - frame * strips;
	const u32 xs = strip * 16u, x0 = xs + 2u * pair;
	const bool live = x0 < W; /* W % 4 == 0: both texels of the pair or none */
	const uint8_t *plane = fs.plane[frame];

	/* rows of this lane: rslot * rps + i, i = wave, wave + nwaves, ... < rps */
	const u32 rps = (H + 7u) >> 3;
	const u32 row0 = rslot * rps;
	u32 px[DOCK_G][2];
#ifndef DOCK_AHEAD
#define DOCK_AHEAD 8 /* loads issued before the first one is used; the rest follow one per binned load.  All 17 at once:
		      * 29.7 us per 4K frame; 2 ahead 29.9, 4: 29.0, 8: 28.7 (natural content 29.2 / 29.3 / 28.4 / 28.1) */
#endif
	auto issue = [&](int g) {
		const u32 i = wave + (u32)g * nwaves;
		const u32 row = row0 + i;
		const bool ok = live && i < rps && row < H;
		const scope_u2 t = scope_load_row2_nt(plane, ok ? row * fs.linesize + x0 * 4u : 0u);
		px[g][0] = t.x;
		px[g][1] = t.y;
	};
#pragma unroll
	for (int g = 0; g < DOCK_G && g < DOCK_AHEAD; g++)
		issue(g);
	auto valid = [&](int g) {
		const u32 i = wave + (u32)g * nwaves;
		return live && i < rps && row0 + i < H;
	};

	/* ---- 1. waveform (+ histogram) ---------------------------------------------------------------------------------- */
	const uint4 zero = make_uint4(0, 0, 0, 0);
	/* the histogram's per-strip totals live BEHIND the vectorscope table, so that they survive into the second phase: their
	 * way to the frame's accumulator (device atomics, then the arrival ticket) is started after the waveform's flush and
	 * only waited for at the very end of the kernel, with the vectorscope phase in between */
	u32 *hsum = tab + VSS_WORDS;
	for (u32 i = threadIdx.x * 4; i < nch * WVS_CH_WORDS; i += blockDim.x * 4)
		*(uint4 *)&tab[i] = zero;
	if (HIS && threadIdx.x < 192u)
		*(uint4 *)&hsum[threadIdx.x * 4] = zero;
	__syncthreads();
#pragma unroll
	for (int g = 0; g < DOCK_G; g++) {
		if (g + DOCK_AHEAD < DOCK_G)
			issue(g + DOCK_AHEAD);
		if (!valid(g))
			continue;
		const u32 p0 = px[g][0], p1 = px[g][1];
		if ((p0 >> 24) != 0) /* src/waveform.c:247-248 */
			wvs_count<CH>(tab, p0, wbyte, 1u);
		if ((p1 >> 24) != 0)
			wvs_count<CH>(tab, p1, wbyte, 65536u);
	}
	__syncthreads();
	{
		/* the narrow-strip flush of scope_wvs_kernel: one (value, column group) per thread, replicas read in rotated order */
		uint8_t *table = wvs_dst + (size_t)frame * ((size_t)SCOPE_WVS_ROWS * W * 4);
		const u32 groups = (min(W - xs, 16u)) >> 2;
		const u32 g4 = threadIdx.x & 3u, rho = (threadIdx.x >> 2) & 3u;
		const u32 value = ((threadIdx.x >> 4) << 2) + rho;
		const u32 g = min(g4, groups - 1u);
		uint2 acc[3] = {make_uint2(0, 0), make_uint2(0, 0), make_uint2(0, 0)};
#pragma unroll
		for (u32 j = 0; j < 4; j++) {
			const u32 w = (value << 5) + (((rho + j) & 3u) << 3) + 2u * g;
			if (CH & 1) {
				const uint2 c = *(const uint2 *)&tab[slot0 * WVS_CH_WORDS + w];
				acc[0].x += c.x;
				acc[0].y += c.y;
			}
			if (CH & 2) {
				const uint2 c = *(const uint2 *)&tab[slot1 * WVS_CH_WORDS + w];
				acc[1].x += c.x;
				acc[1].y += c.y;
			}
			if (CH & 4) {
				const uint2 c = *(const uint2 *)&tab[slot2 * WVS_CH_WORDS + w];
				acc[2].x += c.x;
				acc[2].y += c.y;
			}
		}
		u32 o[4] = {0, 0, 0, 0};
#pragma unroll
		for (u32 c = 0; c < 3; c++) {
			if (!((CH >> c) & 1))
				continue;
			o[0] |= min(acc[c].x & 0xffffu, 255u) << (8u * c);
			o[1] |= min(acc[c].x >> 16, 255u) << (8u * c);
			o[2] |= min(acc[c].y & 0xffffu, 255u) << (8u * c);
			o[3] |= min(acc[c].y >> 16, 255u) << (8u * c);
			if (HIS && g4 < groups)
				atomicAdd(&hsum[c * 256u + value],
					  (acc[c].x & 0xffffu) + (acc[c].x >> 16) + (acc[c].y & 0xffffu) + (acc[c].y >> 16));
		}
		const scope_v4 t = {o[0], o[1], o[2], o[3]};
		__builtin_nontemporal_store(t, (scope_v4 *)(table + (size_t)(xs + (g << 2)) * 4) + (size_t)(255u - value) * (W >> 2));
		__syncthreads();
		if (HIS) { /* fire and forget; scope_last_arrival at the end of the kernel waits for them */
			u32 *acc_g = his.acc + (size_t)frame * SCOPE_HIS_WORDS;
			for (u32 t2 = threadIdx.x; t2 < 768u; t2 += blockDim.x) {
				const u32 byte = t2 >> 8;
				const u32 v = hsum[t2];
				if (v && ((CH & his.chmask) >> byte) & 1u)
					__hip_atomic_fetch_add(&acc_g[(2u - byte) * 256u + (t2 & 255u)], v, __ATOMIC_RELAXED,
							       __HIP_MEMORY_SCOPE_AGENT);
			}
		}
	}

	/* ---- 2. vectorscope: the same registers through K0 ------------------------------------------------------------- */
	for (u32 i = threadIdx.x * 4; i < VSS_WORDS; i += blockDim.x * 4)
		*(uint4 *)&tab[i] = zero;
	__syncthreads();
	VssLane st = {0xffffffffu, 0};
	bool absorb = false; /* wave-uniform */
#pragma unroll
	for (int g = 0; g < DOCK_G; g++) {
		u32 key[2] = {px[g][0], px[g][1]};
		scope_k0_cell<2, true>(key, k0, true, false); /* wave votes inside: every lane goes through, valid or not */
		const bool ok = valid(g);
		if ((g & 3) == 0) /* content is coherent: the verdict holds for four loads */
			absorb = __popcll(__ballot(!ok || key[0] == key[1])) >= 48;
		if (!ok)
			continue;
		if (__builtin_expect(absorb, false)) {
			bool e0 = key[0] == st.colour, e1 = key[1] == st.colour;
			if (!(e0 | e1)) {
				if (st.count)
					vss_add<0>(tab, st.colour, st.count);
				st.colour = key[0];
				st.count = 0;
				e0 = true;
				e1 = key[1] == key[0];
			}
			st.count += (u32)e0 + (u32)e1;
			if (!e0)
				vss_add<0>(tab, key[0], 1);
			if (!e1)
				vss_add<0>(tab, key[1], 1);
		} else {
			vss_add<0>(tab, key[0], 1);
			vss_add<0>(tab, key[1], 1);
		}
	}
	vss_flush_lane<0>(tab, st);
	__syncthreads();
	{
		/* the sparse partial of scope_vss_kernel, slice = strip */
		uint4 *outs = (uint4 *)vss_part + ((size_t)frame * 256u * strips + strip) * 16u;
		u32 wrote = 0, k = 0;
		for (u32 g = threadIdx.x; g < SCOPE_VSS_BINS / 16; g += blockDim.x, k++) {
			const u32 row = 255u - (g >> 4);
			const u32 w0 = (row << 7) + (((g & 15u) << 3) ^ ((row & 15u) << 3));
			const uint4 lo = *(const uint4 *)&tab[w0], hi = *(const uint4 *)&tab[w0 + 4];
			if ((lo.x | lo.y | lo.z | lo.w | hi.x | hi.y | hi.z | hi.w) != 0) {
				const scope_v4 t = {vss_clamp_pair(lo.x, lo.y), vss_clamp_pair(lo.z, lo.w), vss_clamp_pair(hi.x, hi.y),
						    vss_clamp_pair(hi.z, hi.w)};
				__builtin_nontemporal_store(t, (scope_v4 *)outs + ((size_t)(g >> 4) * strips * 16u + (g & 15u)));
				wrote |= 1u << k;
			}
		}
		vss_mask[(((size_t)frame * (blockDim.x >> 4) + (threadIdx.x >> 4)) * strips + strip) * 16u + (threadIdx.x & 15u)] =
			(uint8_t)wrote;
	}
	if (HIS) {
		u32 *acc_g = his.acc + (size_t)frame * SCOPE_HIS_WORDS;
		if (scope_last_arrival(&his.tickets[frame], strips, &hsum[0])) {
			for (u32 t2 = threadIdx.x; t2 < SCOPE_HIS_WORDS; t2 += blockDim.x) {
				const u32 val = t2 >> 2, slot = t2 & 3u;
				his.out[(size_t)frame * SCOPE_HIS_WORDS + t2] =
					__hip_atomic_exchange(&acc_g[slot * 256u + val], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
			if (threadIdx.x == 0)
				__hip_atomic_store(&his.tickets[frame], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
	}
}

/* Capture-side pre-steps: point-sampled 1/scale down-scale and ROI crop in one pass (cm_render_target,
 * src/common.c:249-250,272-282).  `src` points at the first sampled texel, `row_pitch` is the distance in
 * bytes between sampled rows and `col_step` the distance in texels between sampled columns.  A lane
 * produces four adjacent output texels of one row (one 16-byte store when the destination allows it);
 * only the source rows and cache lines that hold a sampled texel are read. */
__global__ __launch_bounds__(256) void scope_stage_kernel(const uint8_t *src, u64 row_pitch, u32 col_step, u32 cx, u32 cy,
							 uint8_t *dst, u32 dst_linesize, u32 vec)
{
	const u32 quads = (cx + 3u) >> 2;
	const u64 cells = (u64)quads * cy;
	for (u64 c = (u64)blockIdx.x * blockDim.x + threadIdx.x; c < cells; c += (u64)gridDim.x * blockDim.x) {
		const u32 j = (u32)(c / quads);
		const u32 i0 = (u32)(c - (u64)j * quads) << 2;
		const uint8_t *row = src + (size_t)j * row_pitch;
		uint8_t *orow = dst + (size_t)j * dst_linesize + (size_t)i0 * 4;
		u32 t[4] = {0, 0, 0, 0};
#pragma unroll
		for (u32 k = 0; k < 4; k++)
			if (i0 + k < cx)
				t[k] = *(const u32 *)(row + (size_t)(i0 + k) * col_step * 4);
		if (vec && i0 + 3 < cx) {
			*(uint4 *)orow = make_uint4(t[0], t[1], t[2], t[3]);
		} else {
			for (u32 k = 0; k < 4; k++)
				if (i0 + k < cx)
					((u32 *)orow)[k] = t[k];
		}
	}
}

/* ------------------------------------------------------------------------------------------
 * K0 standalone, partial-table reductions, multi-GPU helpers
 * ---------------------------------------------------------------------------------------- */

__global__ __launch_bounds__(256) void scope_k0_kernel(const uint8_t *src, u32 src_linesize, uint8_t *dst,
						      u32 dst_linesize, u32 ncols_a, u32 ncols_b, u32 xbyte_b, u32 rows,
						      ScopeK0 k0)
{
	const u32 stride = gridDim.x * blockDim.x;
	const u32 gid = blockIdx.x * blockDim.x + threadIdx.x;
	if (ncols_a) {
		const u32 total = rows * ncols_a;
		for (u32 c = gid; c < total; c += stride) {
			const u32 row = c / ncols_a, col = c - row * ncols_a;
			const uint4 q = *(const uint4 *)(src + (size_t)row * src_linesize + (size_t)col * 16);
			u32 t[4] = {q.x, q.y, q.z, q.w};
			scope_k0_cell<4>(t, k0, true, true);
			*(uint4 *)(dst + (size_t)row * dst_linesize + (size_t)col * 16) = make_uint4(t[0], t[1], t[2], t[3]);
		}
	}
	if (ncols_b) {
		const u32 total = rows * ncols_b;
		for (u32 c = gid; c < total; c += stride) {
			const u32 row = c / ncols_b, col = c - row * ncols_b;
			u32 t[1] = {*(const u32 *)(src + (size_t)row * src_linesize + xbyte_b + (size_t)col * 4)};
			scope_k0_cell<1>(t, k0, true, true);
			*(u32 *)(dst + (size_t)row * dst_linesize + xbyte_b + (size_t)col * 4) = t[0];
		}
	}
}

/* out[f][i] = min(sum_s part[f][s][i], 255); 16 bytes per lane; slices <= 256 so the
 * packed 16-bit lanes (max 256*255 = 65280) cannot overflow. */
__global__ __launch_bounds__(256) void scope_sat_reduce_kernel(const uint8_t *part, uint8_t *out, u32 slices, u64 len16,
							      u32 packed16)
{
	const u32 frame = blockIdx.y;
	const u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= len16)
		return;
	const uint4 *p = (const uint4 *)part + (u64)frame * slices * len16 + g;
	const uint4 r = scope_sat_sum16(p, len16, slices);
	if (!packed16) {
		((uint4 *)out)[(u64)frame * len16 + g] = r;
		return;
	}
	/* the wire format of the multi-GPU sum: two bins per 32-bit word, lo | hi << 16 (scope_table_pack_u8_u16x2) */
	const u32 w[4] = {r.x, r.y, r.z, r.w};
	u32 o[8];
	for (int j = 0; j < 4; j++) {
		o[2 * j] = (w[j] & 0xffu) | ((w[j] & 0xff00u) << 8);
		o[2 * j + 1] = ((w[j] >> 16) & 0xffu) | ((w[j] >> 24) << 16);
	}
	uint4 *q = (uint4 *)out + ((u64)frame * len16 + g) * 2;
	q[0] = make_uint4(o[0], o[1], o[2], o[3]);
	q[1] = make_uint4(o[4], o[5], o[6], o[7]);
}

/* The same sum over SPARSE partials (scope_vss_kernel with pmask): only the 16-bin groups a slice has flagged are read.
 * Parallel in both directions -- round 1's kernel gave a one-frame launch 16 workgroups that walked 256 partials one
 * after the other (74 us): a workgroup of 256 threads takes gpw = 256 >> sl_log2 groups, and the 2^sl_log2 threads
 * that share a group take every 2^sl_log2-th slice, four loads in flight each, and meet in LDS.  One frame of 256
 * slices = 256 workgroups x 16 groups x 16 slice lanes.
 * mask_block = blockDim.x of the producing launch: group g is bit g / mask_block of byte g % mask_block of a mask.
 * grid = (4096 / gpw, n_frames); dynamic LDS = 8 KiB + 32 * gpw bytes. */
/* K = slices a slice lane takes (1 ... 16; >= slices / 2^sl_log2): the loads are unconditional, so a launch of 4
 * slices per frame (the 64-frame batches) issued 16 mask and 16 group loads per thread for ONE useful pair until round 3 --
 * 18.5 us per 64-frame step, 5.6 % of it; with K = 1: see profiles/round3. */
template <int K>
__global__ __launch_bounds__(256) void scope_sat_reduce_sparse_kernel(const uint8_t *part, const uint8_t *pmask, uint8_t *out,
									     u32 slices, u32 mask_block, u32 sl_log2, u32 packed16)
{
	const u32 frame = blockIdx.y;
	const u32 gpw = 256u >> sl_log2, nsl = 1u << sl_log2;
	const u32 gi = threadIdx.x & (gpw - 1u), sl = threadIdx.x >> (8u - sl_log2);
	const u32 g = blockIdx.x * gpw + gi; /* < 4096 */
	const u32 mbyte = g % mask_block, mbit = g / mask_block;
	/* blocked layout (scope_vss_kernel): consecutive slices of one group are 16 cells (256 B) / 16 mask bytes apart */
	const uint4 *p = (const uint4 *)part + ((u64)frame * 256u + (g >> 4)) * slices * 16u + (g & 15u);
	const uint8_t *m = pmask + ((u64)frame * (mask_block >> 4) + (mbyte >> 4)) * slices * 16u + (mbyte & 15u);
	u32 acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; /* even bytes of the four words, then the odd ones: 16-bit lanes, <= 256 * 255 */
	/* Two round trips to memory, not two per slice: a lane has at most 16 slices (256 / 16 slice lanes); all their mask
	 * bytes are requested at once, then all the flagged groups (partials written by other XCDs come from the
	 * Infinity Cache at ~1 us a trip -- with four loads in flight this kernel took 8.4 us for one frame). */
	/* Every load is unconditional (a lane without a slice, or with an unflagged group, re-reads cell 0 of its own
	 * column, which is always mapped, and drops the value): predicated loads made hipcc wait for each one inside
	 * its branch -- 32 serial round trips, 10 us for one frame. */
	u32 mb[K];
	bool on[K];
	uint4 q[K];
#pragma unroll
	for (u32 k = 0; k < (u32)K; k++) {
		const u32 s = sl + k * nsl;
		mb[k] = m[(s < slices ? s : 0u) * 16u];
	}
#pragma unroll
	for (u32 k = 0; k < (u32)K; k++) {
		const u32 s = sl + k * nsl;
		on[k] = (s < slices) & (((mb[k] >> mbit) & 1u) != 0);
		q[k] = p[(on[k] ? s : 0u) * 16u];
	}
#pragma unroll
	for (u32 k = 0; k < (u32)K; k++)
		if (!on[k])
			q[k] = make_uint4(0, 0, 0, 0);
#pragma unroll
	for (u32 k = 0; k < (u32)K; k++) {
		acc[0] += q[k].x & 0x00ff00ffu;
		acc[4] += (q[k].x >> 8) & 0x00ff00ffu;
		acc[1] += q[k].y & 0x00ff00ffu;
		acc[5] += (q[k].y >> 8) & 0x00ff00ffu;
		acc[2] += q[k].z & 0x00ff00ffu;
		acc[6] += (q[k].z >> 8) & 0x00ff00ffu;
		acc[3] += q[k].w & 0x00ff00ffu;
		acc[7] += (q[k].w >> 8) & 0x00ff00ffu;
	}
	if (sl_log2) { /* uniform: the slice lanes of a group meet in LDS */
		u32 *red = scope_lds; /* [nsl][gpw][8] = 2048 words, then the totals [gpw][8] */
#pragma unroll
		for (u32 j = 0; j < 8; j++)
			red[(sl * gpw + gi) * 8u + j] = acc[j];
		__syncthreads();
		u32 *tot = red + 2048;
		for (u32 i = threadIdx.x; i < gpw * 8u; i += blockDim.x) { /* i = gi' * 8 + j */
			u32 sum = 0;
			for (u32 l = 0; l < nsl; l++)
				sum += red[l * gpw * 8u + i];
			tot[i] = sum;
		}
		__syncthreads();
		if (threadIdx.x >= gpw)
			return;
#pragma unroll
		for (u32 j = 0; j < 8; j++)
			acc[j] = tot[threadIdx.x * 8u + j];
	}
	/* acc[j] / acc[4 + j]: bins 0,2 / 1,3 of word j.  Saturate and store. */
	u32 b[16];
#pragma unroll
	for (u32 j = 0; j < 4; j++) {
		b[4 * j + 0] = min(acc[j] & 0xffffu, 255u);
		b[4 * j + 1] = min(acc[4 + j] & 0xffffu, 255u);
		b[4 * j + 2] = min(acc[j] >> 16, 255u);
		b[4 * j + 3] = min(acc[4 + j] >> 16, 255u);
	}
	if (!packed16) {
		((uint4 *)out)[(u64)frame * (SCOPE_VSS_BINS / 16) + g] =
			make_uint4(b[0] | b[1] << 8 | b[2] << 16 | b[3] << 24, b[4] | b[5] << 8 | b[6] << 16 | b[7] << 24,
				   b[8] | b[9] << 8 | b[10] << 16 | b[11] << 24, b[12] | b[13] << 8 | b[14] << 16 | b[15] << 24);
		return;
	}
	/* the wire format of the multi-GPU sum: two bins per 32-bit word, lo | hi << 16 (scope_table_pack_u8_u16x2) */
	uint4 *wire = (uint4 *)out + ((u64)frame * (SCOPE_VSS_BINS / 16) + g) * 2;
	wire[0] = make_uint4(b[0] | b[1] << 16, b[2] | b[3] << 16, b[4] | b[5] << 16, b[6] | b[7] << 16);
	wire[1] = make_uint4(b[8] | b[9] << 16, b[10] | b[11] << 16, b[12] | b[13] << 16, b[14] | b[15] << 16);
}

__global__ __launch_bounds__(256) void scope_sum_reduce_kernel(const u32 *part, u32 *out, u32 slices, u32 len)
{
	const u32 frame = blockIdx.y;
	const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= len)
		return;
	const u32 *p = part + (size_t)frame * slices * len + i;
	u32 sum = 0;
	for (u32 s = 0; s < slices; s++)
		sum += p[(size_t)s * len];
	out[(size_t)frame * len + i] = sum;
}

__global__ __launch_bounds__(256) void scope_widen_kernel(const uint8_t *in, u32 *out, u64 n)
{
	const u64 stride = (u64)gridDim.x * blockDim.x;
	for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
		out[i] = in[i];
}

__global__ __launch_bounds__(256) void scope_clamp_kernel(const u32 *in, uint8_t *out, u64 n)
{
	const u64 stride = (u64)gridDim.x * blockDim.x;
	for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
		out[i] = (uint8_t)min(in[i], 255u);
}

__global__ __launch_bounds__(256) void scope_pack16_kernel(const uint8_t *in, u32 *out, u64 n_pairs)
{
	const u64 stride = (u64)gridDim.x * blockDim.x;
	for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n_pairs; i += stride)
		out[i] = (u32)in[2 * i] | ((u32)in[2 * i + 1] << 16);
}

__global__ __launch_bounds__(256) void scope_unpack16_clamp_kernel(const u32 *in, uint8_t *out, u64 n_pairs)
{
	const u64 stride = (u64)gridDim.x * blockDim.x;
	for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n_pairs; i += stride) {
		const u32 w = in[i];
		out[2 * i] = (uint8_t)min(w & 0xffffu, 255u);
		out[2 * i + 1] = (uint8_t)min(w >> 16, 255u);
	}
}

/* ------------------------------------------------------------------------------------------
 * Launchers
 * ---------------------------------------------------------------------------------------- */


namespace {

struct Regions {
	u32 ncols_a, ncols_b, xbyte_b;
};

/* 16-byte cells where every row start is 16-byte aligned, single pixels elsewhere. */
Regions split_regions(const ScopeFrameSet &fs)
{
	bool aligned = (fs.linesize & 15u) == 0;
	for (u32 i = 0; i < fs.n_frames; i++)
		aligned = aligned && (((uintptr_t)fs.plane[i]) & 15u) == 0;
	Regions r;
	if (aligned) {
		r.ncols_a = fs.width / 4;
		r.ncols_b = fs.width & 3u;
		r.xbyte_b = r.ncols_a * 16;
	} else {
		r.ncols_a = 0;
		r.ncols_b = fs.width;
		r.xbyte_b = 0;
	}
	return r;
}

const u32 HIS_LDS_BYTES = HIS_WORDS * 4u;
const u32 VSS_LDS_BYTES = VSS_WORDS * 4u;
const u32 WVS_HIS_LDS_BYTES = 768u * 4u; /* the histogram totals of one unit */
const u32 WVS_LDS_BYTES = 3u * WVS_CH_WORDS * 4u + WVS_HIS_LDS_BYTES;

/* Every kernel variant is launched through one of these; `allow` only raises the dynamic-LDS limit
 * (scope_kernels_init walks all variants once with allow = true). */
struct VssArgs {
	const ScopeFrameSet *fs;
	u32 slices, glog2, block, rle_force;
	Regions rg;
	ScopeK0 k0;
	uint8_t *dst, *pmask;
	VssFold fold;
	hipStream_t st;
	const ScopeTiming *tm;
};

template <int U, bool NT, int MERGE, bool FUSE, int HASH> hipError_t vss_variant(const VssArgs *a)
{
	/* behind the table: the run-length mode's scratch rows, or (at least) the word the in-launch fold's arrival flag lives in */
	constexpr u32 lds = VssTable<HASH>::words * 4u + (FUSE && HASH == 2 && MERGE == 3 && VSS_RLE_IN(U) ? VSS_RLE_SCRATCH_BYTES : 16u);
	if (!a)
		return hipFuncSetAttribute((const void *)scope_vss_kernel<U, NT, MERGE, FUSE, HASH>,
					   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
	SCOPE_LAUNCH((scope_vss_kernel<U, NT, MERGE, FUSE, HASH>), dim3(a->fs->n_frames * a->slices), dim3(a->block),
		     lds, a->st, a->tm, *a->fs, a->slices, a->rg.ncols_a, a->rg.ncols_b, a->rg.xbyte_b, a->glog2, a->rle_force, a->k0,
		     a->dst, a->pmask, a->fold);
	return hipGetLastError();
}

template <int U, bool NT, int MERGE, bool FUSE> hipError_t vss_pick_hash(int hash, const VssArgs *a)
{
	if (hash == 2 && FUSE) /* the pair layout exists for the fused-K0 form only */
		return vss_variant<U, NT, MERGE, FUSE, FUSE ? 2 : 0>(a);
	return hash == 1 ? vss_variant<U, NT, MERGE, FUSE, 1>(a) : vss_variant<U, NT, MERGE, FUSE, 0>(a);
}
template <int U, bool NT, int MERGE> hipError_t vss_pick_fuse(int fuse, int hash, const VssArgs *a)
{
	return fuse ? vss_pick_hash<U, NT, MERGE, true>(hash, a) : vss_pick_hash<U, NT, MERGE, false>(hash, a);
}
template <int U, bool NT> hipError_t vss_pick_merge(int merge, int fuse, int hash, const VssArgs *a)
{
	if (merge <= 0)
		return vss_pick_fuse<U, NT, 0>(fuse, hash, a);
	if (merge <= 2) /* 1 and 2: always absorb mode */
		return vss_pick_fuse<U, NT, 2>(fuse, hash, a);
	return vss_pick_fuse<U, NT, 3>(fuse, hash, a);
}
template <int U> hipError_t vss_pick_nt(int nt, int merge, int fuse, int hash, const VssArgs *a)
{
	return nt ? vss_pick_merge<U, true>(merge, fuse, hash, a) : vss_pick_merge<U, false>(merge, fuse, hash, a);
}
hipError_t vss_pick(int unroll, int nt, int merge, int fuse, int hash, const VssArgs *a)
{
	if (unroll <= 1)
		return vss_pick_nt<1>(nt, merge, fuse, hash, a);
	if (unroll == 2)
		return vss_pick_nt<2>(nt, merge, fuse, hash, a);
	return vss_pick_nt<4>(nt, merge, fuse, hash, a);
}

struct HisArgs {
	const ScopeFrameSet *fs;
	u32 slices, chmask, block;
	Regions rg;
	ScopeK0 k0;
	u32 *dst, *acc, *tickets;
	hipStream_t st;
	const ScopeTiming *tm;
};

template <int U, bool NT, bool FUSE> hipError_t his_variant(const HisArgs *a)
{
	if (!a)
		return hipFuncSetAttribute((const void *)scope_his_kernel<U, NT, FUSE>, hipFuncAttributeMaxDynamicSharedMemorySize,
					   (int)HIS_LDS_BYTES);
	SCOPE_LAUNCH((scope_his_kernel<U, NT, FUSE>), dim3(a->fs->n_frames * a->slices), dim3(a->block), HIS_LDS_BYTES, a->st,
		     a->tm, *a->fs, a->slices, a->rg.ncols_a, a->rg.ncols_b, a->rg.xbyte_b, a->chmask, a->k0, a->dst, a->acc,
		     a->tickets);
	return hipGetLastError();
}
template <int U, bool NT> hipError_t his_pick_fuse(int fuse, const HisArgs *a)
{
	return fuse ? his_variant<U, NT, true>(a) : his_variant<U, NT, false>(a);
}
template <int U> hipError_t his_pick_nt(int nt, int fuse, const HisArgs *a)
{
	return nt ? his_pick_fuse<U, true>(fuse, a) : his_pick_fuse<U, false>(fuse, a);
}
hipError_t his_pick(int unroll, int nt, int fuse, const HisArgs *a)
{
	if (unroll <= 1)
		return his_pick_nt<1>(nt, fuse, a);
	if (unroll == 2)
		return his_pick_nt<2>(nt, fuse, a);
	return his_pick_nt<4>(nt, fuse, a);
}

struct WvsArgs {
	const ScopeFrameSet *fs;
	u32 strips, bands, lds, grid;
	ScopeK0 k0;
	uint8_t *dst;
	ScopeWvsHis his;
	hipStream_t st;
	const ScopeTiming *tm;
};

template <int CH, bool FUSE, bool WIDE, bool HIS, int SW, bool DEEP> hipError_t wvs_variant(const WvsArgs *a)
{
	if (!a)
		return hipFuncSetAttribute((const void *)scope_wvs_kernel<CH, FUSE, WIDE, HIS, SW, DEEP>,
					   hipFuncAttributeMaxDynamicSharedMemorySize, (int)WVS_LDS_BYTES);
	SCOPE_LAUNCH((scope_wvs_kernel<CH, FUSE, WIDE, HIS, SW, DEEP>), dim3(a->grid), dim3(SCOPE_BLOCK), a->lds, a->st, a->tm,
		     *a->fs, a->strips, a->bands, a->k0, a->dst, a->his);
	return hipGetLastError();
}
/* wide: 0 scalar shape, 1 vector shape, 2 vector shape with narrow strips, 3 vector shape with the deep queue across flushes */
template <int CH, bool HIS> hipError_t wvs_pick_shape(int fuse, int wide, const WvsArgs *a)
{
	if (wide == 3 && !fuse) /* the fused-K0 forms are short of VALU issue slots, not of loads: they lose 3 % with it */
		return wvs_variant<CH, false, true, HIS, 64, true>(a);
	if (wide == 3 && CH == 2) /* ... except the one-row form (luma, BASELINE config 3's Y mode): +2.3 ... 2.7 %, visit d1 */
		return wvs_variant<CH == 2 ? CH : 2, true, true, HIS, 64, true>(a);
	if (wide == 3)
		wide = 1;
	if (wide == 2)
		return fuse ? wvs_variant<CH, true, true, HIS, WVS_NARROW, false>(a)
			    : wvs_variant<CH, false, true, HIS, WVS_NARROW, false>(a);
	if (fuse)
		return wide ? wvs_variant<CH, true, true, HIS, 64, false>(a) : wvs_variant<CH, true, false, HIS, 64, false>(a);
	return wide ? wvs_variant<CH, false, true, HIS, 64, false>(a) : wvs_variant<CH, false, false, HIS, 64, false>(a);
}
template <bool HIS> hipError_t wvs_pick_ch(u32 chmask, int fuse, int wide, const WvsArgs *a)
{
	switch (chmask & 7u) {
	case 1:
		return wvs_pick_shape<1, HIS>(fuse, wide, a);
	case 2:
		return wvs_pick_shape<2, HIS>(fuse, wide, a);
	case 3:
		return wvs_pick_shape<3, HIS>(fuse, wide, a);
	case 4:
		return wvs_pick_shape<4, HIS>(fuse, wide, a);
	case 5:
		return wvs_pick_shape<5, HIS>(fuse, wide, a);
	case 6:
		return wvs_pick_shape<6, HIS>(fuse, wide, a);
	default:
		return wvs_pick_shape<7, HIS>(fuse, wide, a);
	}
}
hipError_t wvs_pick(u32 chmask, int fuse, int wide, int with_his, const WvsArgs *a)
{
	return with_his ? wvs_pick_ch<true>(chmask, fuse, wide, a) : wvs_pick_ch<false>(chmask, fuse, wide, a);
}

struct DockArgs {
	const ScopeFrameSet *fs;
	u32 strips;
	ScopeK0 k0;
	uint8_t *wvs_dst;
	ScopeWvsHis his;
	uint8_t *vss_part, *vss_mask;
	hipStream_t st;
	const ScopeTiming *tm;
};
template <int CH, bool HIS> hipError_t dock_variant(const DockArgs *a)
{
	if (!a)
		return hipFuncSetAttribute((const void *)scope_dock_kernel<CH, HIS>, hipFuncAttributeMaxDynamicSharedMemorySize,
					   (int)(VSS_LDS_BYTES + WVS_HIS_LDS_BYTES));
	SCOPE_LAUNCH((scope_dock_kernel<CH, HIS>), dim3(a->fs->n_frames * a->strips), dim3(SCOPE_BLOCK), VSS_LDS_BYTES + WVS_HIS_LDS_BYTES, a->st,
		     a->tm, *a->fs, a->strips, a->k0, a->wvs_dst, a->his, a->vss_part, a->vss_mask);
	return hipGetLastError();
}
template <bool HIS> hipError_t dock_pick_ch(u32 chmask, const DockArgs *a)
{
	switch (chmask & 7u) {
	case 1:
		return dock_variant<1, HIS>(a);
	case 2:
		return dock_variant<2, HIS>(a);
	case 3:
		return dock_variant<3, HIS>(a);
	case 4:
		return dock_variant<4, HIS>(a);
	case 5:
		return dock_variant<5, HIS>(a);
	case 6:
		return dock_variant<6, HIS>(a);
	default:
		return dock_variant<7, HIS>(a);
	}
}

} // namespace

hipError_t scope_kernels_init(void)
{
	hipError_t e = hipSuccess;
	for (int unroll = 1; unroll <= 4; unroll *= 2)
		for (int nt = 0; nt < 2; nt++)
			for (int fuse = 0; fuse < 2; fuse++) {
				for (int merge = 0; merge < 4; merge++)
					for (int hash = 0; hash < 3; hash++)
						if (e == hipSuccess)
							e = vss_pick(unroll, nt, merge, fuse, hash, NULL);
				if (e == hipSuccess)
					e = his_pick(unroll, nt, fuse, NULL);
			}
	for (u32 chmask = 1; chmask <= 7; chmask++)
		for (int shape = 0; shape < 16; shape++)
			if (e == hipSuccess)
				e = wvs_pick(chmask, shape & 1, (shape >> 1) & 3, shape >> 3, NULL);
	for (u32 chmask = 1; chmask <= 7; chmask++) {
		if (e == hipSuccess)
			e = dock_pick_ch<false>(chmask, NULL);
		if (e == hipSuccess)
			e = dock_pick_ch<true>(chmask, NULL);
	}
	if (e == hipSuccess)
		e = scope_trio_init();
	return e;
}

hipError_t scope_launch_stage(const uint8_t *src, uint64_t row_pitch, uint32_t col_step, uint32_t cx, uint32_t cy,
			      uint8_t *dst, uint32_t dst_linesize, hipStream_t st)
{
	const u32 vec = ((dst_linesize & 15u) == 0 && (((uintptr_t)dst) & 15u) == 0) ? 1u : 0u;
	const u64 cells = (u64)((cx + 3u) >> 2) * cy;
	u32 blocks = (u32)((cells + 255) / 256);
	if (blocks > 256 * 16)
		blocks = 256 * 16;
	if (blocks == 0)
		return hipSuccess;
	hipLaunchKernelGGL(scope_stage_kernel, dim3(blocks), dim3(256), 0, st, src, (u64)row_pitch, col_step, cx, cy, dst,
			   dst_linesize, vec);
	return hipGetLastError();
}

hipError_t scope_launch_k0(const uint8_t *src, uint32_t src_linesize, uint8_t *dst, uint32_t dst_linesize,
			   uint32_t width, uint32_t height, int colorspace, hipStream_t st)
{
	const bool aligned =
		((src_linesize | dst_linesize) & 15u) == 0 && ((((uintptr_t)src) | ((uintptr_t)dst)) & 15u) == 0;
	const u32 ncols_a = aligned ? width / 4 : 0;
	const u32 ncols_b = aligned ? (width & 3u) : width;
	const u32 xbyte_b = ncols_a * 16;
	const u64 cells = (u64)height * (ncols_a ? ncols_a : ncols_b);
	u32 blocks = (u32)((cells + 255) / 256);
	if (blocks > 256 * 8)
		blocks = 256 * 8;
	if (blocks == 0)
		blocks = 1;
	hipLaunchKernelGGL(scope_k0_kernel, dim3(blocks), dim3(256), 0, st, src, src_linesize, dst, dst_linesize, ncols_a,
			   ncols_b, xbyte_b, height, scope_k0_matrix(colorspace));
	return hipGetLastError();
}

/* threads per workgroup: a multiple of 64 up to SCOPE_BLOCK (the kernels size their loops by blockDim) */
static u32 scope_block(int requested)
{
	if (requested < 64)
		return SCOPE_BLOCK;
	const u32 b = (u32)requested & ~63u;
	return b > SCOPE_BLOCK ? SCOPE_BLOCK : b;
}

uint32_t scope_vss_mask_block(const ScopeTuning &tn)
{
	const u32 b = scope_block(tn.vss_block);
	return b >= 512u ? b : 0u; /* a thread's groups must fit the 8 bits of its mask byte */
}

/* lane groups per wave (log2): 8 streams for batches, one for launches of a few frames (see scope_launch_vss) */
static u32 vss_glog2(uint32_t slices, const ScopeTuning &tn)
{
	const u32 g = tn.vss_groups_log2 < 0 ? (slices >= 12 ? 0u : 3u) : (u32)tn.vss_groups_log2;
	return g > 3 ? 3 : g; /* lane groups of >= 8 lanes: every request still covers a whole 128-byte line */
}

/* Does this launch walk its planes in the ROWS form of scope_stream?  The kernels' own predicate (scope_rows_mode), asked on the
 * host with the launcher's own region split and lane groups: what scope_ctx_get_stat("vss.rows_launches" / "his.rows_launches")
 * counts, so that a test can tell that a padded plane took the form and did not fall back to the general one silently. */
bool scope_vss_rows_mode_host(const ScopeFrameSet &fs, uint32_t slices, const ScopeTuning &tn)
{
	const Regions r = split_regions(fs);
	return r.ncols_a != 0 && scope_rows_mode(fs.linesize, r.ncols_a, 16u, fs.height, slices, vss_glog2(slices, tn));
}
bool scope_his_rows_mode_host(const ScopeFrameSet &fs, uint32_t slices)
{
	const Regions r = split_regions(fs);
	return r.ncols_a != 0 && scope_rows_mode(fs.linesize, r.ncols_a, 16u, fs.height, slices, 0u);
}

bool scope_vss_can_fold(uint32_t slices, const uint8_t *pmask, const ScopeTuning &tn)
{
	return tn.vss_fold && pmask && slices >= 2 && slices <= VSS_FOLD_MAX && scope_block(tn.vss_block) == SCOPE_BLOCK;
}

hipError_t scope_launch_vss(const ScopeFrameSet &fs, uint32_t slices, int fuse_k0, int colorspace, uint8_t *dst,
			    uint8_t *pmask, const ScopeTuning &tn, hipStream_t st, const ScopeTiming *tm, const ScopeVssFoldSink *fold)
{
	/* Small launches -- one to a dozen frames, >= 12 workgroups per frame -- give every workgroup well under a megabyte:
	 * cutting that into 8 lane-group streams leaves runs of a few KiB per stream and the memory system answers
	 * 10-15 % slower (12 frames: 0.49 -> 0.59 of peak, 3 frames 0.36 -> 0.41; profiles/round2/latency_r2e.txt).
	 * There the slice is walked as ONE stream, two loads per lane in flight; the big batches keep 8 streams x 4. */
	const bool small = slices >= 12;
	const u32 glog2 = vss_glog2(slices, tn);
	/* the fused-K0 form: the pair-addressed table (HASH 2) everywhere, four loads per lane where the frames are many; small
	 * launches keep two (bench.py --frames 1 / 3 --workload vectorscope_rgb, us per call on distinct HBM-resident frames,
	 * visit s3: table as the plain kernel's + two loads 18.5 / 30.0, pair table + two loads 18.2 / 29.3, pair table +
	 * four loads 19.4 / 30.8) */
	const int fused_unroll = tn.vss_fused_unroll > 0 ? tn.vss_fused_unroll : (small ? 2 : 4);
	const int fused_hash = tn.vss_fused_hash >= 0 ? tn.vss_fused_hash : 2;
	const int unroll = fuse_k0 ? fused_unroll : (tn.vss_unroll > 0 ? tn.vss_unroll : (small ? 2 : 4));
	if (pmask && !scope_vss_mask_block(tn))
		return hipErrorInvalidValue;
	if (fold && !scope_vss_can_fold(slices, pmask, tn))
		return hipErrorInvalidValue;
	const VssFold vf = fold ? VssFold{fold->out, fold->tickets, fold->packed16 ? 1u : 0u} : VssFold{NULL, NULL, 0u};
	const VssArgs a = {&fs, slices, glog2, scope_block(tn.vss_block), tn.vss_rle != 0 ? 1u : 0u, split_regions(fs), scope_k0_matrix(colorspace),
			   dst, pmask, vf, st, tm};
	return vss_pick(unroll, tn.nt_loads, tn.vss_merge, fuse_k0, fuse_k0 ? fused_hash : tn.vss_hash, &a);
}

hipError_t scope_launch_his(const ScopeFrameSet &fs, uint32_t slices, uint32_t chmask, int fuse_k0, int colorspace,
			    uint32_t *dst, uint32_t *acc, uint32_t *tickets, const ScopeTuning &tn, hipStream_t st,
			    const ScopeTiming *tm)
{
	if (!tickets || slices < 2)
		acc = NULL;
	const HisArgs a = {&fs, slices, chmask, scope_block(tn.his_block), split_regions(fs), scope_k0_matrix(colorspace), dst, acc, tickets, st, tm};
	/* one 16-byte load per lane per batch keeps ~32 KiB per CU in flight, what HBM likes best in a batch; a one-frame
	 * launch (>= 100 slices per frame: a workgroup has ~170 KB to read in all) is about round trips: two */
	const int unroll = tn.his_unroll > 0 ? tn.his_unroll : ((slices >= 100 && !fuse_k0) ? 2 : 1);
	return his_pick(unroll, tn.nt_loads, fuse_k0, &a);
}

bool scope_wvs_vector_shape(const ScopeFrameSet &fs, const uint8_t *dst, const ScopeTuning &tn)
{
	/* The vector path (8-byte texel-pair loads, 16-byte table stores) needs width % 4 == 0, 8-byte aligned
	 * rows and a 16-byte aligned table; anything else takes the 4-byte path. */
	bool wide = tn.wvs_wide && (fs.width & 3u) == 0 && (fs.linesize & 7u) == 0 && (((size_t)dst) & 15u) == 0;
	for (u32 f = 0; wide && f < fs.n_frames; f++)
		wide = (((size_t)fs.plane[f]) & 7u) == 0;
	return wide;
}

uint32_t scope_wvs_strips(uint32_t width, int narrow)
{
	const u32 sw = narrow ? (u32)WVS_NARROW : SCOPE_WVS_STRIP;
	return (width + sw - 1u) / sw;
}

hipError_t scope_launch_wvs(const ScopeFrameSet &fs, uint32_t bands, uint32_t chmask, int fuse_k0, int colorspace,
			    uint8_t *dst, uint32_t num_cu, const ScopeTuning &tn, hipStream_t st, const ScopeHisSink *his,
			    const ScopeTiming *tm, int narrow)
{
	const u32 strips = scope_wvs_strips(fs.width, narrow);
	const u32 nch = (chmask & 1u) + ((chmask >> 1) & 1u) + ((chmask >> 2) & 1u);
	if (nch == 0)
		return hipErrorInvalidValue; /* the C-ABI zero-fills such tables itself */
	int wide = scope_wvs_vector_shape(fs, dst, tn) ? 1 : 0;
	if (narrow) {
		if (!wide)
			return hipErrorInvalidValue; /* narrow strips exist in the vector shape only: the caller checks */
		wide = 2;
	}
	/* persistent workgroups: as many as are resident at once (LDS- or thread-limited), each walks units */
	const u32 lds = nch * WVS_CH_WORDS * 4u + (his ? WVS_HIS_LDS_BYTES : 0u);
	u32 per_cu = max(1u, min((u32)(SCOPE_CU_LDS_BYTES / lds), (u32)(2048u / SCOPE_BLOCK)));
	if (tn.wvs_per_cu > 0 && (u32)tn.wvs_per_cu < per_cu)
		per_cu = (u32)tn.wvs_per_cu;
	const u32 resident = per_cu * max(1u, num_cu);
	const u32 total = fs.n_frames * strips * bands;
	u32 grid = (tn.wvs_persist && total > resident) ? resident : total;
	if (narrow && total > resident) {
		/* equal rounds for every workgroup, in whole blocks of 16 (the XCD pairing of wvs_unit) */
		const u32 rounds = (total + resident - 1u) / resident;
		grid = min(resident, (((total + rounds - 1u) / rounds) + 15u) & ~15u);
	}
	/* the deep queue across flushes: batches only -- every unit the same, with enough slots, several units per workgroup */
	if (wide == 1 && (!fuse_k0 || chmask == 2u) && tn.wvs_deep && bands == 1 && total > grid) {
		const u32 nbatch = ((fs.height >> 1) + WVS_U - 1) / WVS_U, nwaves = SCOPE_BLOCK / 64;
		const u32 slots = max(2u, ((nbatch + nwaves - 1) / nwaves + 1u) & ~1u);
		if (slots >= (u32)WVS_DEEP_SETS + 2u)
			wide = 3;
	}
	ScopeWvsHis h = {NULL, NULL, NULL, 0};
	if (his)
		h = {his->acc, his->tickets, his->out, his->chmask & chmask & 7u};
	const WvsArgs a = {&fs, strips, bands, lds, grid, scope_k0_matrix(colorspace), dst, h, st, tm};
	return wvs_pick(chmask, fuse_k0, wide, his != NULL, &a);
}

bool scope_dock_fits(const ScopeFrameSet &fs, const uint8_t *wvs_dst, uint32_t num_cu, const ScopeTuning &tn)
{
	const u32 strips = scope_wvs_strips(fs.width, 1);
	return scope_wvs_vector_shape(fs, wvs_dst, tn) && fs.height <= 8u * DOCK_G * (SCOPE_BLOCK / 64) && fs.n_frames * strips <= num_cu &&
	       strips <= 256u && (u64)fs.height * fs.linesize < (1ull << 31);
}

hipError_t scope_launch_dock(const ScopeFrameSet &fs, uint32_t chmask, int colorspace, uint8_t *wvs_dst, const ScopeHisSink *his,
			     uint8_t *vss_part, uint8_t *vss_mask, hipStream_t st, const ScopeTiming *tm)
{
	if ((chmask & 7u) == 0)
		return hipErrorInvalidValue;
	ScopeWvsHis h = {NULL, NULL, NULL, 0};
	if (his)
		h = {his->acc, his->tickets, his->out, his->chmask & chmask & 7u};
	const DockArgs a = {&fs, scope_wvs_strips(fs.width, 1), scope_k0_matrix(colorspace), wvs_dst, h, vss_part, vss_mask, st, tm};
	return his ? dock_pick_ch<true>(chmask, &a) : dock_pick_ch<false>(chmask, &a);
}

hipError_t scope_launch_sat_reduce(const uint8_t *part, uint8_t *out, uint32_t n_frames, uint32_t slices, uint64_t len,
				   int packed16, hipStream_t st)
{
	const u64 len16 = len / 16;
	const u32 blocks = (u32)((len16 + 255) / 256);
	hipLaunchKernelGGL(scope_sat_reduce_kernel, dim3(blocks, n_frames), dim3(256), 0, st, part, out, slices, len16,
			   packed16 ? 1u : 0u);
	return hipGetLastError();
}

hipError_t scope_launch_sat_reduce_sparse(const uint8_t *part, const uint8_t *pmask, uint8_t *out, uint32_t n_frames,
					  uint32_t slices, uint32_t mask_block, int packed16, hipStream_t st)
{
	u32 sl_log2 = 0;
	while (sl_log2 < 4 && (1u << sl_log2) < slices)
		sl_log2++;
	const u32 gpw = 256u >> sl_log2;
	const u32 lds = sl_log2 ? 8192u + 32u * gpw : 0u;
	const dim3 grid(SCOPE_VSS_BINS / 16 / gpw, n_frames);
	const u32 per_lane = (slices + (1u << sl_log2) - 1u) >> sl_log2; /* slices a slice lane takes: <= 16 for <= 256 slices */
#define SCOPE_REDUCE_K(K)                                                                                               \
	hipLaunchKernelGGL(scope_sat_reduce_sparse_kernel<K>, grid, dim3(256), lds, st, part, pmask, out, slices, mask_block, \
			   sl_log2, packed16 ? 1u : 0u)
	if (per_lane <= 1)
		SCOPE_REDUCE_K(1); /* the 64-frame batches: 4 slices per frame */
	else if (per_lane <= 2)
		SCOPE_REDUCE_K(2);
	else if (per_lane <= 4)
		SCOPE_REDUCE_K(4);
	else if (per_lane <= 6)
		SCOPE_REDUCE_K(6); /* the ring of three: 80 / 85 slices per frame */
	else if (per_lane <= 8)
		SCOPE_REDUCE_K(8);
	else if (per_lane <= 10)
		SCOPE_REDUCE_K(10); /* one frame over staged planes: 160 slices */
	else if (per_lane <= 12)
		SCOPE_REDUCE_K(12);
	else
		SCOPE_REDUCE_K(16); /* one frame, 240 / 256 slices */
#undef SCOPE_REDUCE_K
	return hipGetLastError();
}

hipError_t scope_launch_sum_reduce(const uint32_t *part, uint32_t *out, uint32_t n_frames, uint32_t slices,
				   uint32_t len, hipStream_t st)
{
	hipLaunchKernelGGL(scope_sum_reduce_kernel, dim3((len + 255) / 256, n_frames), dim3(256), 0, st, part, out, slices,
			   len);
	return hipGetLastError();
}

hipError_t scope_launch_widen(const uint8_t *in, uint32_t *out, uint64_t n, hipStream_t st)
{
	u32 blocks = (u32)((n + 255) / 256);
	if (blocks > 2048)
		blocks = 2048;
	if (!blocks)
		blocks = 1;
	hipLaunchKernelGGL(scope_widen_kernel, dim3(blocks), dim3(256), 0, st, in, out, (u64)n);
	return hipGetLastError();
}

hipError_t scope_launch_clamp(const uint32_t *in, uint8_t *out, uint64_t n, hipStream_t st)
{
	u32 blocks = (u32)((n + 255) / 256);
	if (blocks > 2048)
		blocks = 2048;
	if (!blocks)
		blocks = 1;
	hipLaunchKernelGGL(scope_clamp_kernel, dim3(blocks), dim3(256), 0, st, in, out, (u64)n);
	return hipGetLastError();
}

hipError_t scope_launch_pack16(const uint8_t *in, uint32_t *out, uint64_t n_pairs, hipStream_t st)
{
	u32 blocks = (u32)((n_pairs + 255) / 256);
	if (blocks > 2048)
		blocks = 2048;
	if (!blocks)
		blocks = 1;
	hipLaunchKernelGGL(scope_pack16_kernel, dim3(blocks), dim3(256), 0, st, in, out, (u64)n_pairs);
	return hipGetLastError();
}

hipError_t scope_launch_unpack16_clamp(const uint32_t *in, uint8_t *out, uint64_t n_pairs, hipStream_t st)
{
	u32 blocks = (u32)((n_pairs + 255) / 256);
	if (blocks > 2048)
		blocks = 2048;
	if (!blocks)
		blocks = 1;
	hipLaunchKernelGGL(scope_unpack16_clamp_kernel, dim3(blocks), dim3(256), 0, st, in, out, (u64)n_pairs);
	return hipGetLastError();
}
