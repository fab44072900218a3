/*
 * scope_kernels.h -- launch-side interface of the gfx950 scope-accumulation kernels.
 *
 * Internal to libscope_hip.so (the C-ABI is include/scope_hip.h).  Everything here works on
 * DEVICE pointers and enqueues on a hipStream_t; no allocation, no synchronisation.
 *
 * Reference semantics implemented (all relative to /root/reference):
 *   K0  data/common.effect:23-43      RGB -> Y'CbCr (fp32, no contraction, UNORM8 store)
 *   K1  src/vectorscope.c:217-238     256x256 CbCr scatter, u8 saturating
 *   K2  src/waveform.c:220-257        per-column 256-bin histograms, u8 saturating, alpha test
 *   K3  src/histogram.c:357-395       3x256 u32 histogram, alpha test
 */
#ifndef SCOPE_KERNELS_H
#define SCOPE_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#define SCOPE_MAX_FRAMES 64      /* frames per launch (pointers travel as kernel arguments) */
#define SCOPE_VSS_BINS 65536u    /* VS_SIZE * VS_SIZE, src/vectorscope.c:26 */
#define SCOPE_HIS_WORDS 1024u    /* HI_SIZE * 4, src/histogram.c:20 */
#define SCOPE_WVS_ROWS 256u       /* WV_SIZE, src/waveform.c:20 */
#define SCOPE_BLOCK 1024         /* threads per workgroup of every accumulation kernel */
#define SCOPE_CU_LDS_BYTES 163840u /* gfx950: 160 KiB per CU */
#define SCOPE_WVS_STRIP 64u      /* columns owned by one waveform workgroup (= one wave wide) */

/* A batch of same-geometry BGRA8 planes resident in HBM. */
struct ScopeFrameSet {
	const uint8_t *plane[SCOPE_MAX_FRAMES];
	uint32_t n_frames;
	uint32_t linesize; /* bytes */
	uint32_t width, height;
};

/* K0 matrix, rows U,Y,V; columns r,g,b (signed, so every term is an add).  U and V coefficients
 * of one column sit next to each other: the kernels use them as packed pairs (v_pk_mul_f32). */
struct ScopeK0 {
	float ur, vr;
	float ug, vg;
	float ub, vb;
	float yr, yg, yb;
	/* The same rows in fixed point (coefficient * 2^24, last entry = the constant term): the fast path of the
	 * fused kernels, see scope_k0_fixed in scope_kernels.hip. */
	int fu[4], fy[4], fv[4];
};
#define SCOPE_K0_FX_SHIFT 24  /* the output byte is byte 3 of the sum */
#define SCOPE_K0_FX_GUARD 512 /* results closer than this (in 2^-24 of an output step) to a step are recomputed in fp32 */

/* Tunables the host may flip for A/B runs (scope_ctx_set_option). */
struct ScopeTuning {
	int vss_merge;       /* 0 plain adds, 1|2 absorb mode always, 3 absorb mode by wave vote (default) */
	int vss_unroll;      /* 16-byte loads per lane per batch (two batches in flight): 1, 2 or 4; 0 = by launch size */
	int vss_fused_unroll; /* the same for the fused-K0 form; 0 = by launch size (4 in batches, 2 in small launches) */
	int vss_hash;        /* bin -> LDS word placement, see scope_kernels.hip */
	int vss_fused_hash;  /* the same for the fused-K0 form: 2 = rows of 136 words, two pixels' addresses per register; -1 = the default (2) */
	int vss_groups_log2; /* lane groups per wave walking separate image regions: 0..3; -1 = by launch size */
	int his_unroll;      /* 1, 2 or 4; 0 = by launch size.  More bytes in flight than ~32 KiB per CU LOWER the achieved bandwidth of a batch */
	int vss_block, his_block; /* threads per workgroup (0 = SCOPE_BLOCK) */
	int fold;            /* 1: the slices of a histogram frame are combined inside the launch (atomics + last arrival); 0: partial
			      * tables + scope_launch_sum_reduce */
	int vss_sparse;      /* 1: vectorscope partials are written sparse (flagged 16-bin groups only) */
	int vss_fold;        /* 1: launches of 2 .. 4 slices per frame (the 64-frame batches) combine a frame's partial tables inside the
			      * launch (last arrival; scope_device.h, vss_write_out) instead of in scope_sat_reduce_sparse_kernel */
	int wvs_per_cu;      /* cap on persistent waveform workgroups per CU (0 = as many as fit); 1 keeps ~32 KiB of
			      * loads in flight per CU, which is what the HBM likes best (+5 % for 1-2 channels) */
	int wvs_persist;     /* 1: one resident set of workgroups walks all units with a continuous load stream */
	int wvs_wide;        /* 1: 8-byte texel-pair loads where width and alignment allow; 0: always 4-byte */
	int wvs_deep;        /* 1: batches keep WVS_DEEP_SETS more loads in flight across each unit's flush */
	int dock;            /* scope_three_scopes_batch: 1 = one read for any launch (scope_trio_kernel); 2 = round 2's form, one read
			      * for launches of one round of strips only (scope_dock_kernel), two passes otherwise; 0 = always two passes */
	int wvs_strip;       /* 0: 16-column strips for launches that would otherwise need row bands, 64 elsewhere; 64 / 16: always */
	int nt_loads;        /* 1: non-temporal pixel loads */
	int trio_strip;      /* scope_trio_kernel's strip width: 0 = 32 columns for batches, 16 for small launches; 16 / 32: always */
	int vss_rle;         /* the fused-K0 vectorscope under vss_merge 3, and scope_trio_kernel's vectorscope half: 0 = run-length mode by wave
			      * vote, 1 = always (tests) */
};

/* Optional last argument of the accumulation launchers: two HIP events that receive the kernel's own begin and end time
 * stamps (hipExtLaunchKernelGGL), for scope_ctx_profile. */
struct ScopeTiming {
	hipEvent_t start, stop;
};

ScopeK0 scope_k0_matrix(int colorspace);
ScopeK0 scope_k0_matrix_full_range(int colorspace); /* for the self-test of scope_trio_kernel's chroma guard only */
/* host-side mirrors of the kernels' ROWS-form predicate (scope_ctx_get_stat) and the guard's self-test */
bool scope_vss_rows_mode_host(const ScopeFrameSet &fs, uint32_t slices, const ScopeTuning &tn);
bool scope_his_rows_mode_host(const ScopeFrameSet &fs, uint32_t slices);
uint32_t scope_trio_chroma_selftest(void); /* bit 0 / 1: BT.601 / BT.709 fit the compact table; bit 2 / 3: their full-range forms are refused */

/* Dynamic-LDS opt-in for every instantiation; call once per device before launching (scope_trio_init: the part of it that
 * lives in scope_trio.hip; scope_kernels_init calls it). */
hipError_t scope_kernels_init(void);
hipError_t scope_trio_init(void);

/* K0 standalone: src plane (B,G,R,A) -> dst plane (U,Y,V,255); replaces the ConvertRGB_YUV* pass
 * of render_rgb_yuv (src/common.c:202-213). */
hipError_t scope_launch_k0(const uint8_t *src, uint32_t src_linesize, uint8_t *dst, uint32_t dst_linesize,
			   uint32_t width, uint32_t height, int colorspace, hipStream_t st);

/* K1.  dst = uint8[n_frames][slices][65536].  pmask == NULL: complete clamped tables (final when slices == 1, else
 * partials for scope_launch_sat_reduce).  pmask = uint8[n_frames][slices][scope_vss_mask_block(tn)]: SPARSE partials
 * -- only the 16-bin groups holding a count are written and flagged -- for scope_launch_sat_reduce_sparse.
 * fuse_k0: planes are RGB, U/V computed in-kernel. */
/* fold != NULL (scope_vss_can_fold: sparse partials, 2 .. 4 slices per frame, 1024-thread workgroups): the slices of a frame meet
 * INSIDE the launch -- the workgroup that arrives last at the frame's ticket adds the other slices' partials to its own and writes
 * the final table to fold->out[frame] (uint8[65536], or uint32[32768] with packed16); no reduction kernel follows.  tickets =
 * uint32[SCOPE_MAX_FRAMES], zero before the first launch and left zero by every launch. */
struct ScopeVssFoldSink {
	uint8_t *out;
	uint32_t *tickets;
	int packed16;
};
bool scope_vss_can_fold(uint32_t slices, const uint8_t *pmask, const ScopeTuning &tn);
hipError_t scope_launch_vss(const ScopeFrameSet &fs, uint32_t slices, int fuse_k0, int colorspace, uint8_t *dst,
			    uint8_t *pmask, const ScopeTuning &tn, hipStream_t st, const ScopeTiming *tm = NULL,
			    const ScopeVssFoldSink *fold = NULL);
/* Threads per workgroup of the vectorscope launch = bytes per sparse-partial mask; 0: this tuning cannot write masks. */
uint32_t scope_vss_mask_block(const ScopeTuning &tn);

/* K3.  With acc == NULL (or slices == 1) dst = uint32[n_frames][slices][1024]: partials for scope_launch_sum_reduce,
 * or the final tables when slices == 1.  With acc = uint32[n_frames][1024] and tickets = uint32[n_frames] -- both zero
 * before the first launch and left zero by every launch -- the slices of a frame are combined inside the launch
 * (device-scope atomic adds, the last workgroup of a frame moves the sums out) and dst = uint32[n_frames][1024].
 * chmask bit c = count byte c of the texel (bit0 -> slot 2, bit1 -> slot 1, bit2 -> slot 0). */
hipError_t scope_launch_his(const ScopeFrameSet &fs, uint32_t slices, uint32_t chmask, int fuse_k0, int colorspace,
			    uint32_t *dst, uint32_t *acc, uint32_t *tickets, const ScopeTuning &tn, hipStream_t st,
			    const ScopeTiming *tm = NULL);

/* Capture pre-steps: dst(i, j) = texel at src + j*row_pitch + i*col_step*4, i < cx, j < cy (the caller folds
 * scale and ROI origin into src / row_pitch / col_step).  All pointers and pitches 4-byte aligned. */
hipError_t scope_launch_stage(const uint8_t *src, uint64_t row_pitch, uint32_t col_step, uint32_t cx, uint32_t cy,
			      uint8_t *dst, uint32_t dst_linesize, hipStream_t st);

/* K2.  dst = uint8[n_frames][bands][256][width][4]; combine with scope_launch_sat_reduce when
 * bands > 1.  chmask as above (bit c -> byte c of the output texel).
 * his != NULL: the same pass also leaves the K3 count tables of the frames in his->out (uint32[n_frames][256][4], the
 * layout scope_launch_his writes), for the channels his->chmask selects among those of chmask -- no second read of
 * the planes.  acc = uint32[SCOPE_MAX_FRAMES][1024] and tickets = uint32[SCOPE_MAX_FRAMES] are zero before the first
 * launch and left zero by every launch. */
struct ScopeHisSink {
	uint32_t *acc, *tickets, *out;
	uint32_t chmask;
};
hipError_t scope_launch_wvs(const ScopeFrameSet &fs, uint32_t bands, uint32_t chmask, int fuse_k0, int colorspace,
			    uint8_t *dst, uint32_t num_cu, const ScopeTuning &tn, hipStream_t st,
			    const ScopeHisSink *his = NULL, const ScopeTiming *tm = NULL, int narrow = 0);
/* narrow != 0: strips of 16 columns instead of 64 -- four times the units, for launches of one or two frames, which
 * then need no row bands and no partial tables (vector shape only: scope_wvs_vector_shape must hold for fs and dst). */
bool scope_wvs_vector_shape(const ScopeFrameSet &fs, const uint8_t *dst, const ScopeTuning &tn);
uint32_t scope_wvs_strips(uint32_t width, int narrow);

/* The dock's three scopes of every frame from ONE read of its RGB plane (src/roi.c:329-341): the strip's texels stay in
 * registers and are walked twice -- waveform (+ the histogram riding on its flush, his != NULL) into wvs_dst =
 * uint8[n_frames][256][width][4], then the vectorscope through the fused K0 into sparse partial tables, one per 16-column
 * strip: vss_part / vss_mask as scope_launch_vss writes them with slices = scope_wvs_strips(width, 1) and 1024 mask bytes
 * per slice, for scope_launch_sat_reduce_sparse.  Only where scope_dock_fits says so (vector shape, one round of strips on
 * the chip, at most 2176 rows); chmask selects the waveform's RGB channels. */
bool scope_dock_fits(const ScopeFrameSet &fs, const uint8_t *wvs_dst, uint32_t num_cu, const ScopeTuning &tn);
hipError_t scope_launch_dock(const ScopeFrameSet &fs, uint32_t chmask, int colorspace, uint8_t *wvs_dst, const ScopeHisSink *his,
			     uint8_t *vss_part, uint8_t *vss_mask, hipStream_t st, const ScopeTiming *tm = NULL);

/* The same three tables from ONE read for any number of frames (scope_trio_kernel): both the vectorscope's table and a
 * one-replica waveform table of a 16-column strip are resident in LDS, a workgroup streams the strips of one slice of one
 * frame.  slices = scope_trio_slices(fs, num_cu) workgroups per frame; slices == 1: vss_dst = the final uint8[n][65536]
 * tables, vss_mask unused; else vss_dst / vss_mask = sparse partials as scope_launch_vss writes them with 1024 mask bytes
 * per slice, for scope_launch_sat_reduce_sparse.  Where scope_trio_fits says so (vector shape, height <= 65535). */
bool scope_trio_fits(const ScopeFrameSet &fs, const uint8_t *wvs_dst, const ScopeTuning &tn);
/* strip width of the launch (32 columns for batches, 16 for the reference's call shape; tn.trio_strip forces one), the slices
 * per frame that go with it, and the launch */
uint32_t scope_trio_strip_width(const ScopeFrameSet &fs, uint32_t num_cu, const ScopeTuning &tn);
uint32_t scope_trio_slices(const ScopeFrameSet &fs, uint32_t num_cu, uint32_t strip_width);
hipError_t scope_launch_trio(const ScopeFrameSet &fs, uint32_t slices, uint32_t strip_width, uint32_t chmask, int colorspace,
			     uint8_t *wvs_dst, const ScopeHisSink *his, uint8_t *vss_dst, uint8_t *vss_mask, hipStream_t st,
			     const ScopeTiming *tm = NULL, uint32_t rle_force = 0 /* the vectorscope's run-length form on every slot (tests) */,
			     const ScopeVssFoldSink *fold = NULL /* 2 .. 4 slices: the frame's vectorscope partials meet inside the launch, as scope_launch_vss */);

/* out[f][i] = min(sum_s part[f][s][i], 255), i < len (len % 16 == 0, slices <= 256).  packed16: the result is written
 * as uint32[len / 2], two bins per word (lo | hi << 16) -- the wire format of the multi-GPU table sum. */
hipError_t scope_launch_sat_reduce(const uint8_t *part, uint8_t *out, uint32_t n_frames, uint32_t slices, uint64_t len,
				   int packed16, hipStream_t st);

/* The same over sparse partials: part / pmask as scope_launch_vss wrote them with `mask_block` threads per workgroup. */
hipError_t scope_launch_sat_reduce_sparse(const uint8_t *part, const uint8_t *pmask, uint8_t *out, uint32_t n_frames,
					  uint32_t slices, uint32_t mask_block, int packed16, hipStream_t st);

hipError_t scope_launch_sum_reduce(const uint32_t *part, uint32_t *out, uint32_t n_frames, uint32_t slices,
				   uint32_t len, hipStream_t st);

/* Multi-GPU helpers: u8 table -> u32 (for the RCCL sum) and u32 -> u8 with the final clamp. */
hipError_t scope_launch_widen(const uint8_t *in, uint32_t *out, uint64_t n, hipStream_t st);
hipError_t scope_launch_clamp(const uint32_t *in, uint8_t *out, uint64_t n, hipStream_t st);
/* Same idea at half the wire bytes: two bins per 32-bit word (lo | hi << 16).  A sum over up to
 * 257 ranks of values <= 255 stays below 65536, so the halves never carry into each other. */
hipError_t scope_launch_pack16(const uint8_t *in, uint32_t *out, uint64_t n_pairs, hipStream_t st);
hipError_t scope_launch_unpack16_clamp(const uint32_t *in, uint8_t *out, uint64_t n_pairs, hipStream_t st);

/* D1-D3: the display pixel shaders (data/vectorscope.effect, waveform.effect, histogram.effect) as
 * output-sized post-passes; out = RGBA8 bytes R,G,B,A.  display: 0 overlay, 1 stack, 2 parade. */
hipError_t scope_launch_display_vss(const uint8_t *table, uint8_t *out, int intensity, int color_type, int colorspace,
				    float zoom, hipStream_t st);
hipError_t scope_launch_display_wvs(const uint8_t *table, uint8_t *out, uint32_t width, uint32_t n_components,
				    int display, int intensity, uint32_t out_w, uint32_t out_h, hipStream_t st);
hipError_t scope_launch_display_his(const float *tex, uint8_t *out, float hm0, float hm1, float hm2,
				    uint32_t n_components, int display, uint32_t out_w, uint32_t out_h, hipStream_t st);


/* Graticule over a finished picture (RGBA8, bytes R,G,B,A): n_quads label quads (8 floats each: x, y, w, h, u0, v0, u1,
 * v1) then n_segs line segments (4 floats each), blended in list order; strip = the 96x16 glyph texels of
 * scope_graticule_strip on the device; argb as gs_effect_set_color takes it.  Rasterisation rule: scope_display.hip. */
hipError_t scope_launch_graticule(uint8_t *pic, uint32_t w, uint32_t h, const float *quads, uint32_t n_quads, const float *segs,
				  uint32_t n_segs, const uint8_t *strip, uint32_t argb, hipStream_t st);
void scope_graticule_strip(uint8_t *alpha /* [16][96] */);

#endif
