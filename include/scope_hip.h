/*
 * scope_hip.h -- C-ABI of libscope_hip.so, the MI355X (gfx950) implementation of
 * obs-color-monitor's scope-accumulation path.
 *
 * Plain C: opaque context, plain pointers and sizes, int status codes; no C++ or torch types
 * cross this boundary and nothing here throws or aborts.  Each entry point names the reference
 * interface it replaces (file:line relative to the obs-color-monitor source tree, v0.9.5).
 *
 * What the library replaces: the reference stages the converted frame to host memory
 * (gs_stage_texture + gs_stagesurface_map, src/common.c:316-347) and then runs one scalar CPU
 * loop per scope on the plugin's "color-monitor" thread (src/common.c:366-368 calling
 * vss_draw_vectorscope / wvs_draw_waveform / his_draw_histogram).  Here the frame is reduced on
 * the GPU and only the small result tables come back.
 *
 * Threading: a scope_ctx may be shared by any number of threads: every entry point takes the
 * context's internal lock, so calls are serialised per context (the reference runs one
 * "color-monitor" thread per stand-alone source, src/common.c:375-403, plus the render thread).
 * Asynchronous device-pointer calls (SCOPE_ASYNC) are ordered on the context's stream.  Every
 * entry point selects the context's device itself; no current-device state is assumed.
 *
 * Errors: every int-returning function returns SCOPE_OK (0) or a negative SCOPE_E* code and
 * leaves the output buffers untouched on failure (the reference's callbacks are void and keep
 * the previous table on screen when something fails, src/vectorscope.c:252, src/common.c:349-350).
 */
#ifndef SCOPE_HIP_H
#define SCOPE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SCOPE_ABI_VERSION 2

/* ---- status codes (negative errno style) -------------------------------------------------- */
#define SCOPE_OK 0
#define SCOPE_EINVAL (-22)  /* bad argument (NULL, zero size, misaligned device pointer, ...) */
#define SCOPE_ENOMEM (-12)  /* host or device allocation failed */
#define SCOPE_ENODEV (-19)  /* no usable HIP device */
#define SCOPE_EIO (-5)      /* a HIP runtime call failed; see scope_last_error() */
#define SCOPE_ENODATA (-61) /* the plane the scope needs is NULL: nothing was done */
#define SCOPE_EAGAIN (-11)  /* pipeline: every ring slot is in flight (collect first) / nothing to collect */

/* ---- flags -------------------------------------------------------------------------------- */
#define SCOPE_IN_DEVICE 0x1u  /* plane pointers are device (HBM) pointers; default: host */
#define SCOPE_OUT_DEVICE 0x2u /* result pointers are device pointers; default: host */
#define SCOPE_ASYNC 0x4u      /* enqueue only; only valid with IN_DEVICE|OUT_DEVICE; scope_ctx_sync() later */
#define SCOPE_FUSE_K0 0x8u    /* yuv_data is NULL: derive Y'CbCr from rgb_data inside the kernel (K0 fused) */
#define SCOPE_OUT_PACKED16 0x10u /* scope_vectorscope_batch with SCOPE_OUT_DEVICE: out = uint32[n][32768], two bins per
			       * word (lo | hi << 16) -- the wire format of the multi-GPU table sum, saving the
			       * scope_table_pack_u8_u16x2 pass */
#define SCOPE_SIDE_REDUCE 0x20u /* scope_vectorscope_batch with SCOPE_ASYNC: where the launch needs a partial-table reduction (few
				 * frames per call: the reference's call shape, src/common.c:366-368) that reduction is enqueued on
				 * a SECOND stream of the context, ordered behind the accumulation by an event, and the context's
				 * stream is free for the next call at once: the reduction of call i runs under the accumulation of
				 * call i + 1 (two sets of partial tables).  The output table of such a call is complete only after
				 * scope_ctx_fence() (no host synchronisation), scope_ctx_sync(), or the next call on the context
				 * without this flag -- which fences first. */

/* ---- component masks, identical to the reference ------------------------------------------- */
#define SCOPE_COMP_RGB 0x07u /* src/waveform.c:26, src/histogram.c:27 */
#define SCOPE_COMP_Y 0x20u   /* src/waveform.c:27 */
#define SCOPE_COMP_UV 0x50u  /* src/waveform.c:28 */
#define SCOPE_COMP_YUV 0x70u /* src/waveform.c:29 */

#define SCOPE_VS_SIZE 256 /* src/vectorscope.c:26 */
#define SCOPE_WV_SIZE 256 /* src/waveform.c:20 */
#define SCOPE_HI_SIZE 256 /* src/histogram.c:20 */

typedef struct scope_ctx scope_ctx;

/* Frame descriptor.  Field-for-field struct cm_surface_data (src/common.h:24-30), so a
 * `struct cm_surface_data *` may be passed as is.  Planes are BGRA8 rows: rgb_data bytes
 * B,G,R,A; yuv_data bytes U,Y,V,255 (data/common.effect:23-43).  linesize >= 4*width. */
struct scope_surface_data {
	uint8_t *rgb_data, *yuv_data;
	uint32_t linesize, width, height;
	int colorspace; /* 1 = BT.601, otherwise BT.709 (src/common.c:207-208) */
	void *tex;      /* unused, as in the reference */
};

/* A batch of same-geometry planes (several sources, or a queue of frames of one source:
 * the reference's 3-slot ring, src/common.h:46).  planes[i] is one BGRA8 plane. */
struct scope_batch {
	const uint8_t *const *planes;
	uint32_t n_frames;
	uint32_t linesize, width, height;
	int planes_are_rgb; /* 0: planes hold U,Y,V,255; 1: planes hold B,G,R,A */
	int colorspace;     /* used when K0 is fused */
};

struct scope_device_info {
	char name[128];
	char arch[32];
	int device;
	int compute_units;
	int lds_bytes_per_cu;
	int clock_khz;
	uint64_t hbm_bytes;
};

/* ---- context ------------------------------------------------------------------------------- */

/* Creates a context on HIP device `device`; owns a stream and grow-only device scratch. */
int scope_ctx_create(int device, scope_ctx **out);
void scope_ctx_destroy(scope_ctx *ctx);
/* Run on the caller's hipStream_t (e.g. torch's current stream); NULL restores the own stream. */
int scope_ctx_set_stream(scope_ctx *ctx, void *hip_stream);
void *scope_ctx_get_stream(scope_ctx *ctx);
int scope_ctx_sync(scope_ctx *ctx);
/* Makes the context's stream wait (on the device: an event, no host synchronisation) for whatever SCOPE_SIDE_REDUCE calls
 * still have on the context's second stream.  After it, work enqueued on the context's stream sees their tables. */
int scope_ctx_fence(scope_ctx *ctx);
int scope_ctx_device_info(scope_ctx *ctx, struct scope_device_info *info);
/* Tunables for A/B measurements (the defaults are the measured best, profiles/):
 *   "vss.merge" 0|2|3        plain adds | absorb mode always | absorb mode by wave vote
 *   "vss.rle" 0|1            the vectorscope kernels' run-length mode (flat cells as runs across lanes; the batch form of the kernel,
 *                            fused K0 and staged planes): by wave vote (0) or on every content (1, tests)
 *   "vss.unroll" 0|1|2|4, "vss.fused_unroll" 0|1|2|4, "his.unroll" 0|1|2|4   16-byte loads per lane per batch (0: by launch size)
 *   "vss.fused_hash" -1|0|1|2   the fused-K0 vectorscope's LDS table: as "vss.hash" (0, 1) or rows of 136 words with the LDS
 *                              operands of two pixels worked out in one register (2); -1: the default, 2
 *   "vss.hash" 0|1, "vss.groups" -1|0..3   table rows XOR-swizzled across the LDS banks (0) or not (1); log2 of lane groups per
 *                wave (-1: 8 groups for batches, 1 for launches of a few frames)
 *   "vss.block" / "his.block" 0|64..1024   threads per workgroup (0 = 1024)
 *   "wvs.wide" 0|1, "wvs.persist" 0|1, "wvs.per_cu" 0|1|2   waveform load shape / persistent walk / workgroups per CU
 *   "wvs.strip" 0|16|64      waveform strip width: 16 columns for launches that would otherwise need row bands (0), or fixed
 *   "wvs.deep" 0|1           batches: more loads in flight across each unit's table flush
 *   "dock" 0|1|2             scope_three_scopes_batch: one read of the planes for any launch (1, scope_trio_kernel), round 2's form
 *                            (2: one read for one-round launches only) or always two passes (0)
 *   "trio.strip" 0|16|32     scope_trio_kernel's strip width: 32 columns (whole 128-byte lines, one waveform replica) for batches
 *                            and 16 (two replicas; four for a one-channel mask) for launches of a few frames (0), or fixed
 *   "fold" 0|1   the slices of a histogram frame meet inside the launch (atomics + last arrival) | partial tables + a second kernel
 *   "vss.sparse" 0|1   vectorscope partial tables hold only the 16-bin groups that are not empty
 *   "vss.fold" 0|1     launches of 2 .. 4 workgroups per frame (the 64-frame batches) add a frame's partial vectorscope tables up inside
 *                      the launch (the workgroup that arrives last) | in a second kernel
 *   "nt" 0|1     streaming (non-temporal) loads
 *   "slices"     workgroups per frame (0 = automatic) */
int scope_ctx_set_option(scope_ctx *ctx, const char *key, int value);
/* Counters for tests and A/B runs: which form the launches of this context took -- so that a path a test means to cover cannot fall
 * back to another one silently.  "vss.rows_launches" / "his.rows_launches": vectorscope / histogram launches that walked padded
 * planes in whole rows (the ROWS form; the kernels' own predicate asked on the host); "vss.fold_launches" / "trio.fold_launches":
 * launches whose partial vectorscope tables met inside the launch; "trio.chroma_guard_selftest": bit 0 / 1 = BT.601 / BT.709 fit
 * the three-scope kernel's compact table, bit 2 / 3 = their full-range forms are refused by the launch-time guard (15 expected).
 * No reference counterpart. */
int scope_ctx_get_stat(scope_ctx *ctx, const char *key, uint64_t *value);
const char *scope_strerror(int status);
const char *scope_last_error(scope_ctx *ctx);
int scope_abi_version(void);

/* ---- K0: RGB -> Y'CbCr plane ---------------------------------------------------------------
 * Replaces the ConvertRGB_YUV601/709 pass of render_rgb_yuv (src/common.c:202-213,
 * data/common.effect:23-61).  rgb and yuv are both host or both device (SCOPE_IN_DEVICE). */
int scope_convert_rgb_yuv(scope_ctx *ctx, const uint8_t *rgb, uint32_t rgb_linesize, uint8_t *yuv,
			  uint32_t yuv_linesize, uint32_t width, uint32_t height, int colorspace, uint32_t flags);

/* ---- capture-side pre-steps: target_scale and ROI crop --------------------------------------
 * Replaces the scaled render + sub-region draw of cm_render_target: the target is rendered at
 * 1/target_scale of its size (src/common.c:249-250, :141-168) and an ROI source with a valid rectangle
 * stages only that rectangle, given in SCALED coordinates (:272-282, :202-206; src/roi.c:478-499).
 * The reference leaves the down-scale to the target source's sampler; here it is pinned to point
 * sampling at the centre of each scale x scale cell (integer only):
 *   dst(i, j) = src((x + i) * scale + scale / 2, (y + j) * scale + scale / 2),  i < cx, j < cy.
 * cx == 0 or cy == 0 selects the whole scaled frame (x, y ignored), which is what the reference does
 * for a non-ROI source or an invalid rectangle.  SCOPE_EINVAL when scale == 0, the scaled frame is
 * empty or the rectangle leaves it. */
struct scope_region {
	uint32_t scale;        /* target_scale, 1..128 in the reference's UI (src/common.c:123) */
	uint32_t x, y, cx, cy; /* ROI origin and size in scaled texels */
};
/* Size of the staged plane for a width x height target. */
int scope_region_size(uint32_t width, uint32_t height, const struct scope_region *region, uint32_t *out_width,
		      uint32_t *out_height);
/* src and dst are both host or both device (SCOPE_IN_DEVICE); planes of BGRA8 texels. */
int scope_stage_region(scope_ctx *ctx, const uint8_t *src, uint32_t src_linesize, uint32_t width, uint32_t height,
		       const struct scope_region *region, uint8_t *dst, uint32_t dst_linesize, uint32_t flags);

/* ---- one upload, several scopes ------------------------------------------------------------
 * The ROI / dock source hands ONE staged surface to every registered scope (roi_surface_cb, src/roi.c:329-341); the
 * three loops then read the same host planes.  Between scope_surface_begin() and scope_surface_end() the synchronous
 * single-surface calls (scope_vectorscope / scope_waveform / scope_histogram_counts / scope_histogram with host planes)
 * upload each distinct plane pointer once and reuse the device copy: the dock's three scopes cost two uploads (RGB,
 * YUV) instead of three -- or one with SCOPE_FUSE_K0 here, which never uploads the YUV plane and lets scopes that want
 * it convert the RGB plane on the fly (equal to the YUV plane this build's own K0 makes; the plugin's plane comes from
 * OBS's shader, so this is opt-in).  All planes inside one begin/end share one geometry; at most two distinct planes.
 * The context is locked from begin to end (other threads' calls wait), and every entry point of a context is
 * serialised, so one context can be shared by the plugin's pipeline threads and its render thread.
 * scope_surface_end reports how many planes were uploaded and how many uploads were saved (both may be NULL). */
int scope_surface_begin(scope_ctx *ctx, uint32_t flags);
int scope_surface_end(scope_ctx *ctx, uint32_t *uploads, uint32_t *reuses);

/* ---- K1: vectorscope -----------------------------------------------------------------------
 * Replaces vss_draw_vectorscope(dbuf, surface_data) (src/vectorscope.c:217-238).
 * out = uint8[256][256].  Needs sd->yuv_data, or sd->rgb_data with SCOPE_FUSE_K0. */
int scope_vectorscope(scope_ctx *ctx, const struct scope_surface_data *sd, uint32_t flags, uint8_t *out);

/* ---- K2: waveform --------------------------------------------------------------------------
 * Replaces wvs_draw_waveform(src, dbuf, surface_data) (src/waveform.c:220-257);
 * `components` is src->components.  out = uint8[256][width][4]. */
int scope_waveform(scope_ctx *ctx, const struct scope_surface_data *sd, uint32_t components, uint32_t flags,
		   uint8_t *out);

/* ---- K3: histogram counts ------------------------------------------------------------------
 * The counting loop of his_draw_histogram (src/histogram.c:357-395).
 * counts = uint32[256][4], slots R,G,B,unused. */
int scope_histogram_counts(scope_ctx *ctx, const struct scope_surface_data *sd, uint32_t components, uint32_t flags,
			   uint32_t *counts);

/* ---- K2 + K3 in one pass over the plane ------------------------------------------------------
 * The dock hands ONE staged surface to every registered scope (roi_surface_cb, src/roi.c:329-341) and the reference
 * then walks it once per scope.  A waveform strip's table is a histogram per column, so the waveform kernel's flush can
 * add the columns up: the K3 counts of the same plane come out of the K2 pass without a second read and without a
 * single extra LDS atomic per pixel.  Results are exactly those of scope_waveform + scope_histogram_counts.  The pass is
 * shared when both masks select the same plane (both RGB, or both Y'CbCr) and the histogram counts no channel the
 * waveform does not (e.g. 0x07 + 0x07, 0x70 + 0x20); otherwise the two scopes run one after the other.  wvs_out and
 * his_counts are both host or both device.  Inside scope_surface_begin/end a plain scope_waveform call does the same
 * on the side and a following scope_histogram* call on that surface is served from it (counted as a reuse). */
int scope_waveform_histogram(scope_ctx *ctx, const struct scope_surface_data *sd, uint32_t wvs_components,
			     uint32_t his_components, uint32_t flags, uint8_t *wvs_out, uint32_t *his_counts);
/* ---- K1 + K2 + K3 of the dock's surface ------------------------------------------------------
 * All three scopes of every frame of a batch of RGB planes (b->planes_are_rgb; the vectorscope goes through the fused K0
 * with b->colorspace): vss_out = uint8[n][256][256], wvs_out = uint8[n][256][width][4], his_counts = uint32[n][256][4],
 * exactly the tables of scope_vectorscope_batch (SCOPE_FUSE_K0 semantics), scope_waveform_batch and
 * scope_histogram_counts_batch.  For device-resident planes and tables (SCOPE_IN_DEVICE | SCOPE_OUT_DEVICE), RGB component
 * masks (the histogram's a subset of the waveform's), width % 4 == 0 and height <= 65535 the planes are read ONCE, whatever
 * the number of frames (scope_trio_kernel: the vectorscope's table and a waveform table of one strip of columns both live
 * in the LDS, the histogram rides on the waveform's read-out).  Any other combination runs the vectorscope's pass and then
 * the waveform's with the histogram on it.  All three outputs are host or all device. */
int scope_three_scopes_batch(scope_ctx *ctx, const struct scope_batch *b, uint32_t wvs_components, uint32_t his_components,
			     uint32_t flags, uint8_t *vss_out, uint8_t *wvs_out, uint32_t *his_counts);
int scope_waveform_histogram_batch(scope_ctx *ctx, const struct scope_batch *b, uint32_t wvs_components,
				   uint32_t his_components, uint32_t flags, uint8_t *wvs_out, uint32_t *his_counts);

/* ---- K3 + K3b: the whole of his_draw_histogram ---------------------------------------------
 * Replaces his_draw_histogram(src, tex_buf, hi_max, surface_data) (src/histogram.c:357-418):
 * counts on the GPU, then level + float/log conversion on the host with the C library's logf,
 * as the reference does.  tex_buf = 4096 host bytes (float[256][4] on return), hi_max = uint32[3].
 * SCOPE_OUT_DEVICE is not accepted here. */
int scope_histogram(scope_ctx *ctx, const struct scope_surface_data *sd, uint32_t components, int logscale,
		    int level_fixed_value, int level_ratio_value, uint32_t flags, uint8_t *tex_buf, uint32_t *hi_max);

/* K3b alone on the host: his_calculate_max / his_fix_max_level / float conversion
 * (src/histogram.c:330-355,397-417), in place on a counted table. */
void scope_histogram_level(uint8_t *tex_buf, uint32_t *hi_max, uint32_t width, uint32_t height, uint32_t components,
			   int logscale, int level_fixed_value, int level_ratio_value);

/* ---- batched forms (one launch for up to 64 frames, any n_frames accepted) ------------------
 * out tables are contiguous: uint8[n][65536], uint8[n][256][width][4], uint32[n][1024]. */
int scope_vectorscope_batch(scope_ctx *ctx, const struct scope_batch *b, uint32_t flags, uint8_t *out);
int scope_waveform_batch(scope_ctx *ctx, const struct scope_batch *b, uint32_t components, uint32_t flags,
			 uint8_t *out);
int scope_histogram_counts_batch(scope_ctx *ctx, const struct scope_batch *b, uint32_t components, uint32_t flags,
				 uint32_t *counts);

/* ---- D1-D3: display post-passes (tables -> scope picture) -------------------------------------
 * The reference's display pixel shaders as output-sized GPU passes; out = RGBA8 bytes R,G,B,A,
 * row-major.  `display`: 0 overlay, 1 stack, 2 parade (DISP_*, src/waveform.c:22-24).  Tables and
 * pictures are both host or both device (SCOPE_IN_DEVICE | SCOPE_OUT_DEVICE together).
 * The reference leaves the final FLOAT->UNORM8 rounding to the GPU OBS runs on; here it is pinned
 * to floor(clamp01(x)*255+0.5), like K0. */

/* data/vectorscope.effect:27-33 + src/vectorscope.c:414-439.  color_type: 0 white, 1 UV
 * colouring (BT.601 constants when colorspace == 1).  rgba = 256*256*4 bytes. */
int scope_display_vectorscope(scope_ctx *ctx, const uint8_t *table, int intensity, int color_type, int colorspace,
			      uint32_t flags, uint8_t *rgba);
/* The same under the scope's zoom (mouse wheel, src/vectorscope.c:476-482): for zoom > 1.01 the picture is the
 * sprite scaled about its centre 127.5 (:395-408), point-sampled (data/vectorscope.effect:8-12). */
int scope_display_vectorscope_zoom(scope_ctx *ctx, const uint8_t *table, int intensity, int color_type, int colorspace,
				   float zoom, uint32_t flags, uint8_t *rgba);
/* data/waveform.effect:30-83 + src/waveform.c:334-360.  table = uint8[256][width][4];
 * rgba = out_w*out_h*4 bytes with the sizes scope_display_size(width, 256, ...) returns. */
int scope_display_waveform(scope_ctx *ctx, const uint8_t *table, uint32_t width, uint32_t components, int display,
			   int intensity, uint32_t flags, uint8_t *rgba);
/* data/histogram.effect:30-85 + src/histogram.c:522-548.  tex = float[256][4] (after K3b), hi_max = uint32[3];
 * sizes: scope_display_size(256, level_height, ...). */
int scope_display_histogram(scope_ctx *ctx, const float *tex, const uint32_t *hi_max, uint32_t components, int display,
			    uint32_t level_height, uint32_t flags, uint8_t *rgba);
/* get_width/get_height of the sources: width x n for parade, height x n for stack
 * (src/waveform.c:181-199, src/histogram.c:304-322). */
void scope_display_size(uint32_t base_w, uint32_t base_h, uint32_t components, int display, uint32_t *out_w,
			uint32_t *out_h);

/* ---- graticule over a finished picture -----------------------------------------------------------
 * Replaces the two graticule draws of the scopes' video_render: the textured label quads of the vectorscope
 * (DrawGraticule, data/vectorscope.effect:35-39 + src/vectorscope.c:447-454) and the GS_LINES lists drawn with OBS's
 * "Solid" effect (src/vectorscope.c:456-462, src/waveform.c:308-332, src/histogram.c:492-520).  The geometry -- where
 * the brackets, the skin-tone line and the scale lines go -- is the host's (obs-color-monitor_amd/host/
 * scope_graticule.h restates the reference's vertex buffers); this call rasterises it into the RGBA8 picture a
 * scope_display_* call produced: labels first, then the line segments (vertices 2k, 2k+1), each blended in list order
 * with OBS's default source-alpha blend.  argb = what the reference hands gs_effect_set_color (amber 0x80FFBF00, green
 * 0x8000FF00, src/vectorscope.c:141-144, src/waveform.c:311).
 * PINNED, because the reference leaves it to the GPU OBS runs on: the rasterisation rule (pixel centres, major-axis
 * sampling, half-open ends: csrc/scope_display.hip) and the label glyphs (this build's own 5x7 letters "R", "B", "Cy",
 * "Yl", "G", "Mg", one 16x16 cell each, in the order of scope_graticule_vectorscope's labels; the reference's
 * vectorscope-graticule.png is not copied).  The picture is a host or a device buffer (SCOPE_IN_DEVICE |
 * SCOPE_OUT_DEVICE together); the lists are host memory; the call is synchronous. */
struct scope_graticule_vertex {
	float x, y;
};
struct scope_graticule_label {
	float x, y, w, h;     /* quad in picture pixels */
	float u0, v0, u1, v1; /* its range in the label strip, 0..1 */
};
#define SCOPE_LABEL_STRIP_W 96
#define SCOPE_LABEL_STRIP_H 16
int scope_display_graticule(scope_ctx *ctx, uint8_t *rgba, uint32_t width, uint32_t height,
			    const struct scope_graticule_label *labels, uint32_t n_labels,
			    const struct scope_graticule_vertex *lines, uint32_t n_line_vertices, uint32_t argb, uint32_t flags);
/* The glyph texels (0 or 1) the labels are cut from: uint8[SCOPE_LABEL_STRIP_H][SCOPE_LABEL_STRIP_W]. */
void scope_graticule_label_strip(uint8_t *alpha);

/* ---- frame pipeline: the staging ring moved into HBM ----------------------------------------
 * Replaces the reference's 3-slot staging ring + pipeline thread (CM_SURFACE_QUEUE_SIZE,
 * src/common.h:46; cm_render_target / cm_pipeline_thread, src/common.c:223-403) for a host that
 * produces RGB frames at video rate: per frame one host->HBM copy of the RGB plane on an upload
 * stream, then K0-fused vectorscope + waveform + histogram on the compute stream, then the three
 * small tables back on a download stream -- three hipStreams overlapped across `depth` frames in
 * flight.  Source planes in pinned memory (hipHostMalloc / hipHostRegister) make the upload truly
 * asynchronous; they must stay valid until the frame is collected. */
typedef struct scope_pipe scope_pipe;

struct scope_pipe_config {
	uint32_t width, height;  /* of the submitted (full-size) frames */
	uint32_t depth;          /* frames in flight, 1..16; the reference uses 3 */
	int colorspace;          /* for the fused K0 */
	uint32_t wvs_components; /* 0 = no waveform */
	uint32_t his_components; /* 0 = no histogram */
	int want_vectorscope;
	/* target_scale and ROI applied on the GPU after the upload (scope_stage_region semantics).
	 * region.scale == 0 means 1; the scopes then see the staged size (scope_pipe_size()). */
	struct scope_region region;
};

int scope_pipe_create(scope_ctx *ctx, const struct scope_pipe_config *cfg, scope_pipe **out);
/* Width and height of the plane the scopes reduce (= the waveform table's width). */
int scope_pipe_size(const scope_pipe *pipe, uint32_t *width, uint32_t *height);
void scope_pipe_destroy(scope_pipe *pipe);
/* Enqueue one RGB frame (bytes B,G,R,A).  SCOPE_EAGAIN when `depth` frames are already in flight. */
int scope_pipe_submit(scope_pipe *pipe, const uint8_t *rgb_host, uint32_t linesize);
/* Wait for the OLDEST frame in flight and hand out its tables: pinned host memory owned by the pipe, which rotates
 * depth + 1 table buffers, so the pointers stay valid until the NEXT scope_pipe_collect, however many frames are
 * submitted in between -- also when the ring was full and the next submit re-uses the collected frame's slot at once.  Pointers for scopes that were not
 * configured come back NULL.  his_counts is uint32[256][4] (apply scope_histogram_level for K3b).
 * SCOPE_EAGAIN when nothing is in flight.
 * A scope_pipe_submit that fails half-way drains what it had queued and does NOT take a ring slot: the frame is simply
 * not in the pipe, the next submit starts clean. */
int scope_pipe_collect(scope_pipe *pipe, const uint8_t **vectorscope, const uint8_t **waveform,
		       const uint32_t **his_counts);

/* ---- multi-GPU glue ------------------------------------------------------------------------
 * A row-tiled frame gives each GPU a clamped u8 partial table; saturating add of clamped
 * partials is exact, so ranks widen to u32, all-reduce(sum) over RCCL, and clamp once.
 * Both work on device pointers and enqueue on the context's stream. */
int scope_table_widen_u8_u32(scope_ctx *ctx, const uint8_t *in, uint32_t *out, uint64_t n);
int scope_table_clamp_u32_u8(scope_ctx *ctx, const uint32_t *in, uint8_t *out, uint64_t n);
/* Half the wire bytes: two bins per 32-bit word (lo | hi << 16); exact for up to 257 ranks
 * (257 * 255 < 65536, so the halves never carry).  n_pairs = number of bins / 2. */
int scope_table_pack_u8_u16x2(scope_ctx *ctx, const uint8_t *in, uint32_t *out, uint64_t n_pairs);
int scope_table_unpack_clamp_u16x2_u8(scope_ctx *ctx, const uint32_t *in, uint8_t *out, uint64_t n_pairs);

/* The sum itself, callable from C: an in-place ncclAllReduce(sum, uint32) over RCCL / xGMI of `n_words` packed
 * (scope_table_pack_u8_u16x2, or SCOPE_OUT_PACKED16 straight from scope_vectorscope_batch) or plain uint32
 * (histogram counts) words on the context's stream, i.e. ordered behind the kernels that filled `words` and before
 * scope_table_unpack_clamp_u16x2_u8.  The communicator is the host's own ncclComm_t handed in as is (one rank per
 * GPU, created on the context's device); hosts that have none can make one with the three helpers below: rank 0
 * calls scope_rccl_unique_id and passes the 128 bytes to the other ranks by its own means (a file, a socket, MPI,
 * torch.distributed), then every rank calls scope_rccl_comm_create.  RCCL is loaded on first use (dlopen): a
 * single-GPU host never touches it.  The reference has no multi-GPU path; this replaces nothing in it. */
#define SCOPE_RCCL_ID_BYTES 128
int scope_rccl_unique_id(uint8_t id[SCOPE_RCCL_ID_BYTES]);
int scope_rccl_comm_create(scope_ctx *ctx, const uint8_t id[SCOPE_RCCL_ID_BYTES], int n_ranks, int rank, void **rccl_comm);
int scope_rccl_comm_destroy(void *rccl_comm);
int scope_table_allreduce(scope_ctx *ctx, void *rccl_comm, uint32_t *words, uint64_t n_words);
/* What the collective library itself says about a communicator: ncclCommCount, ncclCommUserRank and ncclGetVersion
 * (major*10000 + minor*100 + patch).  Any of the three pointers may be NULL; a value the library does not offer comes
 * back as -1; rccl_comm may be NULL when only the version is asked for.  A host (bench.py) prints these next to its
 * numbers so that a line says how many ranks RCCL saw, not how many were asked for.
 * The environment variable SCOPE_RCCL_LIBRARY (a path or soname) names the library to bind instead of the ROCm librccl. */
int scope_rccl_comm_info(void *rccl_comm, int *n_ranks, int *rank, int *version);

/* ---- frame pools ----------------------------------------------------------------------------
 * Where a long-lived pool of frames lies in HBM decides a few per cent of every streaming kernel's rate on this chip
 * (pools read at 0.79 ... 0.86 of peak on some boxes, alike on others).  A host that keeps frames
 * resident -- a ring of staged surfaces, a batch it reduces again and again -- can let the library look before it settles:
 * scope_frame_pool_alloc makes up to `candidates` device allocations of `bytes` (all held at the same time: candidates x
 * bytes of HBM for a few milliseconds), reads each with the vectorscope accumulation kernel itself in three interleaved
 * rounds (the bytes taken as frames; timed by the kernel's own begin / end stamps; the median round decides), keeps the
 * fastest and frees the rest.
 * candidates <= 1 is a plain allocation.  read_gbps (optional, room for `candidates` floats) receives the measured rates,
 * the kept allocation's first; *tried (optional) how many were made (fewer than asked for when memory ran out).  The
 * pool is ordinary device memory (use it with SCOPE_IN_DEVICE); give it back with scope_frame_pool_free.
 * The reference has no counterpart (its frames live in libobs textures); this replaces nothing in it. */
#define SCOPE_POOL_MAX_CANDIDATES 16
int scope_frame_pool_alloc(scope_ctx *ctx, uint64_t bytes, uint32_t candidates, void **pool, float *read_gbps,
			   uint32_t *tried);
int scope_frame_pool_free(scope_ctx *ctx, void *pool);

/* ---- measurement ---------------------------------------------------------------------------
 * With profiling on, every accumulation launch (the dominant kernel of a call, not the small
 * partial-table reduction) is bracketed by HIP events on the context's stream.
 * scope_ctx_kernel_ms() waits for them and returns how many launches were bracketed since
 * profiling was switched on (the most recent 512 at most) and their summed duration. */
int scope_ctx_profile(scope_ctx *ctx, int enable);
int scope_ctx_kernel_ms(scope_ctx *ctx, uint32_t *launches, float *total_ms);

#ifdef __cplusplus
}
#endif
#endif
