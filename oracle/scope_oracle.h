/*
 * scope_oracle.h -- CPU restatement of obs-color-monitor's scope-accumulation path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (obs-color-monitor_amd/,
 * include/) may include, link or call this.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg use it, and only as the checker / reported baseline.
 *
 * PARITY STATUS: "parity unpinned".  The reference ships no golden vectors or
 * known-answer tests for this path, and its sources cannot be
 * built in this image without stand-ins for libobs headers (obs-module.h is absent),
 * so no reference-produced outputs exist to pin this restatement against.  It is
 * written from the reference's semantics, function by function, with each function
 * citing the reference file:line it follows, and it is cross-checked against an
 * independently written numpy restatement (oracle/scope_oracle.py) and against
 * hand-derived known answers (tests/test_oracle.py).  The only expected values the
 * reference itself carries for this path -- the vectorscope graticule's colour targets
 * (src/vectorscope.c:283-316) and its integer RGB2U/V macros (src/vectorscope.c:28-34) --
 * are tolerances, not vectors (a +-6 box; 10-bit coefficients): K0 is held against them
 * (within 2 of every target, within 3 / 2 of the macros over all 2^24 colours), which pins
 * the matrix, the axes and the scale, not the last bit.
 *
 * All citations are relative to /root/reference (obs-color-monitor v0.9.5).
 */
#ifndef SCOPE_ORACLE_H
#define SCOPE_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Frame descriptor handed to every reduction.  Field-for-field the meaning of
 * struct cm_surface_data (src/common.h:24-30) minus the unused `tex`. */
struct oracle_surface {
	const uint8_t *rgb_data; /* BGRA8 rows: bytes B,G,R,A  (may be NULL) */
	const uint8_t *yuv_data; /* BGRA8 rows: bytes U,Y,V,255 (may be NULL) */
	uint32_t linesize;       /* row pitch in bytes, >= 4*width */
	uint32_t width, height;
	int colorspace;          /* 1 = BT.601, anything else = BT.709 (src/common.c:207-208) */
};

/* K0: RGB -> Y'CbCr, data/common.effect:23-43 driven by src/common.c:202-213.
 * fp32, terms summed left to right as written, no FMA contraction; the
 * FLOAT->UNORM8 store (left to the GPU by the reference) is pinned here to
 * floor(clamp(x,0,1)*255 + 0.5).  In: bytes B,G,R,A.  Out: bytes U,Y,V,255. */
void oracle_rgb_to_yuv(const uint8_t *rgb, uint32_t rgb_linesize, uint8_t *yuv, uint32_t yuv_linesize,
		       uint32_t width, uint32_t height, int colorspace);

/* K1: vss_draw_vectorscope, src/vectorscope.c:217-238.  out = uint8[256][256]. */
void oracle_vectorscope(uint8_t *out, const struct oracle_surface *sd);

/* K2: wvs_draw_waveform, src/waveform.c:220-257 (+ inc_uint8 :201-205).
 * out = uint8[256][width][4]. */
void oracle_waveform(uint8_t *out, const struct oracle_surface *sd, uint32_t components);

/* K3: counting part of his_draw_histogram, src/histogram.c:357-395.
 * counts = uint32[256][4], slots R,G,B,unused.  Returns 0 when no plane was
 * selectable (the reference returns early there, histogram.c:372-373), else 1. */
int oracle_histogram_counts(uint32_t *counts, const struct oracle_surface *sd, uint32_t components);

/* K3 + K3b: the whole of his_draw_histogram, src/histogram.c:357-418 with
 * his_calculate_max / his_fix_max_level (:330-355).  tex_buf = 4096 bytes that
 * end up holding float[256][4]; hi_max = uint32[3]. */
void oracle_histogram(uint8_t *tex_buf, uint32_t *hi_max, const struct oracle_surface *sd, uint32_t components,
		      int logscale, int level_fixed_value, int level_ratio_value);

/* K3b alone, applied in place to an already counted table (used to check the
 * host-side post-pass of the product, which receives exact counts from the GPU). */
void oracle_histogram_level(uint8_t *tex_buf, uint32_t *hi_max, uint32_t width, uint32_t height, uint32_t components,
			    int logscale, int level_fixed_value, int level_ratio_value);

/* D1-D3: the display pixel shaders that turn the tables into the scope picture.
 * fp32, operations in the shaders' order, texel centres at (x+0.5)/size, point sampling, and the
 * FLOAT->UNORM8 store pinned like K0 (the reference leaves it to the GPU: parity unpinned).
 * Output: RGBA8 bytes R,G,B,A, row-major.  display: 0 overlay, 1 stack, 2 parade. */

/* data/vectorscope.effect:27-33 with the uniforms of src/vectorscope.c:414-439.
 * color_type 0 = white, 1 = UV colouring (601 constants when colorspace == 1).  out = 256*256*4 bytes. */
void oracle_display_vectorscope(uint8_t *out, const uint8_t *table, int intensity, int color_type, int colorspace);
/* The same under the scope's zoom (src/vectorscope.c:395-408: the sprite is drawn scaled by `zoom` about the centre
 * 127.5 when zoom > 1.01; mouse wheel, :476-482).  Point sampling (data/vectorscope.effect:8-12); the interpolated
 * texture coordinate is pinned to ((p + 0.5) - ofst) / zoom / 256 in fp32. */
void oracle_display_vectorscope_zoom(uint8_t *out, const uint8_t *table, int intensity, int color_type, int colorspace,
				     float zoom);

/* data/waveform.effect:30-83 with the technique choice of src/waveform.c:334-360.
 * table = uint8[256][width][4].  Output size via oracle_display_size(). */
void oracle_display_waveform(uint8_t *out, const uint8_t *table, uint32_t width, uint32_t components, int display,
			     int intensity);

/* data/histogram.effect:30-85 with src/histogram.c:522-548.  tex = float[256][4], hi_max = uint32[3]. */
void oracle_display_histogram(uint8_t *out, const float *tex, const uint32_t *hi_max, uint32_t components, int display,
			      uint32_t level_height);

/* Output width/height of D2 (base_w = frame width, base_h = 256) and D3 (base_w = 256, base_h = level_height). */
void oracle_display_size(uint32_t base_w, uint32_t base_h, uint32_t components, int display, uint32_t *out_w,
			 uint32_t *out_h);

/* "Courtesy" CPU baselines: the same tables from `threads` POSIX threads, each reducing a tile of rows
 * into private u32 counters that are summed (and, for the u8 tables, clamped at 255) at the end -- what a
 * multi-core host could do at best; the reference itself runs one thread.  Results equal the single-thread
 * functions bit for bit.  `reps` (>= 1) is a timing aid: every thread reduces its tile that many times, restarting
 * its counters each time, so thread start-up is paid once per call.  Return 0, or -1 when threads == 0 or
 * memory runs out. */
int oracle_vectorscope_mt(uint8_t *out, const struct oracle_surface *sd, uint32_t threads, uint32_t reps);
int oracle_histogram_counts_mt(uint32_t *counts, const struct oracle_surface *sd, uint32_t components, uint32_t threads,
			       uint32_t reps);
int oracle_waveform_mt(uint8_t *out, const struct oracle_surface *sd, uint32_t components, uint32_t threads, uint32_t reps);

/* Capture-side pre-steps of cm_render_target: the target is rendered at 1/scale of its size
 * (src/common.c:249-250, :141-168) and, for an ROI source with a valid rectangle, only the sub-rectangle
 * (x, y, cx, cy) -- in SCALED coordinates -- is drawn into the staged surface (:272-282, :202-206).
 * The reference leaves the down-scale to the target source's own sampler; this restatement pins it to
 * point sampling at the centre of each scale x scale cell:
 *   dst(i, j) = src((x + i) * scale + scale / 2, (y + j) * scale + scale / 2).
 * Returns 0, or -1 when scale == 0 or the rectangle is empty / outside the scaled frame. */
int oracle_region_size(uint32_t width, uint32_t height, uint32_t scale, uint32_t *scaled_w, uint32_t *scaled_h);
int oracle_stage_region(const uint8_t *src, uint32_t src_linesize, uint32_t width, uint32_t height, uint32_t scale,
			uint32_t x, uint32_t y, uint32_t cx, uint32_t cy, uint8_t *dst, uint32_t dst_linesize);

/* Deterministic frame generator shared by tests and bench (xorshift64).
 * dist: 1 uniform, 2 concentrated (128 + U{-8..7}), 3 flat (R,G,B)=(0xC6,0xCC,0xD7),
 *       4 eight vertical colour bars, 5 uniform with alpha==0 on a 16x16 checker.
 * Padding bytes between 4*width and linesize are filled with 0xA5. */
void oracle_fill_frame(uint8_t *dst, uint32_t linesize, uint32_t width, uint32_t height, int dist, uint64_t seed);

/* FNV-1a 64 over a byte buffer (fixture hashes for large tables). */
uint64_t oracle_fnv1a64(const void *data, uint64_t len);

#ifdef __cplusplus
}
#endif
#endif
