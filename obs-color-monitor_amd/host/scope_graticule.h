/* Graticule geometry of the three scopes: pure host arithmetic, no GPU.
 *
 * The reference builds vertex buffers and lets libobs draw them with transforms (gs_matrix_mul) and a
 * start-vertex offset; these functions return the vertices AFTER those transforms, i.e. the line list a
 * host would draw over the picture scope_display_* produced, in the picture's own pixel coordinates.
 * Line lists are GS_LINES style: vertices 2k, 2k+1 form one segment. */
#ifndef SCOPE_GRATICULE_H
#define SCOPE_GRATICULE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

struct scope_vertex {
	float x, y;
};

/* One textured label quad: position/size in the 256x256 vectorscope picture, texture range in the label
 * strip (six glyphs side by side). */
struct scope_label {
	float x, y, w, h;
	float u0, v0, u1, v1;
};

#define SCOPE_GRATICULE_IQ 256u        /* src/vectorscope.c:24: skin tone line drawn as the I/Q cross */
#define SCOPE_GRATICULE_COLOR_MASK 3u  /* src/vectorscope.c:25: 1 amber, 2 green (picked by the caller) */
#define SCOPE_SKIN_TONE_LINE 0x0054FFu /* src/vectorscope.c:26: default skin tone colour, 0xBBGGRR */
#define SCOPE_VSS_LABELS 6u
#define SCOPE_VSS_LINE_VERTICES_MAX (12u * 16u + 4u)

/* Vectorscope, create_graticule_vbuf (src/vectorscope.c:267-380).  colorspace 1 = BT.601, else BT.709.
 * labels: exactly SCOPE_VSS_LABELS quads.  lines: corner brackets round the twelve colour targets, then
 * the skin tone line (two vertices, or four with SCOPE_GRATICULE_IQ; none when the colour is neutral).
 * Returns the number of line vertices written (<= SCOPE_VSS_LINE_VERTICES_MAX). */
uint32_t scope_graticule_vectorscope(int colorspace, uint32_t graticule, uint32_t skintone_color,
				     struct scope_label *labels, struct scope_vertex *lines);

/* Waveform, create_graticule_vbuf + wvs_render_graticule (src/waveform.c:291-332): graticule_lines + 1
 * horizontal lines per band; display 0 overlay, 1 stack, 2 parade; tex_width = the waveform table's width.
 * Returns the number of vertices the draw would emit; writes at most max of them. */
uint32_t scope_graticule_waveform(int graticule_lines, int display, uint32_t n_components, uint32_t tex_width,
				  struct scope_vertex *lines, uint32_t max);

/* Histogram, create_graticule_vbuf + his_render_graticule (src/histogram.c:452-520).  level_* and logscale as
 * in scope_histogram (they decide whether horizontal lines exist), horizontal_step in the unit of the
 * active level setting, level_height = height of one band in pixels. */
uint32_t scope_graticule_histogram(int vertical_lines, float horizontal_step, int logscale, int level_fixed_value,
				   int level_ratio_value, int display, uint32_t n_components, uint32_t level_height,
				   struct scope_vertex *lines, uint32_t max);

#ifdef __cplusplus
}
#endif
#endif
