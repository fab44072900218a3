/* Graticule geometry: see scope_graticule.h.  Float arithmetic is written in the reference's order so the
 * vertices match it bit for bit (they are plain IEEE single operations, no transcendental except hypotf). */
#include "scope_graticule.h"

#include <math.h>
#include <stddef.h>

/* Colour targets (U, V) in the 256x256 vectorscope plane, the values the reference uses (it credits FFmpeg's
 * vectorscope filter, src/vectorscope.c:283-316): the first six carry a label (in the order of the label
 * strip's glyphs) and a bracket, the other six only a bracket. */
static const uint8_t vss_targets[2][12][2] = {
	{{90, 240}, {240, 110}, {166, 16}, {16, 146}, {54, 34}, {202, 222},
	 {44, 142}, {156, 44}, {72, 58}, {184, 198}, {100, 212}, {212, 114}}, /* BT.601 */
	{{102, 240}, {240, 118}, {154, 16}, {16, 138}, {42, 26}, {214, 230},
	 {212, 120}, {109, 212}, {193, 204}, {63, 52}, {147, 44}, {44, 136}}, /* BT.709 */
};

/* Integer RGB -> U, V of the skin tone colour, C division (truncation toward zero): src/vectorscope.c:28-34. */
static int skin_u(int cs, int r, int g, int b)
{
	return cs == 1 ? (-150 * r - 296 * g + 448 * b) / 1024 + 128 : (-102 * r - 346 * g + 450 * b) / 1024 + 128;
}
static int skin_v(int cs, int r, int g, int b)
{
	return cs == 1 ? (+448 * r - 374 * g - 72 * b) / 1024 + 128 : (+450 * r - 408 * g - 40 * b) / 1024 + 128;
}

static struct scope_vertex vtx(float x, float y)
{
	const struct scope_vertex v = {x, y};
	return v;
}

uint32_t scope_graticule_vectorscope(int colorspace, uint32_t graticule, uint32_t skintone_color,
				     struct scope_label *labels, struct scope_vertex *lines)
{
	const int cs = colorspace == 1 ? 1 : 2;
	const uint8_t(*pp)[2] = vss_targets[cs - 1];

	if (labels) {
		for (int i = 0; i < (int)SCOPE_VSS_LABELS; i++) { /* :320-333 */
			float x = (float)pp[i][0];
			float y = 256.f - (float)pp[i][1];
			/* push the glyph away from its bracket, towards the outside of the plane */
			if (x < 72)
				y += 20;
			else if (x > 184)
				y -= 20;
			else if (y > 128)
				x += 20;
			else
				x -= 20;
			labels[i].x = x - 8;
			labels[i].y = y - 8;
			labels[i].w = labels[i].h = 16;
			labels[i].u0 = (float)i / 6.f;
			labels[i].v0 = 0.f;
			labels[i].u1 = (float)(i + 1) / 6.f;
			labels[i].v1 = 1.f;
		}
	}
	if (!lines)
		return 0;

	uint32_t n = 0;
	for (int i = 0; i < 12; i++) { /* :339-350: four corner brackets, arms of 4 px on a 12 px box */
		const float x = (float)pp[i][0];
		const float y = 256.f - (float)pp[i][1];
		for (int sy = -1; sy <= 1; sy += 2) {
			for (int sx = -1; sx <= 1; sx += 2) {
				const float cx = (float)(6 * sx), cy = (float)(6 * sy);
				lines[n++] = vtx(x + cx, y + cy);
				lines[n++] = vtx(x + (float)(2 * sx), y + cy);
				lines[n++] = vtx(x + cx, y + cy);
				lines[n++] = vtx(x + cx, y + (float)(2 * sy));
			}
		}
	}

	/* skin tone line, :352-377 */
	const int b = (int)(skintone_color >> 16 & 0xFF), g = (int)(skintone_color >> 8 & 0xFF), r = (int)(skintone_color & 0xFF);
	float u = (float)skin_u(cs, r, g, b), v = (float)skin_v(cs, r, g, b);
	const float norm = hypotf(u - 128.0f, v - 128.0f);
	if (norm > 1.0f) {
		u = (u - 128.0f) * 128.f / norm + 128.0f;
		v = (v - 128.0f) * 128.f / norm + 128.0f;
		if (graticule & SCOPE_GRATICULE_IQ) {
			lines[n++] = vtx(255.f - u, v);
			lines[n++] = vtx(u, 255.f - v);
			lines[n++] = vtx(v, u);
			lines[n++] = vtx(255.f - v, 255.f - u);
		} else {
			lines[n++] = vtx(127.5f, 127.5f);
			lines[n++] = vtx(u, 255.f - v);
		}
	}
	return n;
}

static void emit(struct scope_vertex *lines, uint32_t max, uint32_t *n, float x, float y)
{
	if (lines && *n < max)
		lines[*n] = vtx(x, y);
	(*n)++;
}

uint32_t scope_graticule_waveform(int graticule_lines, int display, uint32_t n_components, uint32_t tex_width,
				  struct scope_vertex *lines, uint32_t max)
{
	uint32_t n = 0;
	if (graticule_lines <= 0)
		return 0;
	const int stack = display == 1, parade = display == 2;
	const int n_stack = stack ? (int)n_components : 1;
	for (int i = 0; i < n_stack; i++) {
		const float yoff = stack ? 256 * i + 0.5f : 0.0f; /* src/waveform.c:316 */
		const float xcoe = (float)(tex_width * (parade ? n_components : 1u));
		/* a band below the first starts at vertex 2: its top line is the previous band's bottom line (:327) */
		for (int k = (stack && i) ? 1 : 0; k <= graticule_lines; k++) {
			const float y = 256.0f * k / graticule_lines; /* :299-300 */
			emit(lines, max, &n, 0.0f * xcoe, y + yoff);
			emit(lines, max, &n, 1.0f * xcoe, y + yoff);
		}
	}
	return n;
}

uint32_t scope_graticule_histogram(int vertical_lines, float horizontal_step, int logscale, int level_fixed_value,
				   int level_ratio_value, int display, uint32_t n_components, uint32_t level_height,
				   struct scope_vertex *lines, uint32_t max)
{
	float y_max = 0; /* src/histogram.c:454-461 */
	if (logscale)
		y_max = 0;
	else if (level_fixed_value)
		y_max = (float)level_fixed_value;
	else if (level_ratio_value)
		y_max = level_ratio_value / 10.f;
	const float y_step = y_max > 0 ? horizontal_step / y_max : 0.0f;
	const int has_v = vertical_lines > 0;
	const int has_h = y_step > 1.0f / 64; /* GRATICULE_H_MAX, :36,:464 */
	uint32_t n = 0;
	if (!has_v && !has_h)
		return 0;
	const int stack = display == 1, parade = display == 2;
	const int n_parade = parade ? (int)n_components : 1, n_stack = stack ? (int)n_components : 1;
	for (int j = 0; j < n_stack; j++) {
		for (int i = 0; i < n_parade; i++) {
			const float ycoe = (float)(level_height * 1); /* :503-506 */
			const float yoff = (float)(level_height * (uint32_t)j);
			const float xoff = parade ? 256 * i + 0.0f : 1.0f;
			uint32_t skip = (parade && i) ? 2 : 0; /* :516: the left edge is the previous band's right edge */
			if (has_v) {
				for (int k = 0; k <= vertical_lines; k++) {
					const float x = 256.0f * k / vertical_lines; /* :476-478 */
					if (skip) {
						skip -= 2;
						continue;
					}
					emit(lines, max, &n, x + xoff, 0.0f * ycoe + yoff);
					emit(lines, max, &n, x + xoff, 1.0f * ycoe + yoff);
				}
			}
			if (has_h) {
				for (float y = 1.0f; y >= 0.0f; y -= y_step) { /* :482-485 */
					if (skip) {
						skip -= 2;
						continue;
					}
					emit(lines, max, &n, 0.0f + xoff, y * ycoe + yoff);
					emit(lines, max, &n, 256.0f + xoff, y * ycoe + yoff);
				}
			}
		}
	}
	return n;
}
