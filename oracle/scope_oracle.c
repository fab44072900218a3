/*
 * scope_oracle.c -- CPU restatement of obs-color-monitor's scope-accumulation path.
 *
 * TEST INFRASTRUCTURE ONLY (see scope_oracle.h).  PARITY STATUS: "parity unpinned".
 *
 * Build with -ffp-contract=off: K0 must not be FMA-contracted.
 * Citations are relative to /root/reference.
 */
#include "scope_oracle.h"

#include <math.h>
#include <string.h>

/* ---------------------------------------------------------------- K0 ---- */

struct k0_matrix {
	float u[3], y[3], v[3]; /* coefficients of r, g, b */
};

/* data/common.effect:27-29 (601) and :38-40 (709) */
static const struct k0_matrix k0_bt601 = {
	{-0.147643f, -0.289855f, +0.437500f},
	{+0.299000f, +0.587000f, +0.114000f},
	{+0.437500f, -0.366351f, -0.071147f},
};
static const struct k0_matrix k0_bt709 = {
	{-0.100643f, -0.338571f, +0.439216f},
	{+0.212600f, +0.715200f, +0.072200f},
	{+0.439216f, -0.398941f, -0.040273f},
};

/* FLOAT -> UNORM8 store rule pinned by this build (the reference leaves it to the GPU). */
static uint8_t k0_quantise(float x)
{
	if (!(x > 0.0f))
		x = 0.0f;
	else if (x > 1.0f)
		x = 1.0f;
	return (uint8_t)floorf(x * 255.0f + 0.5f);
}

/* One texel: in B,G,R,A bytes -> out U,Y,V,255 bytes.
 * common.effect samples rgb.x=R, rgb.y=G, rgb.z=B and returns float4(V,Y,U,1) into a
 * BGRA target, so memory byte0 = .z = U, byte1 = .y = Y, byte2 = .x = V, byte3 = 255. */
static void k0_texel(const struct k0_matrix *m, const uint8_t *in, uint8_t *out)
{
	const float r = (float)in[2] / 255.0f;
	const float g = (float)in[1] / 255.0f;
	const float b = (float)in[0] / 255.0f;

	/* a*r - c*g == a*r + (-c)*g exactly in IEEE, so signed coefficients keep the
	 * left-to-right order of the shader expression. */
	float u = m->u[0] * r;
	u = u + m->u[1] * g;
	u = u + m->u[2] * b;
	u = u + 0.5f;
	u = u - 1.0f / 256.0f;

	float y = m->y[0] * r;
	y = y + m->y[1] * g;
	y = y + m->y[2] * b;

	float v = m->v[0] * r;
	v = v + m->v[1] * g;
	v = v + m->v[2] * b;
	v = v + 0.5f;

	out[0] = k0_quantise(u);
	out[1] = k0_quantise(y);
	out[2] = k0_quantise(v);
	out[3] = 255;
}

void oracle_rgb_to_yuv(const uint8_t *rgb, uint32_t rgb_linesize, uint8_t *yuv, uint32_t yuv_linesize,
		       uint32_t width, uint32_t height, int colorspace)
{
	/* src/common.c:207-208: colorspace == 1 picks 601, everything else 709 */
	const struct k0_matrix *m = colorspace == 1 ? &k0_bt601 : &k0_bt709;
	for (uint32_t row = 0; row < height; row++) {
		const uint8_t *in = rgb + (size_t)rgb_linesize * row;
		uint8_t *out = yuv + (size_t)yuv_linesize * row;
		for (uint32_t col = 0; col < width; col++)
			k0_texel(m, in + 4 * col, out + 4 * col);
	}
}

/* ---------------------------------------------------------------- K1 ---- */

static inline void sat_inc_u8(uint8_t *cell)
{
	/* inc_uint8, src/waveform.c:201-205; same test inline at src/vectorscope.c:233-234 */
	const unsigned n = *cell;
	*cell = (uint8_t)(n + (n < 255));
}

void oracle_vectorscope(uint8_t *out, const struct oracle_surface *sd)
{
	memset(out, 0, 256 * 256); /* vectorscope.c:219-220 */
	if (!sd->yuv_data)
		return; /* guard lives in vss_surface_cb, vectorscope.c:252 */
	for (uint32_t row = 0; row < sd->height; row++) {
		const uint8_t *px = sd->yuv_data + (size_t)sd->linesize * row; /* vd_add, :225,236 */
		for (uint32_t col = 0; col < sd->width; col++, px += 4) {
			const unsigned u = px[0], v = px[2];       /* :228-230, alpha not read */
			sat_inc_u8(&out[u + 256u * (255u - v)]); /* :232-234 */
		}
	}
}

/* ---------------------------------------------------------- plane pick -- */

/* waveform.c:228-234 == histogram.c:367-373: any low-nibble bit selects the RGB plane,
 * else any of bits 4-6 selects the YUV plane, else nothing. */
static const uint8_t *pick_plane(const struct oracle_surface *sd, uint32_t components)
{
	if (components & 0x07)
		return sd->rgb_data;
	if (components & 0x70)
		return sd->yuv_data;
	return 0;
}

/* ---------------------------------------------------------------- K2 ---- */

void oracle_waveform(uint8_t *out, const struct oracle_surface *sd, uint32_t components)
{
	const size_t rowbytes = (size_t)sd->width * 4;
	memset(out, 0, rowbytes * 256); /* waveform.c:225-226 */

	const uint8_t *plane = pick_plane(sd, components);
	if (!plane)
		return; /* :233-234 */

	/* :236-238 -- bit c or bit c+4 enables byte c of the texel */
	int on[3];
	for (int c = 0; c < 3; c++)
		on[c] = (components & (0x11u << c)) != 0;

	for (uint32_t row = 0; row < sd->height; row++) {
		const uint8_t *px = plane + (size_t)sd->linesize * row; /* :241 */
		for (uint32_t col = 0; col < sd->width; col++, px += 4) {
			if (px[3] == 0)
				continue; /* :247-248 */
			for (int c = 0; c < 3; c++)
				if (on[c]) /* :249-254 */
					sat_inc_u8(out + (255u - px[c]) * rowbytes + (size_t)col * 4 + c);
		}
	}
}

/* ---------------------------------------------------------------- K3 ---- */

int oracle_histogram_counts(uint32_t *counts, const struct oracle_surface *sd, uint32_t components)
{
	memset(counts, 0, 1024 * sizeof(uint32_t)); /* histogram.c:363-365 */

	const uint8_t *plane = pick_plane(sd, components);
	if (!plane)
		return 0; /* :372-373 */

	const int on_b = (components & 0x11) != 0; /* :375-377 */
	const int on_g = (components & 0x22) != 0;
	const int on_r = (components & 0x44) != 0;

	for (uint32_t row = 0; row < sd->height; row++) {
		const uint8_t *px = plane + (size_t)sd->linesize * row; /* :380 */
		for (uint32_t col = 0; col < sd->width; col++, px += 4) {
			if (px[3] == 0)
				continue; /* :386-387 */
			/* :388-393 -- slot order is R,G,B: byte2 -> slot0, byte1 -> slot1, byte0 -> slot2 */
			if (on_r)
				counts[px[2] * 4 + 0] += 1;
			if (on_g)
				counts[px[1] * 4 + 1] += 1;
			if (on_b)
				counts[px[0] * 4 + 2] += 1;
		}
	}
	return 1;
}

/* --------------------------------------------------------------- K3b ---- */

void oracle_histogram_level(uint8_t *tex_buf, uint32_t *hi_max, uint32_t width, uint32_t height, uint32_t components,
			    int logscale, int level_fixed_value, int level_ratio_value)
{
	uint32_t cnt[1024];
	memcpy(cnt, tex_buf, sizeof cnt);

	/* histogram.c:397-402 */
	if (level_fixed_value > 0 || level_ratio_value > 0) {
		uint32_t lvl = level_fixed_value > 0
				       ? (uint32_t)level_fixed_value
				       : (uint32_t)((uint64_t)width * height * (uint64_t)level_ratio_value / 1000);
		if (lvl == 0)
			lvl = 1; /* his_fix_max_level, :349-355 */
		hi_max[0] = hi_max[1] = hi_max[2] = lvl;
	} else {
		/* his_calculate_max, :330-347; index 0=R (0x44), 1=G (0x22), 2=B (0x11) */
		for (int j = 0; j < 3; j++) {
			uint32_t best = 1;
			if (components & (0x44u >> j))
				for (int i = 0; i < 256; i++)
					if (cnt[i * 4 + j] > best)
						best = cnt[i * 4 + j];
			hi_max[j] = best;
		}
	}

	float flt[1024];
	memcpy(flt, cnt, sizeof flt); /* disabled channels keep their raw bit pattern in logscale */
	if (logscale) {
		/* :405-413 */
		for (int j = 0; j < 3; j++) {
			if (!(components & (0x44u >> j)))
				continue;
			const float s = 1.0f / logf((float)(hi_max[j] + 1));
			for (int i = 0; i < 256; i++) {
				const uint32_t n = cnt[i * 4 + j];
				flt[i * 4 + j] = n ? logf((float)(n + 1)) * s : 0.0f;
			}
			hi_max[j] = 1;
		}
	} else {
		/* :414-417 */
		for (int i = 0; i < 1024; i++)
			flt[i] = (float)cnt[i];
	}
	memcpy(tex_buf, flt, sizeof flt);
}

void oracle_histogram(uint8_t *tex_buf, uint32_t *hi_max, const struct oracle_surface *sd, uint32_t components,
		      int logscale, int level_fixed_value, int level_ratio_value)
{
	uint32_t cnt[1024];
	const int counted = oracle_histogram_counts(cnt, sd, components);
	memcpy(tex_buf, cnt, sizeof cnt);
	if (!counted)
		return; /* histogram.c:372-373 returns before the level pass; hi_max untouched */
	oracle_histogram_level(tex_buf, hi_max, sd->width, sd->height, components, logscale, level_fixed_value,
			       level_ratio_value);
}

/* ------------------------------------------------------------ D1-D3 ---- */

static const float disp_color[3][3] = {
	/* rows of the effects' `color` matrix: data/waveform.effect:4-9, data/histogram.effect:4-9 */
	{1.00f, 0.41f, 0.41f},
	{0.00f, 1.00f, 0.00f},
	{0.53f, 0.53f, 1.00f},
};

/* n_components(): population count of the plane bits, src/waveform.c:172-179, src/histogram.c:295-302 */
static uint32_t disp_n_components(uint32_t components)
{
	uint32_t n = 0;
	for (uint32_t c = components & 0x77u; c; c >>= 1)
		n += c & 1u;
	return n;
}

/* ------------------------------------------- multi-threaded courtesy baselines ---- */

#include <pthread.h>
#include <stdlib.h>

struct mt_job {
	const struct oracle_surface *sd;
	const uint8_t *plane;
	uint32_t components;
	uint32_t tid, threads;
	uint32_t reps;            /* timing aid: the tile is reduced this many times (tables restarted each time) */
	uint32_t *private_counts; /* vectorscope: [65536]; histogram: [1024] */
	uint8_t *out;             /* waveform: the shared table, threads own disjoint columns */
};

static void *mt_vectorscope(void *arg)
{
	struct mt_job *j = arg;
	const struct oracle_surface *sd = j->sd;
	const uint32_t r0 = (uint32_t)((uint64_t)sd->height * j->tid / j->threads);
	const uint32_t r1 = (uint32_t)((uint64_t)sd->height * (j->tid + 1) / j->threads);
	for (uint32_t rep = 0; rep < j->reps; rep++) {
		memset(j->private_counts, 0, 65536 * sizeof(uint32_t));
		for (uint32_t row = r0; row < r1; row++) {
			const uint8_t *px = sd->yuv_data + (size_t)sd->linesize * row;
			for (uint32_t col = 0; col < sd->width; col++, px += 4)
				j->private_counts[px[0] + 256u * (255u - px[2])]++;
		}
	}
	return NULL;
}

static void *mt_histogram(void *arg)
{
	struct mt_job *j = arg;
	const struct oracle_surface *sd = j->sd;
	const int on_b = (j->components & 0x11) != 0, on_g = (j->components & 0x22) != 0, on_r = (j->components & 0x44) != 0;
	const uint32_t r0 = (uint32_t)((uint64_t)sd->height * j->tid / j->threads);
	const uint32_t r1 = (uint32_t)((uint64_t)sd->height * (j->tid + 1) / j->threads);
	for (uint32_t rep = 0; rep < j->reps; rep++) {
		memset(j->private_counts, 0, 1024 * sizeof(uint32_t));
		for (uint32_t row = r0; row < r1; row++) {
			const uint8_t *px = j->plane + (size_t)sd->linesize * row;
			for (uint32_t col = 0; col < sd->width; col++, px += 4) {
				if (px[3] == 0)
					continue;
				if (on_r)
					j->private_counts[px[2] * 4 + 0]++;
				if (on_g)
					j->private_counts[px[1] * 4 + 1]++;
				if (on_b)
					j->private_counts[px[0] * 4 + 2]++;
			}
		}
	}
	return NULL;
}

static void *mt_waveform(void *arg)
{
	/* column tiles: a column's counters depend on that column only, so threads never share a counter */
	struct mt_job *j = arg;
	const struct oracle_surface *sd = j->sd;
	const size_t rowbytes = (size_t)sd->width * 4;
	const uint32_t c0 = (uint32_t)((uint64_t)sd->width * j->tid / j->threads);
	const uint32_t c1 = (uint32_t)((uint64_t)sd->width * (j->tid + 1) / j->threads);
	int on[3];
	for (int c = 0; c < 3; c++)
		on[c] = (j->components & (0x11u << c)) != 0;
	for (uint32_t rep = 0; rep < j->reps; rep++) {
		for (uint32_t v = 0; v < 256 && c1 > c0; v++) /* this thread's columns of the table */
			memset(j->out + v * rowbytes + (size_t)c0 * 4, 0, (size_t)(c1 - c0) * 4);
		for (uint32_t row = 0; row < sd->height; row++) {
			const uint8_t *px = j->plane + (size_t)sd->linesize * row + (size_t)c0 * 4;
			for (uint32_t col = c0; col < c1; col++, px += 4) {
				if (px[3] == 0)
					continue;
				for (int c = 0; c < 3; c++)
					if (on[c])
						sat_inc_u8(j->out + (255u - px[c]) * rowbytes + (size_t)col * 4 + c);
			}
		}
	}
	return NULL;
}

static int mt_run(void *(*fn)(void *), struct mt_job *jobs, uint32_t threads)
{
	pthread_t *th = malloc(sizeof(pthread_t) * threads);
	if (!th)
		return -1;
	uint32_t started = 0;
	int rc = 0;
	for (; started < threads; started++)
		if (pthread_create(&th[started], NULL, fn, &jobs[started]) != 0) {
			rc = -1;
			break;
		}
	for (uint32_t i = 0; i < started; i++)
		pthread_join(th[i], NULL);
	free(th);
	return rc;
}

int oracle_vectorscope_mt(uint8_t *out, const struct oracle_surface *sd, uint32_t threads, uint32_t reps)
{
	memset(out, 0, 65536);
	if (threads == 0)
		return -1;
	if (!sd->yuv_data)
		return 0;
	struct mt_job *jobs = calloc(threads, sizeof *jobs);
	uint32_t *priv = calloc((size_t)threads * 65536, sizeof(uint32_t));
	int rc = (jobs && priv) ? 0 : -1;
	if (!rc) {
		for (uint32_t t = 0; t < threads; t++)
			jobs[t] = (struct mt_job){sd, sd->yuv_data, 0, t, threads, reps ? reps : 1, priv + (size_t)t * 65536, NULL};
		rc = mt_run(mt_vectorscope, jobs, threads);
		for (uint32_t i = 0; i < 65536 && !rc; i++) {
			uint64_t sum = 0;
			for (uint32_t t = 0; t < threads; t++)
				sum += priv[(size_t)t * 65536 + i];
			out[i] = sum > 255 ? 255 : (uint8_t)sum;
		}
	}
	free(priv);
	free(jobs);
	return rc;
}

int oracle_histogram_counts_mt(uint32_t *counts, const struct oracle_surface *sd, uint32_t components, uint32_t threads,
			       uint32_t reps)
{
	memset(counts, 0, 1024 * sizeof(uint32_t));
	if (threads == 0)
		return -1;
	const uint8_t *plane = pick_plane(sd, components);
	if (!plane)
		return 0;
	struct mt_job *jobs = calloc(threads, sizeof *jobs);
	uint32_t *priv = calloc((size_t)threads * 1024, sizeof(uint32_t));
	int rc = (jobs && priv) ? 0 : -1;
	if (!rc) {
		for (uint32_t t = 0; t < threads; t++)
			jobs[t] = (struct mt_job){sd, plane, components, t, threads, reps ? reps : 1, priv + (size_t)t * 1024, NULL};
		rc = mt_run(mt_histogram, jobs, threads);
		for (uint32_t i = 0; i < 1024 && !rc; i++)
			for (uint32_t t = 0; t < threads; t++)
				counts[i] += priv[(size_t)t * 1024 + i];
	}
	free(priv);
	free(jobs);
	return rc;
}

int oracle_waveform_mt(uint8_t *out, const struct oracle_surface *sd, uint32_t components, uint32_t threads, uint32_t reps)
{
	memset(out, 0, (size_t)sd->width * 4 * 256);
	if (threads == 0)
		return -1;
	const uint8_t *plane = pick_plane(sd, components);
	if (!plane)
		return 0;
	struct mt_job *jobs = calloc(threads, sizeof *jobs);
	int rc = jobs ? 0 : -1;
	if (!rc) {
		for (uint32_t t = 0; t < threads; t++)
			jobs[t] = (struct mt_job){sd, plane, components, t, threads, reps ? reps : 1, NULL, out};
		rc = mt_run(mt_waveform, jobs, threads);
	}
	free(jobs);
	return rc;
}

int oracle_region_size(uint32_t width, uint32_t height, uint32_t scale, uint32_t *scaled_w, uint32_t *scaled_h)
{
	if (scale == 0)
		return -1;
	*scaled_w = width / scale; /* src/common.c:249-250 */
	*scaled_h = height / scale;
	return 0;
}

int oracle_stage_region(const uint8_t *src, uint32_t src_linesize, uint32_t width, uint32_t height, uint32_t scale,
			uint32_t x, uint32_t y, uint32_t cx, uint32_t cy, uint8_t *dst, uint32_t dst_linesize)
{
	uint32_t sw, sh;
	if (oracle_region_size(width, height, scale, &sw, &sh))
		return -1;
	if (cx == 0 || cy == 0 || x >= sw || y >= sh || cx > sw - x || cy > sh - y)
		return -1;
	for (uint32_t j = 0; j < cy; j++) {
		const uint8_t *row = src + (size_t)((y + j) * scale + scale / 2) * src_linesize;
		for (uint32_t i = 0; i < cx; i++)
			memcpy(dst + (size_t)j * dst_linesize + (size_t)i * 4, row + (size_t)((x + i) * scale + scale / 2) * 4, 4);
	}
	return 0;
}

void oracle_display_size(uint32_t base_w, uint32_t base_h, uint32_t components, int display, uint32_t *out_w,
			 uint32_t *out_h)
{
	const uint32_t n = disp_n_components(components);
	*out_w = display == 2 ? base_w * n : base_w; /* src/waveform.c:349-351, histogram.c:537-539 */
	*out_h = display == 1 ? base_h * n : base_h; /* src/waveform.c:345-347, histogram.c:533-535 */
}

/* Which pixel shader runs and where it samples: the technique names of waveform.c:343-356. */
struct disp_pick {
	int overlay; /* DrawOverlay: all three channels at once */
	int chan;    /* otherwise: channel index i into the sampled texel / colour row */
	float u, v;  /* texture coordinate the shader samples at */
	float level; /* histogram only: the coordinate the threshold uses (uv.y or its per-band fraction) */
};

static struct disp_pick disp_select(float uvx, float uvy, uint32_t n, int display)
{
	struct disp_pick p = {1, 0, uvx, uvy, uvy};
	if (display == 1 && n == 3) { /* PSDrawStack */
		p.overlay = 0;
		p.chan = (int)(uvy * 3);
		p.v = uvy * 3 - (float)p.chan;
		p.level = p.v;
	} else if (display == 1 && n == 2) { /* PSDrawStackUV */
		p.overlay = 0;
		p.chan = uvy < 0.5f ? 0 : 2;
		p.v = uvy * 2;
		p.level = uvy < 0.5f ? uvy * 2 : uvy * 2 - 1;
	} else if (display == 2 && n == 3) { /* PSDrawParade */
		p.overlay = 0;
		p.chan = (int)(uvx * 3);
		p.u = uvx * 3 - (float)p.chan;
	} else if (display == 2 && n == 2) { /* PSDrawParadeUV */
		p.overlay = 0;
		p.chan = uvx < 0.5f ? 0 : 2;
		p.u = uvx * 2;
	}
	return p;
}

/* Point sampling with Repeat addressing: texel index of coordinate t on an axis of `size` texels. */
static uint32_t disp_texel(float t, uint32_t size)
{
	const float w = t - floorf(t);
	uint32_t i = (uint32_t)floorf(w * (float)size);
	return i >= size ? size - 1 : i;
}

void oracle_display_vectorscope(uint8_t *out, const uint8_t *table, int intensity, int color_type, int colorspace)
{
	oracle_display_vectorscope_zoom(out, table, intensity, color_type, colorspace, 1.0f);
}

void oracle_display_vectorscope_zoom(uint8_t *out, const uint8_t *table, int intensity, int color_type, int colorspace,
				     float zoom)
{
	/* src/vectorscope.c:418-439; the zoom transform of :395-408 applied to the 256x256 sprite */
	const int zoomed = zoom > 1.01f;
	const float ofst = 127.5f * (1.0f - zoom);
	float color[4] = {1.0f, 1.0f, 1.0f, 1.0f}, cu[3] = {0, 0, 0}, cv[3] = {0, 0, 0};
	if (color_type == 1) {
		color[0] = color[1] = color[2] = 0.5f;
		if (colorspace == 1) {
			cu[1] = -0.3441f, cu[2] = +1.772f;
			cv[0] = +1.402f, cv[1] = -0.7141f;
		} else {
			cu[1] = -0.1873f, cu[2] = +1.8556f;
			cv[0] = +1.5748f, cv[1] = -0.4681f;
		}
	}
	for (uint32_t py = 0; py < 256; py++) {
		for (uint32_t px = 0; px < 256; px++) {
			float uvx = ((float)px + 0.5f) / 256.0f, uvy = ((float)py + 0.5f) / 256.0f;
			uint32_t tx = px, ty = py;
			if (zoomed) { /* the pixel centre taken back through x' = zoom * x + ofst; point filter */
				uvx = (((float)px + 0.5f) - ofst) / zoom / 256.0f;
				uvy = (((float)py + 0.5f) - ofst) / zoom / 256.0f;
				tx = disp_texel(uvx, 256);
				ty = disp_texel(uvy, 256);
			}
			const float s = uvx * 2.0f - 1.0f, t = 1.0f - uvy * 2.0f;
			float r = ((float)table[ty * 256 + tx] / 255.0f) * (float)intensity;
			if (r > 1.0f)
				r = 1.0f;
			uint8_t *o = out + ((size_t)py * 256 + px) * 4;
			for (int c = 0; c < 3; c++) {
				float rgb = color[c] + cu[c] * s;
				rgb = rgb + cv[c] * t;
				o[c] = k0_quantise(rgb * r);
			}
			o[3] = k0_quantise(color[3]);
		}
	}
}

void oracle_display_waveform(uint8_t *out, const uint8_t *table, uint32_t width, uint32_t components, int display,
			     int intensity)
{
	uint32_t ow, oh;
	oracle_display_size(width, 256, components, display, &ow, &oh);
	const uint32_t n = disp_n_components(components);
	for (uint32_t py = 0; py < oh; py++) {
		for (uint32_t px = 0; px < ow; px++) {
			const float uvx = ((float)px + 0.5f) / (float)ow, uvy = ((float)py + 0.5f) / (float)oh;
			const struct disp_pick p = disp_select(uvx, uvy, n, display);
			const uint8_t *texel = table + ((size_t)disp_texel(p.v, 256) * width + disp_texel(p.u, width)) * 4;
			uint8_t *o = out + ((size_t)py * ow + px) * 4;
			/* a GS_BGRX texel samples as x = R = byte 2, y = G = byte 1, z = B = byte 0 */
			if (p.overlay) {
				for (int c = 0; c < 3; c++) {
					float r = ((float)texel[2 - c] / 255.0f) * (float)intensity;
					if (r > 1.0f)
						r = 1.0f;
					o[c] = k0_quantise(r);
				}
			} else {
				float r = ((float)texel[2 - p.chan] / 255.0f) * (float)intensity;
				if (r > 1.0f)
					r = 1.0f;
				for (int c = 0; c < 3; c++)
					o[c] = k0_quantise(disp_color[p.chan][c] * r);
			}
			o[3] = 255;
		}
	}
}

void oracle_display_histogram(uint8_t *out, const float *tex, const uint32_t *hi_max, uint32_t components, int display,
			      uint32_t level_height)
{
	uint32_t ow, oh;
	oracle_display_size(256, level_height, components, display, &ow, &oh);
	const uint32_t n = disp_n_components(components);
	const float hm[3] = {(float)hi_max[0], (float)hi_max[1], (float)hi_max[2]}; /* his_set_image, histogram.c:428-429 */
	for (uint32_t py = 0; py < oh; py++) {
		for (uint32_t px = 0; px < ow; px++) {
			const float uvx = ((float)px + 0.5f) / (float)ow, uvy = ((float)py + 0.5f) / (float)oh;
			const struct disp_pick p = disp_select(uvx, uvy, n, display);
			/* the histogram parade samples at uv.x*3 (uv.x*2) without subtracting i: Repeat wraps it */
			const float *texel = tex + (size_t)disp_texel(p.u, 256) * 4;
			uint8_t *o = out + ((size_t)py * ow + px) * 4;
			if (p.overlay) {
				for (int c = 0; c < 3; c++)
					o[c] = texel[c] >= (1 - uvy) * hm[c] ? 255 : 0;
			} else {
				const float r = texel[p.chan] >= (1 - p.level) * hm[p.chan] ? 1.0f : 0.0f;
				for (int c = 0; c < 3; c++)
					o[c] = k0_quantise(disp_color[p.chan][c] * r);
			}
			o[3] = 255;
		}
	}
}

/* ------------------------------------------------------- test inputs ---- */

static inline uint64_t xorshift64(uint64_t *s)
{
	uint64_t x = *s;
	x ^= x << 13;
	x ^= x >> 7;
	x ^= x << 17;
	return *s = x;
}

void oracle_fill_frame(uint8_t *dst, uint32_t linesize, uint32_t width, uint32_t height, int dist, uint64_t seed)
{
	static const uint8_t bars[8][3] = {
		/* B, G, R : white yellow cyan green magenta red blue black at 75 % */
		{191, 191, 191}, {0, 191, 191}, {191, 191, 0}, {0, 191, 0},
		{191, 0, 191},   {0, 0, 191},   {191, 0, 0},   {0, 0, 0},
	};
	uint64_t s = seed ? seed : 0x9E3779B97F4A7C15ull;
	for (uint32_t row = 0; row < height; row++) {
		uint8_t *p = dst + (size_t)linesize * row;
		for (uint32_t col = 0; col < width; col++, p += 4) {
			const uint64_t r = xorshift64(&s);
			switch (dist) {
			default:
			case 1:
			case 5:
				p[0] = (uint8_t)r;
				p[1] = (uint8_t)(r >> 8);
				p[2] = (uint8_t)(r >> 16);
				break;
			case 2:
				p[0] = (uint8_t)(120 + (r & 15));
				p[1] = (uint8_t)(120 + ((r >> 8) & 15));
				p[2] = (uint8_t)(120 + ((r >> 16) & 15));
				break;
			case 3:
				p[0] = 0xD7;
				p[1] = 0xCC;
				p[2] = 0xC6;
				break;
			case 4: {
				const uint32_t bar = (uint32_t)((uint64_t)col * 8 / width);
				p[0] = bars[bar][0];
				p[1] = bars[bar][1];
				p[2] = bars[bar][2];
			} break;
			}
			p[3] = 255;
			if (dist == 5 && (((row >> 4) ^ (col >> 4)) & 1))
				p[3] = 0;
		}
		for (uint32_t k = width * 4; k < linesize; k++)
			dst[(size_t)linesize * row + k] = 0xA5;
	}
}

uint64_t oracle_fnv1a64(const void *data, uint64_t len)
{
	const uint8_t *p = data;
	uint64_t h = 0xcbf29ce484222325ull;
	for (uint64_t i = 0; i < len; i++) {
		h ^= p[i];
		h *= 0x100000001b3ull;
	}
	return h;
}
