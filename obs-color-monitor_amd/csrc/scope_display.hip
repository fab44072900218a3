/*
 * scope_display.hip -- D1-D3: the reference's display pixel shaders as output-sized HIP post-passes
 * (SURVEY.md 8f rank 1), so that "frame in -> scope picture out" needs no host round trip.
 *
 *   D1  data/vectorscope.effect:27-33   uniforms src/vectorscope.c:414-439
 *   D2  data/waveform.effect:30-83      technique choice src/waveform.c:334-360
 *   D3  data/histogram.effect:30-85     technique choice src/histogram.c:522-548
 *
 * fp32 in the shaders' operation order (-ffp-contract=off), texel centres at (x+0.5)/size, point
 * sampling with Repeat addressing, FLOAT->UNORM8 store pinned to floor(clamp01(x)*255+0.5) as for
 * K0 (the reference leaves that rounding to the GPU OBS runs on).  One lane per output pixel: these
 * passes touch kilobytes, not frames, and are launch-bound by construction.
 */
#include "scope_kernels.h"

typedef uint32_t u32;

__device__ __forceinline__ u32 disp_quantise(float x)
{
	x = fminf(fmaxf(x, 0.0f), 1.0f);
	return (u32)__fadd_rn(__fmul_rn(x, 255.0f), 0.5f);
}

__device__ __forceinline__ float disp_color(int row, int c)
{
	/* rows of the effects' `color` matrix (data/waveform.effect:4-9, data/histogram.effect:4-9) */
	const float m[3][3] = {{1.00f, 0.41f, 0.41f}, {0.00f, 1.00f, 0.00f}, {0.53f, 0.53f, 1.00f}};
	return m[row][c];
}

/* point sampling, Repeat addressing */
__device__ __forceinline__ u32 disp_texel(float t, u32 size)
{
	const float w = __fsub_rn(t, floorf(t));
	const u32 i = (u32)floorf(__fmul_rn(w, (float)size));
	return i >= size ? size - 1 : i;
}

struct DispPick {
	int overlay, chan;
	float u, v, level;
};

/* which pixel shader runs and where it samples (technique names of src/waveform.c:343-356) */
__device__ __forceinline__ DispPick disp_select(float uvx, float uvy, u32 n, int display)
{
	DispPick p = {1, 0, uvx, uvy, uvy};
	if (display == 1 && n == 3) { /* PSDrawStack */
		p.overlay = 0;
		p.chan = (int)__fmul_rn(uvy, 3.0f);
		p.v = __fsub_rn(__fmul_rn(uvy, 3.0f), (float)p.chan);
		p.level = p.v;
	} else if (display == 1 && n == 2) { /* PSDrawStackUV */
		p.overlay = 0;
		p.chan = uvy < 0.5f ? 0 : 2;
		p.v = __fmul_rn(uvy, 2.0f);
		p.level = uvy < 0.5f ? __fmul_rn(uvy, 2.0f) : __fsub_rn(__fmul_rn(uvy, 2.0f), 1.0f);
	} else if (display == 2 && n == 3) { /* PSDrawParade */
		p.overlay = 0;
		p.chan = (int)__fmul_rn(uvx, 3.0f);
		p.u = __fsub_rn(__fmul_rn(uvx, 3.0f), (float)p.chan);
	} else if (display == 2 && n == 2) { /* PSDrawParadeUV */
		p.overlay = 0;
		p.chan = uvx < 0.5f ? 0 : 2;
		p.u = __fmul_rn(uvx, 2.0f);
	}
	return p;
}

__global__ __launch_bounds__(256) void scope_display_vss_kernel(const uint8_t *table, u32 *out, float intensity,
							       float c0, float c1, float c2, float ca, float u0, float u1,
							       float u2, float v0, float v1, float v2, float zoom)
{
	const u32 i = blockIdx.x * blockDim.x + threadIdx.x; /* 65536 pixels */
	const u32 px = i & 255u, py = i >> 8;
	float uvx = __fadd_rn((float)px, 0.5f) / 256.0f, uvy = __fadd_rn((float)py, 0.5f) / 256.0f;
	u32 texel = i;
	if (zoom > 1.01f) { /* src/vectorscope.c:395-408: the sprite scaled about 127.5; point filter */
		const float ofst = __fmul_rn(127.5f, __fsub_rn(1.0f, zoom));
		uvx = __fsub_rn(__fadd_rn((float)px, 0.5f), ofst) / zoom / 256.0f;
		uvy = __fsub_rn(__fadd_rn((float)py, 0.5f), ofst) / zoom / 256.0f;
		texel = disp_texel(uvy, 256) * 256u + disp_texel(uvx, 256);
	}
	const float s = __fsub_rn(__fmul_rn(uvx, 2.0f), 1.0f), t = __fsub_rn(1.0f, __fmul_rn(uvy, 2.0f));
	float r = __fmul_rn((float)table[texel] / 255.0f, intensity);
	if (r > 1.0f)
		r = 1.0f;
	const float col[3] = {c0, c1, c2}, cu[3] = {u0, u1, u2}, cv[3] = {v0, v1, v2};
	u32 o = disp_quantise(ca) << 24;
#pragma unroll
	for (int c = 0; c < 3; c++) {
		float rgb = __fadd_rn(col[c], __fmul_rn(cu[c], s));
		rgb = __fadd_rn(rgb, __fmul_rn(cv[c], t));
		o |= disp_quantise(__fmul_rn(rgb, r)) << (8 * c);
	}
	out[i] = o;
}

__global__ __launch_bounds__(256) void scope_display_wvs_kernel(const uint8_t *table, u32 *out, u32 width, u32 n,
							       int display, float intensity, u32 ow, u32 oh)
{
	const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= ow * oh)
		return;
	const u32 py = i / ow, px = i - py * ow;
	const float uvx = __fadd_rn((float)px, 0.5f) / (float)ow, uvy = __fadd_rn((float)py, 0.5f) / (float)oh;
	const DispPick p = disp_select(uvx, uvy, n, display);
	const uint8_t *texel = table + ((size_t)disp_texel(p.v, 256) * width + disp_texel(p.u, width)) * 4;
	u32 o = 0xff000000u;
	/* a GS_BGRX texel samples as x = R = byte 2, y = G = byte 1, z = B = byte 0 */
	if (p.overlay) {
#pragma unroll
		for (int c = 0; c < 3; c++) {
			float r = __fmul_rn((float)texel[2 - c] / 255.0f, intensity);
			if (r > 1.0f)
				r = 1.0f;
			o |= disp_quantise(r) << (8 * c);
		}
	} else {
		float r = __fmul_rn((float)texel[2 - p.chan] / 255.0f, intensity);
		if (r > 1.0f)
			r = 1.0f;
#pragma unroll
		for (int c = 0; c < 3; c++)
			o |= disp_quantise(__fmul_rn(disp_color(p.chan, c), r)) << (8 * c);
	}
	out[i] = o;
}

__global__ __launch_bounds__(256) void scope_display_his_kernel(const float *tex, u32 *out, float hm0, float hm1,
							       float hm2, u32 n, int display, u32 ow, u32 oh)
{
	const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= ow * oh)
		return;
	const u32 py = i / ow, px = i - py * ow;
	const float uvx = __fadd_rn((float)px, 0.5f) / (float)ow, uvy = __fadd_rn((float)py, 0.5f) / (float)oh;
	const DispPick p = disp_select(uvx, uvy, n, display);
	const float *texel = tex + (size_t)disp_texel(p.u, 256) * 4;
	const float hm[3] = {hm0, hm1, hm2};
	u32 o = 0xff000000u;
	if (p.overlay) {
#pragma unroll
		for (int c = 0; c < 3; c++)
			if (texel[c] >= __fmul_rn(__fsub_rn(1.0f, uvy), hm[c]))
				o |= 0xffu << (8 * c);
	} else {
		const float r = texel[p.chan] >= __fmul_rn(__fsub_rn(1.0f, p.level), hm[p.chan]) ? 1.0f : 0.0f;
#pragma unroll
		for (int c = 0; c < 3; c++)
			o |= disp_quantise(__fmul_rn(disp_color(p.chan, c), r)) << (8 * c);
	}
	out[i] = o;
}

hipError_t scope_launch_display_vss(const uint8_t *table, uint8_t *out, int intensity, int color_type, int colorspace,
				    float zoom, hipStream_t st)
{
	/* src/vectorscope.c:418-439 */
	float c[4] = {1.0f, 1.0f, 1.0f, 1.0f}, cu[3] = {0, 0, 0}, cv[3] = {0, 0, 0};
	if (color_type == 1) {
		c[0] = c[1] = c[2] = 0.5f;
		if (colorspace == 1) {
			cu[1] = -0.3441f, cu[2] = +1.772f;
			cv[0] = +1.402f, cv[1] = -0.7141f;
		} else {
			cu[1] = -0.1873f, cu[2] = +1.8556f;
			cv[0] = +1.5748f, cv[1] = -0.4681f;
		}
	}
	hipLaunchKernelGGL(scope_display_vss_kernel, dim3(256), dim3(256), 0, st, table, (u32 *)out, (float)intensity, c[0],
			   c[1], c[2], c[3], cu[0], cu[1], cu[2], cv[0], cv[1], cv[2], zoom);
	return hipGetLastError();
}

hipError_t scope_launch_display_wvs(const uint8_t *table, uint8_t *out, uint32_t width, uint32_t n_components,
				    int display, int intensity, uint32_t out_w, uint32_t out_h, hipStream_t st)
{
	const u32 total = out_w * out_h;
	if (!total)
		return hipSuccess;
	hipLaunchKernelGGL(scope_display_wvs_kernel, dim3((total + 255) / 256), dim3(256), 0, st, table, (u32 *)out, width,
			   n_components, display, (float)intensity, out_w, out_h);
	return hipGetLastError();
}

hipError_t scope_launch_display_his(const float *tex, uint8_t *out, float hm0, float hm1, float hm2,
				    uint32_t n_components, int display, uint32_t out_w, uint32_t out_h, hipStream_t st)
{
	const u32 total = out_w * out_h;
	if (!total)
		return hipSuccess;
	hipLaunchKernelGGL(scope_display_his_kernel, dim3((total + 255) / 256), dim3(256), 0, st, tex, (u32 *)out, hm0, hm1,
			   hm2, n_components, display, out_w, out_h);
	return hipGetLastError();
}

/* ------------------------------------------------------------------------------------------------------------------
 * Graticule: the label quads (DrawGraticule, data/vectorscope.effect:35-39, src/vectorscope.c:447-454) and the GS_LINES
 * lists drawn with OBS's "Solid" effect (src/vectorscope.c:456-462, src/waveform.c:308-332, src/histogram.c:492-520)
 * over a finished scope picture.
 *
 * The reference leaves rasterisation to the GPU OBS runs on; here it is PINNED, integer-exact and order preserving:
 *   - one lane per picture pixel walks the quads, then the segments, in list order, and blends whatever covers its
 *     pixel centre (x + 0.5, y + 0.5) -- a pixel two lines cross is blended twice, as two draws would;
 *   - a segment (x0, y0)-(x1, y1) covers, along its major axis, the pixels whose centre coordinate c lies in
 *     [min, max) of the end points, and in each of them the pixel floor(p0 + (c - c0) * (dp / dc)) of the minor axis
 *     (fp32, this operation order, no contraction); a segment of zero length covers nothing;
 *   - a quad covers the pixels whose centre lies in [x, x + w) x [y, y + h) and samples the label strip with a point
 *     filter at u0 + (cx - x) / w * (u1 - u0) (v likewise); glyph texels are 0 or 1;
 *   - blend (OBS's default source-alpha / inverse-source-alpha on all four channels): out = src * a + dst * (1 - a)
 *     with a = colour alpha x glyph texel, dst = byte / 255, stored floor(clamp01(x) * 255 + 0.5) like every display
 *     pass of this file.
 * The label strip is this build's own 5x7 glyphs ("R", "B", "Cy", "Yl", "G", "Mg" in the order of the colour targets,
 * 16x16 texels each): the reference's data/vectorscope-graticule.png is an asset of the reference and is not copied. */
#define GRAT_STRIP_W 96u
#define GRAT_STRIP_H 16u

struct ScopeGratQuad {
	float x, y, w, h, u0, v0, u1, v1;
};

__device__ __forceinline__ u32 grat_blend(u32 dst, float r, float g, float b, float a)
{
	const float src[4] = {r, g, b, a};
	const float ia = __fsub_rn(1.0f, a);
	u32 o = 0;
#pragma unroll
	for (int c = 0; c < 4; c++) {
		const float d = (float)((dst >> (8 * c)) & 0xffu) / 255.0f;
		o |= disp_quantise(__fadd_rn(__fmul_rn(src[c], a), __fmul_rn(d, ia))) << (8 * c);
	}
	return o;
}

__global__ __launch_bounds__(256) void scope_graticule_kernel(u32 *pic, u32 w, u32 h, const ScopeGratQuad *quads, u32 n_quads,
							     const float *segs, u32 n_segs, const uint8_t *strip, float r,
							     float g, float b, float a)
{
	const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= w * h)
		return;
	const u32 py = i / w, px = i - py * w;
	const float cx = __fadd_rn((float)px, 0.5f), cy = __fadd_rn((float)py, 0.5f);
	u32 o = pic[i];
	for (u32 q = 0; q < n_quads; q++) {
		const ScopeGratQuad k = quads[q];
		if (!(cx >= k.x && cx < __fadd_rn(k.x, k.w) && cy >= k.y && cy < __fadd_rn(k.y, k.h)))
			continue;
		const float u = __fadd_rn(k.u0, __fmul_rn(__fsub_rn(cx, k.x) / k.w, __fsub_rn(k.u1, k.u0)));
		const float v = __fadd_rn(k.v0, __fmul_rn(__fsub_rn(cy, k.y) / k.h, __fsub_rn(k.v1, k.v0)));
		int tx = (int)floorf(__fmul_rn(u, (float)GRAT_STRIP_W)), ty = (int)floorf(__fmul_rn(v, (float)GRAT_STRIP_H));
		tx = tx < 0 ? 0 : tx >= (int)GRAT_STRIP_W ? (int)GRAT_STRIP_W - 1 : tx; /* Clamp addressing, data/vectorscope.effect:8-12 */
		ty = ty < 0 ? 0 : ty >= (int)GRAT_STRIP_H ? (int)GRAT_STRIP_H - 1 : ty;
		if (strip[(u32)ty * GRAT_STRIP_W + (u32)tx])
			o = grat_blend(o, r, g, b, a);
	}
	for (u32 s = 0; s < n_segs; s++) {
		const float x0 = segs[4 * s], y0 = segs[4 * s + 1], x1 = segs[4 * s + 2], y1 = segs[4 * s + 3];
		const float dx = __fsub_rn(x1, x0), dy = __fsub_rn(y1, y0);
		bool hit = false;
		if (fabsf(dx) >= fabsf(dy)) {
			if (dx != 0.0f && cx >= fminf(x0, x1) && cx < fmaxf(x0, x1))
				hit = floorf(__fadd_rn(y0, __fmul_rn(__fsub_rn(cx, x0), dy / dx))) == (float)py;
		} else if (cy >= fminf(y0, y1) && cy < fmaxf(y0, y1)) {
			hit = floorf(__fadd_rn(x0, __fmul_rn(__fsub_rn(cy, y0), dx / dy))) == (float)px;
		}
		if (hit)
			o = grat_blend(o, r, g, b, a);
	}
	pic[i] = o;
}

void scope_graticule_strip(uint8_t *alpha)
{
	/* 5 x 7 glyphs, one row per byte, bit 4 = left column */
	static const uint8_t R[7] = {0x1e, 0x11, 0x11, 0x1e, 0x14, 0x12, 0x11}, B[7] = {0x1e, 0x11, 0x11, 0x1e, 0x11, 0x11, 0x1e},
			     Cc[7] = {0x0e, 0x11, 0x10, 0x10, 0x10, 0x11, 0x0e}, G[7] = {0x0e, 0x11, 0x10, 0x17, 0x11, 0x11, 0x0f},
			     M[7] = {0x11, 0x1b, 0x15, 0x15, 0x11, 0x11, 0x11}, Y[7] = {0x11, 0x11, 0x0a, 0x04, 0x04, 0x04, 0x04},
			     y[7] = {0x00, 0x00, 0x11, 0x11, 0x0f, 0x01, 0x0e}, l[7] = {0x0c, 0x04, 0x04, 0x04, 0x04, 0x04, 0x0e},
			     gg[7] = {0x00, 0x00, 0x0f, 0x11, 0x0f, 0x01, 0x0e};
	static const uint8_t *const cell[6][2] = {{R, 0}, {B, 0}, {Cc, y}, {Y, l}, {G, 0}, {M, gg}}; /* the order of the colour targets */
	for (u32 i = 0; i < GRAT_STRIP_W * GRAT_STRIP_H; i++)
		alpha[i] = 0;
	for (u32 c = 0; c < 6; c++) {
		const u32 n = cell[c][1] ? 2u : 1u;
		const u32 x0 = c * 16u + (16u - (n * 5u + (n - 1u))) / 2u, y0 = 4u; /* centred: 5 or 11 of 16 columns, rows 4..10 */
		for (u32 k = 0; k < n; k++)
			for (u32 row = 0; row < 7; row++)
				for (u32 col = 0; col < 5; col++)
					if ((cell[c][k][row] >> (4u - col)) & 1u)
						alpha[(y0 + row) * GRAT_STRIP_W + x0 + k * 6u + col] = 1;
	}
}

hipError_t scope_launch_graticule(uint8_t *pic, uint32_t w, uint32_t h, const float *quads, uint32_t n_quads, const float *segs,
				  uint32_t n_segs, const uint8_t *strip, uint32_t argb, hipStream_t st)
{
	const u32 total = w * h;
	if (!total || (!n_quads && !n_segs))
		return hipSuccess;
	/* gs_effect_set_color takes 0xAARRGGBB (src/vectorscope.c:141-144: amber 0x80FFBF00, green 0x8000FF00) */
	const float a = (float)(argb >> 24) / 255.0f, r = (float)((argb >> 16) & 0xffu) / 255.0f, g = (float)((argb >> 8) & 0xffu) / 255.0f,
		    b = (float)(argb & 0xffu) / 255.0f;
	hipLaunchKernelGGL(scope_graticule_kernel, dim3((total + 255) / 256), dim3(256), 0, st, (u32 *)pic, w, h,
			   (const ScopeGratQuad *)quads, n_quads, segs, n_segs, strip, r, g, b, a);
	return hipGetLastError();
}
