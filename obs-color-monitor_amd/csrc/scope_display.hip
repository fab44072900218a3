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
