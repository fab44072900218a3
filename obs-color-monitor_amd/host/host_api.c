/*
 * host_api.c -- what a host (OBS's render thread in the plugin; the tests and bench here) reads back
 * from the scope sources: the table the worker finished last, i.e. tex_buf[w_tex_buf ^ 1], exactly
 * what vss_render / wvs_render / his_render upload with gs_texture_set_image
 * (src/vectorscope.c:410-412, src/waveform.c:399, src/histogram.c:563).
 */
#include "scope_sources.h"
#include <string.h>

struct cm_source *scope_host_cm(void *source)
{
	return (struct cm_source *)source; /* every source starts with its struct cm_source */
}

/* The pointer forms hand out tex_buf[r] itself: legal only while the source is QUIET (after cm_drain with nothing staged since, or
 * with its threads stopped) -- two flips later the worker writes that buffer again and ensure_tex_buf_size may free it.  The
 * metadata is read under the same lock as the index.  A host that reads while frames flow takes the *_copy forms below. */
const uint8_t *scope_host_vss_table(void *data, int *colorspace)
{
	struct vss_source *src = data;
	pthread_mutex_lock(&src->tex_mutex);
	const int r = src->w_tex_buf ^ 1;
	const uint8_t *t = src->tex_buf[r];
	if (colorspace)
		*colorspace = src->tex_cs[r];
	pthread_mutex_unlock(&src->tex_mutex);
	return t;
}

const uint8_t *scope_host_wvs_table(void *data, uint32_t *width)
{
	struct wvs_source *src = data;
	pthread_mutex_lock(&src->tex_mutex);
	const int r = src->w_tex_buf ^ 1;
	const uint8_t *t = src->tex_buf[r];
	if (width)
		*width = src->tex_buf_width[r];
	pthread_mutex_unlock(&src->tex_mutex);
	return t;
}

const float *scope_host_his_table(void *data, uint32_t *hi_max)
{
	struct his_source *src = data;
	pthread_mutex_lock(&src->tex_mutex);
	const int r = src->w_tex_buf ^ 1;
	const float *t = (const float *)src->tex_buf[r];
	if (hi_max)
		for (int i = 0; i < 3; i++)
			hi_max[i] = src->hi_max[r][i];
	pthread_mutex_unlock(&src->tex_mutex);
	return t;
}

/* The same tables COPIED OUT under the source's tex_mutex -- the section the render side holds while it reads tex_buf[w_tex_buf ^ 1]
 * (scope_sources.h: the worker flips only under it, so the buffer cannot be rewritten or freed while the copy runs).  Safe with
 * frames in flight.  Return the bytes copied; 0 when there is no table yet or `cap` is too small (nothing written then, *width
 * still says what is needed). */
size_t scope_host_vss_copy(void *data, uint8_t *out, size_t cap, int *colorspace)
{
	struct vss_source *src = data;
	size_t n = 0;
	pthread_mutex_lock(&src->tex_mutex);
	const int r = src->w_tex_buf ^ 1;
	if (colorspace)
		*colorspace = src->tex_cs[r];
	if (src->tex_buf[r] && out && cap >= (size_t)VS_SIZE * VS_SIZE) {
		n = (size_t)VS_SIZE * VS_SIZE;
		memcpy(out, src->tex_buf[r], n);
	}
	pthread_mutex_unlock(&src->tex_mutex);
	return n;
}

size_t scope_host_wvs_copy(void *data, uint8_t *out, size_t cap, uint32_t *width)
{
	struct wvs_source *src = data;
	size_t n = 0;
	pthread_mutex_lock(&src->tex_mutex);
	const int r = src->w_tex_buf ^ 1;
	const size_t need = (size_t)WV_SIZE * src->tex_buf_width[r] * 4;
	if (width)
		*width = src->tex_buf_width[r];
	if (src->tex_buf[r] && out && need && cap >= need) {
		n = need;
		memcpy(out, src->tex_buf[r], n);
	}
	pthread_mutex_unlock(&src->tex_mutex);
	return n;
}

size_t scope_host_his_copy(void *data, float *out, size_t cap, uint32_t *hi_max)
{
	struct his_source *src = data;
	size_t n = 0;
	pthread_mutex_lock(&src->tex_mutex);
	const int r = src->w_tex_buf ^ 1;
	if (hi_max)
		for (int i = 0; i < 3; i++)
			hi_max[i] = src->hi_max[r][i];
	if (src->tex_buf[r] && out && cap >= (size_t)HI_SIZE * 4 * sizeof(float)) {
		n = (size_t)HI_SIZE * 4 * sizeof(float);
		memcpy(out, src->tex_buf[r], n);
	}
	pthread_mutex_unlock(&src->tex_mutex);
	return n;
}

/* The level settings his_update left behind (src/histogram.c:131-156): what the next K3b pass and the graticule use. */
void scope_host_his_levels(void *data, int *level_fixed_value, int *level_ratio_value, float *graticule_horizontal_step)
{
	const struct his_source *src = data;
	if (level_fixed_value)
		*level_fixed_value = src->level_fixed_value;
	if (level_ratio_value)
		*level_ratio_value = src->level_ratio_value;
	if (graticule_horizontal_step)
		*graticule_horizontal_step = src->graticule_horizontal_step;
}

uint64_t scope_host_frames_processed(void *source)
{
	return scope_host_cm(source)->frames_processed;
}

uint64_t scope_host_frames_dropped(void *source)
{
	return scope_host_cm(source)->frames_dropped;
}

uint64_t scope_host_plane_uploads(void *source)
{
	return scope_host_cm(source)->plane_uploads;
}

uint64_t scope_host_plane_reuses(void *source)
{
	return scope_host_cm(source)->plane_reuses;
}

/* The picture the source's video_render produced (RGBA8, row-major); NULL before the first render. */
const uint8_t *scope_host_picture(void *source, int kind, uint32_t *width, uint32_t *height)
{
	const struct scope_picture *pic = kind == 0   ? &((struct vss_source *)source)->pic
					  : kind == 1 ? &((struct wvs_source *)source)->pic
						      : &((struct his_source *)source)->pic;
	if (width)
		*width = pic->width;
	if (height)
		*height = pic->height;
	return pic->rgba;
}
