/*
 * host_api.c -- what a host (OBS's render thread in the plugin; the tests and bench here) reads back
 * from the scope sources: the table the worker finished last, i.e. tex_buf[w_tex_buf ^ 1], exactly
 * what vss_render / wvs_render / his_render upload with gs_texture_set_image
 * (src/vectorscope.c:410-412, src/waveform.c:399, src/histogram.c:563).
 */
#include "scope_sources.h"

struct cm_source *scope_host_cm(void *source)
{
	return (struct cm_source *)source; /* every source starts with its struct cm_source */
}

const uint8_t *scope_host_vss_table(void *data, int *colorspace)
{
	struct vss_source *src = data;
	pthread_mutex_lock(&src->tex_mutex);
	const int r = src->w_tex_buf ^ 1;
	pthread_mutex_unlock(&src->tex_mutex);
	if (colorspace)
		*colorspace = src->tex_cs[r];
	return src->tex_buf[r];
}

const uint8_t *scope_host_wvs_table(void *data, uint32_t *width)
{
	struct wvs_source *src = data;
	pthread_mutex_lock(&src->tex_mutex);
	const int r = src->w_tex_buf ^ 1;
	pthread_mutex_unlock(&src->tex_mutex);
	if (width)
		*width = src->tex_buf_width[r];
	return src->tex_buf[r];
}

const float *scope_host_his_table(void *data, uint32_t *hi_max)
{
	struct his_source *src = data;
	pthread_mutex_lock(&src->tex_mutex);
	const int r = src->w_tex_buf ^ 1;
	pthread_mutex_unlock(&src->tex_mutex);
	if (hi_max)
		for (int i = 0; i < 3; i++)
			hi_max[i] = src->hi_max[r][i];
	return (const float *)src->tex_buf[r];
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
