/*
 * scope_common.c -- the staging ring and the "color-monitor" consumer thread behind the reference's cm_* interface
 * (src/common.c:223-403; the ring itself is this build's own single-producer single-consumer sequence ring, scope_common.h)
 * for the stand-alone host, and the module's shared GPU contexts.
 *
 * One reference quirk is NOT reproduced: in the reference the producer stacks a raw RGB copy in
 * front of the YUV rows whenever CM_FLAG_RAW_TEXTURE is set (src/common.c:191-205) but the consumer skips a first plane only
 * for CM_FLAG_CONVERT_RGB (:358-364), so an ROI (always RAW_TEXTURE, src/roi.c:35) whose only scope is a vectorscope
 * hands RGB rows to it as yuv_data.  Here the staged surface holds exactly the planes the flags ask for and the
 * vectorscope gets the Y'CbCr rows.  Default docks never hit the case (waveform and histogram default to RGB).
 */
#include "scope_common.h"

#include <stdlib.h>
#include <string.h>

/* Two contexts: the pipeline threads share one (scope_surface_begin keeps it locked for the 1-2 ms a 4K surface takes to
 * cross PCIe), the render thread's display passes have their own, so OBS's render thread never waits for a "color-monitor"
 * thread -- in the reference it never waits for the CPU loops either. */
static scope_ctx *g_gpu, *g_gpu_render;
static pthread_mutex_t g_gpu_mutex = PTHREAD_MUTEX_INITIALIZER;

int cm_gpu_open(int device)
{
	pthread_mutex_lock(&g_gpu_mutex);
	int st = SCOPE_OK;
	if (!g_gpu) {
		st = scope_ctx_create(device, &g_gpu);
		if (st == SCOPE_OK && (st = scope_ctx_create(device, &g_gpu_render)) != SCOPE_OK) {
			scope_ctx_destroy(g_gpu);
			g_gpu = NULL;
		}
		if (st != SCOPE_OK)
			blog(LOG_ERROR, "[obs-color-monitor] scope_ctx_create(%d) failed: %s", device, scope_strerror(st));
	}
	pthread_mutex_unlock(&g_gpu_mutex);
	return st;
}

void cm_gpu_close(void)
{
	pthread_mutex_lock(&g_gpu_mutex);
	scope_ctx_destroy(g_gpu);
	scope_ctx_destroy(g_gpu_render);
	g_gpu = g_gpu_render = NULL;
	pthread_mutex_unlock(&g_gpu_mutex);
}

scope_ctx *cm_gpu(void)
{
	return g_gpu;
}

scope_ctx *cm_gpu_render(void)
{
	return g_gpu_render;
}

/* src/common.c:335-373: build the surface descriptor from the slot's flags and call the scope */
static void cm_pipeline_thread_loop(struct cm_source *src, struct cm_surface_queue_item *item)
{
	if (!(item->flags & (CM_FLAG_CONVERT_RGB | CM_FLAG_CONVERT_YUV)))
		return;
	uint8_t *video_data = item->video_data;
	if (!video_data)
		return; /* the reference returns when gs_stagesurface_map fails, :349-350 */

	struct cm_surface_data surface_data = {
		.linesize = item->linesize,
		.width = item->width,
		.height = item->height,
		.colorspace = item->colorspace,
	};
	if (item->flags & CM_FLAG_CONVERT_RGB) {
		surface_data.rgb_data = video_data;
		video_data += (size_t)item->linesize * item->height; /* :358-359 */
	}
	if (item->flags & CM_FLAG_CONVERT_YUV)
		surface_data.yuv_data = video_data;

	if (item->cb) {
		/* one staged surface, possibly several scopes behind the callback (roi_surface_cb fans out, src/roi.c:329-341):
		 * every plane crosses PCIe once however many of them read it */
		PROFILE_START(PROF_PIPELINE_THREAD); /* the reference brackets the whole loop body, src/common.c:393-395 */
		const bool cached = scope_surface_begin(cm_gpu(), 0) == SCOPE_OK;
		item->cb(item->cb_data, &surface_data);
		if (cached) {
			uint32_t uploads = 0, reuses = 0;
			if (scope_surface_end(cm_gpu(), &uploads, &reuses) == SCOPE_OK) {
				src->plane_uploads += uploads;
				src->plane_reuses += reuses;
			}
		}
		PROFILE_END(PROF_PIPELINE_THREAD);
	}
	src->frames_processed++;
}

/* Ring the doorbell: whoever sleeps on it looks at the counters again. */
static void cm_ring_bell(struct cm_source *src)
{
	pthread_mutex_lock(&src->pipeline_mutex);
	pthread_cond_broadcast(&src->pipeline_cond);
	pthread_mutex_unlock(&src->pipeline_mutex);
}

/* The consumer ("color-monitor" thread; the reference's loop is src/common.c:375-403): take the oldest published frame it has not
 * started on, run the callback on its slot, say so.  It sleeps on the doorbell only while nothing is published. */
static void *cm_pipeline_thread(void *data)
{
	struct cm_source *src = data;
	for (;;) {
		const uint64_t k = atomic_load_explicit(&src->taken, memory_order_relaxed); /* only this thread moves it */
		if (atomic_load_explicit(&src->staged, memory_order_acquire) == k) {
			pthread_mutex_lock(&src->pipeline_mutex);
			while (!atomic_load_explicit(&src->request_exit, memory_order_acquire) &&
			       atomic_load_explicit(&src->staged, memory_order_acquire) == k)
				pthread_cond_wait(&src->pipeline_cond, &src->pipeline_mutex);
			pthread_mutex_unlock(&src->pipeline_mutex);
		}
		if (atomic_load_explicit(&src->request_exit, memory_order_acquire))
			break;
		if (atomic_load_explicit(&src->staged, memory_order_acquire) == k)
			continue;
		atomic_store_explicit(&src->taken, k + 1, memory_order_release); /* the producer may now reuse slot (k - 1) % N, not k % N */
		cm_pipeline_thread_loop(src, &src->queue[k % CM_SURFACE_QUEUE_SIZE]);
		atomic_store_explicit(&src->done, k + 1, memory_order_release);
		cm_ring_bell(src); /* lets cm_drain() see the progress */
	}
	return NULL;
}

static void cm_start_thread(struct cm_source *src)
{
	if (src->pipeline_thread_running)
		return;
	atomic_store_explicit(&src->request_exit, false, memory_order_release);
	src->pipeline_thread_running = pthread_create(&src->pipeline_thread, NULL, cm_pipeline_thread, src) == 0;
}

static void cm_stop_thread(struct cm_source *src)
{
	if (!src->pipeline_thread_running)
		return;
	atomic_store_explicit(&src->request_exit, true, memory_order_release);
	cm_ring_bell(src);
	pthread_join(src->pipeline_thread, NULL);
	src->pipeline_thread_running = false;
}

void cm_get_defaults(obs_data_t *settings)
{
	obs_data_set_default_int(settings, "target_scale", 2); /* src/common.c:124, src/vectorscope.c:157 */
	obs_data_set_default_int(settings, "colorspace", 0);
	obs_data_set_default_bool(settings, "bypass", false);
}

void cm_get_properties(struct cm_source *src, obs_properties_t *props)
{
	if (!src)
		return;
	/* "target_name" lists OBS's sources (src/common.c:120-122): the stand-alone host has none to offer */
	obs_properties_add_int(props, "target_scale", obs_module_text("Scale"), 1, 128, 1);
	if (!(src->flags & CM_FLAG_ROI))
		obs_properties_add_bool(props, "bypass", obs_module_text("Bypass"));
}

void cm_enum_sources(void *data, obs_source_enum_proc_t enum_callback, void *param)
{
	struct cm_source *src = data;
	if (src->enumerating)
		return;
	src->enumerating = true;
	if (src->target)
		enum_callback(src->self, src->target, param);
	src->enumerating = false;
}

void cm_update(struct cm_source *src, obs_data_t *settings)
{
	src->target_scale = (int)obs_data_get_int(settings, "target_scale"); /* src/common.c:88-90 */
	if (src->target_scale < 1)
		src->target_scale = 1;
	/* util.c:25-41: 1 -> 601, 2 -> 709, auto -> the video info's, which defaults to 709 */
	const int cs = (int)obs_data_get_int(settings, "colorspace");
	src->colorspace = cs == 1 ? 1 : 2;
	src->bypass = obs_data_get_bool(settings, "bypass");
}

void cm_create(struct cm_source *src, obs_data_t *settings, obs_source_t *source)
{
	uint32_t flags = src->flags; /* the scope sets flags before or after, as in the reference */
	memset(src, 0, sizeof *src);
	src->flags = flags;
	src->self = source;
	atomic_init(&src->staged, 0); /* an empty ring: nothing published, nothing taken */
	atomic_init(&src->taken, 0);
	atomic_init(&src->done, 0);
	atomic_init(&src->request_exit, false);
	pthread_mutex_init(&src->pipeline_mutex, NULL);
	pthread_cond_init(&src->pipeline_cond, NULL);
	cm_update(src, settings);
}

void cm_destroy(struct cm_source *src)
{
	cm_stop_thread(src);
	for (int i = 0; i < CM_SURFACE_QUEUE_SIZE; i++)
		bfree(src->queue[i].video_data);
	pthread_mutex_destroy(&src->pipeline_mutex);
	pthread_cond_destroy(&src->pipeline_cond);
}

void cm_request(struct cm_source *src, cm_surface_cb_t callback, void *data)
{
	src->callback = callback;
	src->callback_data = data;
}

void cm_tick(void *data, float unused)
{
	UNUSED_PARAMETER(unused);
	struct cm_source *src = data;
	/* src/common.c:587-590: a scope attached to an ROI has no thread of its own */
	if (src->roi || src->bypass)
		cm_stop_thread(src);
	else
		cm_start_thread(src);
}

/* The consumer is a full ring behind -> this frame is not staged but dropped (the reference's rule, src/common.c:260-268).  With
 * `staged - taken == CM_SURFACE_QUEUE_SIZE - 1` frames published and not yet started on, the next frame's slot is the one the
 * consumer may still be reading (frame taken - 1). */
static bool drop_if_behind(struct cm_source *src)
{
	const uint64_t n = atomic_load_explicit(&src->staged, memory_order_relaxed); /* only the producer moves it */
	const bool behind = n - atomic_load_explicit(&src->taken, memory_order_acquire) >= CM_SURFACE_QUEUE_SIZE - 1;
	if (behind)
		src->frames_dropped++;
	return behind;
}

bool cm_stage_surface(struct cm_source *src, const uint8_t *video_data, uint32_t linesize, uint32_t width,
		      uint32_t height)
{
	const bool has_rgb = !src->bypass && (src->flags & CM_FLAG_CONVERT_RGB);
	const bool has_yuv = !src->bypass && (src->flags & CM_FLAG_CONVERT_YUV);
	if (!has_rgb && !has_yuv)
		return false;

	if (drop_if_behind(src))
		return false;
	const uint64_t n = atomic_load_explicit(&src->staged, memory_order_relaxed); /* this frame's number; slot n % N is the producer's */
	const int w = (int)(n % CM_SURFACE_QUEUE_SIZE);

	/* src/common.c:272-282: the ROI rectangle, when valid, selects the sub-rectangle that is staged */
	uint32_t x = 0, y = 0, cx = width, cy = height;
	if ((src->flags & CM_FLAG_ROI) && 0 <= src->x0 && src->x0 < src->x1 && 0 <= src->y0 && src->y0 < src->y1 &&
	    (uint32_t)src->x1 <= width && (uint32_t)src->y1 <= height) {
		x = (uint32_t)src->x0;
		y = (uint32_t)src->y0;
		cx = (uint32_t)src->x1 - x;
		cy = (uint32_t)src->y1 - y;
	}
	struct cm_surface_queue_item *item = &src->queue[w];
	const uint32_t planes = (has_rgb ? 1u : 0u) + (has_yuv ? 1u : 0u);
	const uint32_t sheight = cy * planes; /* :284-291 */
	const uint32_t out_linesize = cx * 4;
	const size_t bytes = (size_t)out_linesize * sheight;
	if (item->capacity < bytes) {
		bfree(item->video_data);
		item->video_data = bzalloc(bytes);
		item->capacity = item->video_data ? bytes : 0;
	}
	if (item->video_data) { /* gs_draw_sprite_subregion + gs_stage_texture stand-in */
		for (uint32_t p = 0; p < planes; p++)
			for (uint32_t r = 0; r < cy; r++)
				memcpy(item->video_data + ((size_t)p * cy + r) * out_linesize,
				       video_data + ((size_t)p * height + y + r) * linesize + (size_t)x * 4, out_linesize);
	}
	linesize = out_linesize;
	width = cx;
	height = cy;
	item->linesize = linesize;
	item->width = width;
	item->height = height;
	item->sheight = sheight;
	item->flags = src->flags & (CM_FLAG_CONVERT_RGB | CM_FLAG_CONVERT_YUV | CM_FLAG_RAW_TEXTURE);
	item->colorspace = src->colorspace;
	item->cb = src->callback;
	item->cb_data = src->callback_data;

	/* publish: the slot's bytes and fields are written; the release store makes them the consumer's (which takes the frame at
	 * once: nothing here is still in flight, unlike the reference's stage copy, src/common.c:322-329) */
	atomic_store_explicit(&src->staged, n + 1, memory_order_release);
	cm_ring_bell(src);
	return true;
}

bool cm_render_target(struct cm_source *src, const uint8_t *target_rgb, uint32_t linesize, uint32_t target_width,
		      uint32_t target_height)
{
	const bool has_rgb = !src->bypass && (src->flags & CM_FLAG_CONVERT_RGB);
	const bool has_yuv = !src->bypass && (src->flags & CM_FLAG_CONVERT_YUV);
	if (!has_rgb && !has_yuv)
		return false;
	const uint32_t scale = src->target_scale > 0 ? (uint32_t)src->target_scale : 1u;
	const uint32_t scaled_width = target_width / scale, scaled_height = target_height / scale; /* :249-250 */
	if (scaled_width == 0 || scaled_height == 0)
		return false; /* :251-254 */

	if (drop_if_behind(src)) /* before spending GPU time, as the reference does (:260-268) */
		return false;

	struct scope_region region = {scale, 0, 0, 0, 0};
	if ((src->flags & CM_FLAG_ROI) && 0 <= src->x0 && src->x0 < src->x1 && 0 <= src->y0 && src->y0 < src->y1 &&
	    (uint32_t)src->x1 <= scaled_width && (uint32_t)src->y1 <= scaled_height) { /* :272-282 */
		region.x = (uint32_t)src->x0;
		region.y = (uint32_t)src->y0;
		region.cx = (uint32_t)(src->x1 - src->x0);
		region.cy = (uint32_t)(src->y1 - src->y0);
	}
	uint32_t cx, cy;
	if (scope_region_size(target_width, target_height, &region, &cx, &cy) != SCOPE_OK)
		return false;
	PROFILE_START(PROF_RENDER_TARGET);
	const uint32_t out_linesize = cx * 4;
	const size_t plane = (size_t)out_linesize * cy;
	uint8_t *stacked = bzalloc(plane * 2);
	if (!stacked) {
		PROFILE_END(PROF_RENDER_TARGET);
		return false;
	}
	/* RGB rows first, YUV rows after them -- the staged surface's layout (:284-291).  This is the render thread
	 * (video_render, src/vectorscope.c:393): its two GPU calls go to the render thread's own context, so they never queue
	 * behind a pipeline thread that holds cm_gpu() across scope_surface_begin..end (upload + three scopes). */
	scope_ctx *gpu = cm_gpu_render();
	int st = scope_stage_region(gpu, target_rgb, linesize, target_width, target_height, &region, stacked,
				    out_linesize, 0);
	if (st == SCOPE_OK && has_yuv) {
		PROFILE_START(PROF_CONVERT_YUV);
		st = scope_convert_rgb_yuv(gpu, stacked, out_linesize, stacked + plane, out_linesize, cx, cy,
					   src->colorspace, 0);
		PROFILE_END(PROF_CONVERT_YUV);
	}
	PROFILE_END(PROF_RENDER_TARGET);
	bool staged = false;
	if (st == SCOPE_OK) {
		PROFILE_START(PROF_STAGE_SURFACE);
		/* the rectangle is already applied: stage the planes as they are */
		const uint32_t flags = src->flags;
		src->flags &= ~(uint32_t)CM_FLAG_ROI;
		if (has_rgb)
			staged = cm_stage_surface(src, stacked, out_linesize, cx, cy);
		else { /* YUV only: the surface holds just the YUV rows */
			staged = cm_stage_surface(src, stacked + plane, out_linesize, cx, cy);
		}
		src->flags = flags;
		PROFILE_END(PROF_STAGE_SURFACE);
	} else {
		blog(LOG_ERROR, "[obs-color-monitor] cm_render_target: %s (%s)", scope_strerror(st), scope_last_error(gpu));
	}
	bfree(stacked);
	return staged;
}

void cm_drain(struct cm_source *src)
{
	/* every published frame has been through its callback (a source without a running thread has nobody to wait for) */
	pthread_mutex_lock(&src->pipeline_mutex);
	while (src->pipeline_thread_running &&
	       atomic_load_explicit(&src->done, memory_order_acquire) != atomic_load_explicit(&src->staged, memory_order_acquire))
		pthread_cond_wait(&src->pipeline_cond, &src->pipeline_mutex);
	pthread_mutex_unlock(&src->pipeline_mutex);
}
