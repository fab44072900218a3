/*
 * scope_common.h -- host-side mirror of the reference's capture pipeline boundary
 * (src/common.h, src/common.c), reduced to what the scope accumulation path uses.
 *
 * Same names, same argument meaning, same error behaviour as the reference, so that the scope
 * sources in this directory read like the reference's and a maintainer can move the GPU calls into
 * the real tree one function at a time.
 */
#ifndef SCOPE_COMMON_H
#define SCOPE_COMMON_H

#include <pthread.h>
#include <stdatomic.h>

#include "../../include/scope_hip.h"
#include "obs_shim.h"

#ifdef __cplusplus
extern "C" {
#endif

/* src/common.h:24-30 -- layout-identical to struct scope_surface_data of the C-ABI */
struct cm_surface_data {
	uint8_t *rgb_data, *yuv_data;
	uint32_t linesize, width, height;
	int colorspace;
	void *tex; /* gs_texture_t * in the reference; unused on this path */
};

/* src/common.h:32 */
typedef void (*cm_surface_cb_t)(void *data, struct cm_surface_data *surface_data);

/* src/common.h:89-92 */
#define CM_FLAG_CONVERT_RGB 1
#define CM_FLAG_CONVERT_YUV 2
#define CM_FLAG_RAW_TEXTURE 4
#define CM_FLAG_ROI 8

#define CM_SURFACE_QUEUE_SIZE 3 /* src/common.h:46 */

/* The reference's profiler zones, same names (src/common.c:10-21, src/vectorscope.c:10-20, src/waveform.c:8-18,
 * src/histogram.c:10-19, src/roi.c:11-18): a build with ENABLE_PROFILE shows the GPU calls in the very zones the CPU loops
 * used to fill, so before/after traces of OBS's profiler line up.  The stand-alone host builds with ENABLE_PROFILE and a
 * counting profile_start/profile_end (fake_obs.c). */
#ifdef ENABLE_PROFILE
#define PROFILE_START(x) profile_start(x)
#define PROFILE_END(x) profile_end(x)
#else
#define PROFILE_START(x)
#define PROFILE_END(x)
#endif
#define PROF_PIPELINE_THREAD "cm_pipeline_thread_loop"
#define PROF_RENDER_TARGET "render_target"
#define PROF_CONVERT_YUV "convert_yuv"
#define PROF_STAGE_SURFACE "stage_surface"
#define PROF_STAGESURFACE_MAP "stage_surface_map"
#define PROF_VSS_RENDER "vss_render"
#define PROF_DRAW_VECTORSCOPE "draw_vectorscope"
#define PROF_WVS_RENDER "wvs_render"
#define PROF_DRAW_WAVEFORM "draw_waveform"
#define PROF_HIS_RENDER "his_render"
#define PROF_DRAW_HISTOGRAM "draw_histogram"
#define PROF_ROI_RENDER "roi_render"
#define PROF_DRAW "draw"
#define PROF_DRAW_GRATICULE "graticule"

/* One staged surface of the ring: what gs_stage_texture + gs_stagesurface_map hand to the
 * pipeline thread in the reference (src/common.h:34-44, src/common.c:343-364). */
struct cm_surface_queue_item {
	uint8_t *video_data; /* stacked BGRA8 planes: RGB rows first (if any), then YUV rows */
	size_t capacity;
	uint32_t linesize;
	uint32_t width, height, sheight;
	uint32_t flags;
	int colorspace;
	cm_surface_cb_t cb;
	void *cb_data;
};

struct roi_source;

/* src/common.h:48-87, minus everything that only exists to talk to libobs graphics */
struct cm_source {
	struct cm_surface_queue_item queue[CM_SURFACE_QUEUE_SIZE];
	/* The ring is this build's own: a single-producer single-consumer SEQUENCE ring.  Frame k lives in slot k % CM_SURFACE_QUEUE_SIZE;
	 * `staged` counts the frames the producer (the graphics thread) has published, `taken` the frames the consumer (the
	 * source's "color-monitor" thread) has started on, `done` the frames whose callback has returned -- three monotonic C11
	 * atomics: the producer publishes with a release store of `staged` behind the slot's bytes, the consumer acquires it; the
	 * consumer's release stores of `taken` / `done` are what the producer's "am I a ring behind?" test and cm_drain acquire.
	 * The mutex and the condition variable are only a doorbell for sleeping (nobody's data is ordered by them).
	 * What is kept of the reference's ring (src/common.h:46,54-55, src/common.c:260-268): three slots, and a frame is DROPPED,
	 * not queued, when the consumer is a full ring behind (staged - taken == 2: slot (taken - 1) % 3 may still be read).  What is
	 * not: its three `volatile` indices read bare, and the slot held back until the NEXT frame is staged (i_staging_queue: in
	 * the reference that slot's GPU -> host copy is still in flight; here a staged slot's bytes are there when it is published,
	 * so the consumer takes the newest frame at once -- one frame less latency, and the last frame of a stream is not left
	 * waiting for a successor). */
	_Atomic uint64_t staged, taken, done;

	pthread_t pipeline_thread;
	pthread_mutex_t pipeline_mutex; /* the doorbell: sleeping and waking only */
	pthread_cond_t pipeline_cond;
	bool pipeline_thread_running; /* start / stop / drain: the thread that ticks the source (the graphics thread) */
	_Atomic bool request_exit;

	cm_surface_cb_t callback;
	void *callback_data;

	struct roi_source *roi;
	int x0, x1, y0, y1; /* ROI rectangle in capture pixels (src/common.h:62); used when CM_FLAG_ROI */

	int target_scale;
	int colorspace;
	uint32_t flags;
	bool bypass;

	obs_source_t *self, *target; /* the scope's own source and the one it watches (a weak reference in the reference) */
	bool enumerating;

	uint64_t frames_dropped, frames_processed;
	uint64_t plane_uploads, plane_reuses; /* host->HBM uploads done / saved by the per-surface cache */
};

void cm_create(struct cm_source *src, obs_data_t *settings, obs_source_t *source); /* src/common.c:33-52 */
void cm_destroy(struct cm_source *src);                                            /* src/common.c:54-79 */
void cm_update(struct cm_source *src, obs_data_t *settings);                       /* src/common.c:81-98 */
void cm_get_defaults(obs_data_t *settings);
void cm_get_properties(struct cm_source *src, obs_properties_t *props);                     /* src/common.c:112-139 */
void cm_enum_sources(void *data, obs_source_enum_proc_t enum_callback, void *param);        /* src/common.c:98-110 */
void cm_request(struct cm_source *src, cm_surface_cb_t callback, void *data);      /* src/common.c:615-620 */
void cm_tick(void *data, float unused);                                            /* src/common.c:575-595 */

/* Producer half of cm_render_target (src/common.c:223-333) for a host that already holds the
 * captured planes stacked vertically (RGB rows, then YUV rows, `height` rows each): stages them
 * into the next ring slot -- cropped to the ROI rectangle when CM_FLAG_ROI is set and the rectangle
 * is valid (src/common.c:272-282) -- and wakes the pipeline thread.  Returns false when the
 * consumer is a full ring behind and the frame was dropped (src/common.c:260-268). */
bool cm_stage_surface(struct cm_source *src, const uint8_t *video_data, uint32_t linesize, uint32_t width,
		      uint32_t height);
/* The whole producer half of cm_render_target (src/common.c:223-333) for a host that holds the target's
 * full-size RGB frame: down-scale by target_scale (:249-250), crop to the ROI rectangle given in scaled
 * pixels (:272-282), convert the YUV rows (render_rgb_yuv, :170-221) -- all on the GPU through the C-ABI
 * (scope_stage_region, scope_convert_rgb_yuv) -- and stage the result.  Returns false when nothing was
 * staged: the scaled size is zero (:251-254), the consumer is a ring behind (:260-268), or the GPU failed. */
bool cm_render_target(struct cm_source *src, const uint8_t *target_rgb, uint32_t linesize, uint32_t target_width,
		      uint32_t target_height);
/* Blocks until every staged surface has been through the callback (test/bench helper). */
void cm_drain(struct cm_source *src);

/* The GPU contexts of this module: cm_gpu() for the pipeline threads' scope calls, cm_gpu_render() for the render thread's
 * display passes (a context is locked per call, and from scope_surface_begin to _end: the render thread must not queue up
 * behind a surface upload). */
scope_ctx *cm_gpu(void);
scope_ctx *cm_gpu_render(void);
int cm_gpu_open(int device);
void cm_gpu_close(void);

#ifdef __cplusplus
}
#endif
#endif
