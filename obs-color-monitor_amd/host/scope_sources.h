/*
 * scope_sources.h -- the three scope sources and the ROI fan-out, hot-path members only.
 * Mirrors struct vss_source (src/vectorscope.c:36-66), struct wvs_source (src/waveform.c:31-49),
 * struct his_source (src/histogram.c:37-60) and struct roi_source (src/roi.h:10-25).
 */
#ifndef SCOPE_SOURCES_H
#define SCOPE_SOURCES_H

#include "scope_common.h"

#ifdef __cplusplus
extern "C" {
#endif

#define VS_SIZE 256
#define WV_SIZE 256
#define HI_SIZE 256
#define COMP_RGB 0x07
#define COMP_Y 0x20
#define COMP_UV 0x50
#define COMP_YUV (COMP_Y | COMP_UV)

/* What video_render produced last: the scope picture (RGBA8) the display shader would have drawn. */
struct scope_picture {
	uint8_t *rgba;
	size_t capacity;
	uint32_t width, height;
};

/* The double buffer's hand-off.  The reference flips a `volatile int w_tex_buf` on the worker thread and reads `^ 1` on the render
 * thread with no ordering at all (src/vectorscope.c:48,264,410; src/waveform.c:40-41,288,399; src/histogram.c:47,449,563): the
 * render thread may still be reading a table when the worker, one flip later, writes into it.  Here the same two buffers and
 * the same flip, ordered: the worker fills tex_buf[w_tex_buf] holding nothing and takes tex_mutex only to FLIP; the render side
 * holds tex_mutex while it reads tex_buf[w_tex_buf ^ 1].  A buffer is written between two flips and read only in sections that
 * lie before the first or behind the second, so writer and reader are never on one buffer, the worker waits for one table
 * upload at most, and ThreadSanitizer has nothing to report (tests/emul/host_tsan.c runs the three threads flat out). */
#define TEX_FLIP(src)                                   \
	do {                                            \
		pthread_mutex_lock(&(src)->tex_mutex);   \
		(src)->w_tex_buf ^= 1;                  \
		pthread_mutex_unlock(&(src)->tex_mutex); \
	} while (0)

struct vss_source {
	struct cm_source cm;
	uint8_t *tex_buf[2];
	int tex_cs[2];
	int w_tex_buf; /* written by the worker under tex_mutex, read by the render side under it */
	pthread_mutex_t tex_mutex;
	int intensity;  /* src/vectorscope.c:55,129-131 */
	int color_type; /* 0 white, 1 UV (src/vectorscope.c:36-39) */
	float zoom;     /* 1.0 on creation (src/vectorscope.c:80); the mouse wheel changes it (:476-482) */
	int graticule;  /* 0 none; & 3: 1 amber, 2 green; | 256: skin tone line as the I/Q cross (src/vectorscope.c:24-25,135-145) */
	uint32_t graticule_color;          /* 0xAARRGGBB as handed to gs_effect_set_color */
	uint32_t graticule_skintone_color; /* 0xBBGGRR */
	struct scope_picture pic;
};
/* vss_mouse_wheel (src/vectorscope.c:473-483): zoom *= exp(y_delta * 5e-4), never below 1. */
void vss_mouse_wheel(void *data, const struct obs_mouse_event *event, int x_delta, int y_delta);

struct wvs_source {
	struct cm_source cm;
	uint8_t *tex_buf[2];
	uint32_t tex_buf_width[2];
	int w_tex_buf;
	pthread_mutex_t tex_mutex;
	uint32_t components;
	int display;   /* DISP_OVERLAY / DISP_STACK / DISP_PARADE, src/waveform.c:22-24 */
	int intensity; /* src/waveform.c:104-106 */
	int graticule_lines; /* src/waveform.c:108: 0 none, else the scale is cut into this many parts */
	struct scope_picture pic;
};

struct his_source {
	struct cm_source cm;
	uint8_t *tex_buf[2];
	uint32_t hi_max[2][3];
	int w_tex_buf;
	pthread_mutex_t tex_mutex;
	uint32_t components;
	int level_fixed_value;
	int level_ratio_value;
	bool logscale;
	int display;
	int level_height; /* src/histogram.c:53,123 */
	int graticule_vertical_lines;    /* src/histogram.c:57,158 */
	float graticule_horizontal_step; /* :58,140-151: in pixels (level_mode 1) or per cent (level_mode 2) */
	struct scope_picture pic;
};

#define ROI_MAX_SOURCES 16
struct roi_source {
	struct cm_source cm;
	int n_interleave, i_interleave; /* src/roi.h:13-14: process one frame in n_interleave + 1 */
	bool interleave_rendered;
	pthread_mutex_t sources_mutex;
	struct cm_source *sources[ROI_MAX_SOURCES]; /* DARRAY(struct cm_source *) in the reference */
	size_t n_sources;
};

extern const struct obs_source_info colormonitor_vectorscope_v1; /* id "vectorscope_source", obsolete (src/vectorscope.c:484-500) */
extern const struct obs_source_info colormonitor_vectorscope_v2; /* id "vectorscope_source", version 2 */
extern const struct obs_source_info colormonitor_waveform;       /* id "waveform_source" */
extern const struct obs_source_info colormonitor_histogram;      /* id "histogram_source" */
extern const struct obs_source_info colormonitor_roi;            /* id "colormonitor_roi" */

/* The callbacks the pipeline thread calls (cm_surface_cb_t): same names as the reference's statics. */
void vss_surface_cb(void *data, struct cm_surface_data *surface_data); /* src/vectorscope.c:248-265 */
void wvs_surface_cb(void *data, struct cm_surface_data *surface_data); /* src/waveform.c:272-289 */
void his_surface_cb(void *data, struct cm_surface_data *surface_data); /* src/histogram.c:432-450 */
void roi_surface_cb(void *data, struct cm_surface_data *surface_data); /* src/roi.c:329-341 */

void roi_register_source(struct roi_source *src, struct cm_source *cm);   /* src/roi.c:315-320 */
void roi_unregister_source(struct roi_source *src, struct cm_source *cm); /* src/roi.c:322-327 */
void roi_update_flags(struct roi_source *src);                            /* flag merge of roi_tick, src/roi.c:534-540 */
/* src/roi.c:266-277 with the host's captured planes in place of the OBS render: stages the frame
 * unless this tick is skipped by interleaving.  Returns what the reference returns. */
bool roi_target_render_full(struct roi_source *src, const uint8_t *target_rgb, uint32_t linesize, uint32_t target_width,
			    uint32_t target_height);
bool roi_target_render(struct roi_source *src, const uint8_t *video_data, uint32_t linesize, uint32_t width,
		       uint32_t height);

bool obs_module_load(void); /* src/plugin-main.c:58-108, the four sources of this path */
void obs_module_unload(void);

#ifdef __cplusplus
}
#endif
#endif
