/*
 * obs_shim.h -- the few libobs names the scope sources touch on the hot path.
 *
 * With -DHAVE_LIBOBS the real <obs-module.h> is used and this file declares nothing: that is the
 * build a maintainer does inside the obs-color-monitor tree.  Without it (this repository: no
 * libobs headers exist in the image) the declarations below are a clean-room minimum written from
 * how the reference *uses* them (src/vectorscope.c:484-519, src/waveform.c:402-417,
 * src/histogram.c:580-595, src/roi.c:572-593, src/plugin-main.c:44-56); fake_obs.c implements them
 * for the stand-alone host.  `struct obs_source_info` here carries only the members those tables
 * set -- it is NOT layout compatible with libobs and is never handed to a real OBS.
 */
#ifndef SCOPE_OBS_SHIM_H
#define SCOPE_OBS_SHIM_H

#ifdef HAVE_LIBOBS
#include <obs-module.h>
#else

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct obs_data obs_data_t;
typedef struct obs_source obs_source_t;
typedef struct gs_effect gs_effect_t;
typedef struct obs_properties obs_properties_t;
typedef struct obs_property obs_property_t;
struct obs_mouse_event; /* only ever passed through (vss_mouse_wheel ignores it, src/vectorscope.c:473-476) */
typedef void (*obs_source_enum_proc_t)(obs_source_t *parent, obs_source_t *child, void *param);

enum obs_source_type { OBS_SOURCE_TYPE_INPUT, OBS_SOURCE_TYPE_FILTER };
#define OBS_SOURCE_VIDEO (1 << 0)
#define OBS_SOURCE_CUSTOM_DRAW (1 << 3)
#define OBS_SOURCE_INTERACTION (1 << 5)
#define OBS_SOURCE_CAP_OBSOLETE (1 << 9)

struct obs_source_info {
	const char *id;
	uint32_t version;
	enum obs_source_type type;
	uint32_t output_flags;
	const char *(*get_name)(void *type_data);
	void *(*create)(obs_data_t *settings, obs_source_t *source);
	void (*destroy)(void *data);
	uint32_t (*get_width)(void *data);
	uint32_t (*get_height)(void *data);
	void (*get_defaults)(obs_data_t *settings);
	void (*update)(void *data, obs_data_t *settings);
	void (*video_tick)(void *data, float seconds);
	void (*video_render)(void *data, gs_effect_t *effect);
	/* the rest of what the reference's tables set (src/vectorscope.c:484-519, src/waveform.c:402-417,
	 * src/histogram.c:580-595, src/roi.c:572-593); appended so that the members above keep their offsets */
	obs_properties_t *(*get_properties)(void *data);
	void (*enum_active_sources)(void *data, obs_source_enum_proc_t enum_callback, void *param);
	void (*mouse_wheel)(void *data, const struct obs_mouse_event *event, int x_delta, int y_delta);
};

enum { LOG_ERROR = 100, LOG_WARNING = 200, LOG_INFO = 300, LOG_DEBUG = 400 };
void blog(int level, const char *format, ...);
void *bzalloc(size_t size);
void bfree(void *ptr);

long long obs_data_get_int(obs_data_t *data, const char *name);
double obs_data_get_double(obs_data_t *data, const char *name);
bool obs_data_get_bool(obs_data_t *data, const char *name);
void obs_data_set_default_int(obs_data_t *data, const char *name, long long val);
void obs_data_set_default_double(obs_data_t *data, const char *name, double val);
void obs_data_set_default_bool(obs_data_t *data, const char *name, bool val);
void obs_register_source(const struct obs_source_info *info);

/* the property calls the get_properties callbacks of this path make */
enum obs_combo_type { OBS_COMBO_TYPE_LIST = 2 };
enum obs_combo_format { OBS_COMBO_FORMAT_INT = 1 };
obs_properties_t *obs_properties_create(void);
void obs_properties_destroy(obs_properties_t *props);
obs_property_t *obs_properties_add_int(obs_properties_t *props, const char *name, const char *description, int min, int max,
				       int step);
obs_property_t *obs_properties_add_bool(obs_properties_t *props, const char *name, const char *description);
obs_property_t *obs_properties_add_list(obs_properties_t *props, const char *name, const char *description,
					enum obs_combo_type type, enum obs_combo_format format);
size_t obs_property_list_add_int(obs_property_t *p, const char *name, long long val);
const char *obs_module_text(const char *lookup);

/* libobs' profiler (util/profiler.h), which the reference's ENABLE_PROFILE build calls with the zone names mirrored in
 * scope_common.h */
void profile_start(const char *name);
void profile_end(const char *name);

#define UNUSED_PARAMETER(x) ((void)(x))

#ifdef __cplusplus
}
#endif
#endif /* HAVE_LIBOBS */
#endif
