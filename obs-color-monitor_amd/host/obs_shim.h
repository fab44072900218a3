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
};

enum { LOG_ERROR = 100, LOG_WARNING = 200, LOG_INFO = 300, LOG_DEBUG = 400 };
void blog(int level, const char *format, ...);
void *bzalloc(size_t size);
void bfree(void *ptr);

long long obs_data_get_int(obs_data_t *data, const char *name);
bool obs_data_get_bool(obs_data_t *data, const char *name);
void obs_data_set_default_int(obs_data_t *data, const char *name, long long val);
void obs_data_set_default_bool(obs_data_t *data, const char *name, bool val);
void obs_register_source(const struct obs_source_info *info);

#define UNUSED_PARAMETER(x) ((void)(x))

#ifdef __cplusplus
}
#endif
#endif /* HAVE_LIBOBS */
#endif
