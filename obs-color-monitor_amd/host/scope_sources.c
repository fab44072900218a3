/*
 * scope_sources.c -- vectorscope / waveform / histogram sources and the ROI fan-out with the
 * per-pixel loops replaced by calls into libscope_hip.so.
 *
 * Everything around the three scope_* calls follows the reference: the NULL-plane guards, the lazy
 * double buffer, the flip of w_tex_buf by the worker thread, the silent skip on failure (the
 * callbacks are void; a failed frame leaves the previous table on screen).
 */
#include "scope_sources.h"
#include "scope_graticule.h"

#include <math.h>
#include <string.h>

/* -- shared -------------------------------------------------------------------------------------- */

static const struct scope_surface_data *as_scope(const struct cm_surface_data *sd)
{
	/* field-for-field the same struct (include/scope_hip.h) */
	return (const struct scope_surface_data *)sd;
}

static void log_failure(const char *what, int st)
{
	if (st != SCOPE_OK)
		blog(LOG_ERROR, "[obs-color-monitor] %s failed: %s", what, scope_strerror(st));
}

/* The display shaders (D1-D3) run as GPU post-passes; the picture lands where the plugin would have
 * drawn it.  Silent skip on failure, like every render path of the reference. */
static uint8_t *picture_reserve(struct scope_picture *pic, uint32_t w, uint32_t h)
{
	const size_t need = (size_t)w * h * 4;
	if (pic->capacity < need) {
		bfree(pic->rgba);
		pic->rgba = bzalloc(need);
		pic->capacity = pic->rgba ? need : 0;
	}
	pic->width = w;
	pic->height = h;
	return pic->rgba;
}

/* -- vectorscope ------------------------------------------------------------------------------------ */

static const char *vss_get_name(void *unused)
{
	UNUSED_PARAMETER(unused);
	return "Vectorscope";
}

static void vss_update(void *data, obs_data_t *settings)
{
	struct vss_source *src = data;
	cm_update(&src->cm, settings);
	src->intensity = (int)obs_data_get_int(settings, "intensity"); /* src/vectorscope.c:129-133 */
	if (src->intensity < 1)
		src->intensity = 1;
	src->color_type = (int)obs_data_get_int(settings, "color_type");
	src->graticule = (int)obs_data_get_int(settings, "graticule"); /* :135-145 */
	switch (src->graticule & (int)SCOPE_GRATICULE_COLOR_MASK) {
	case 1:
		src->graticule_color = 0x80FFBF00; /* amber */
		break;
	case 2:
		src->graticule_color = 0x8000FF00; /* green */
		break;
	}
	src->graticule_skintone_color = (uint32_t)obs_data_get_int(settings, "graticule_skintone_color") & 0xFFFFFF; /* :147 */
}

static void *vss_create(obs_data_t *settings, obs_source_t *source)
{
	struct vss_source *src = bzalloc(sizeof(struct vss_source));
	src->cm.flags = CM_FLAG_CONVERT_YUV; /* src/vectorscope.c:79 */
	cm_create(&src->cm, settings, source);
	cm_request(&src->cm, vss_surface_cb, src);
	pthread_mutex_init(&src->tex_mutex, NULL);
	src->zoom = 1.0f; /* src/vectorscope.c:80 */
	vss_update(src, settings);
	return src;
}

static void vss_destroy(void *data)
{
	struct vss_source *src = data;
	cm_destroy(&src->cm);
	bfree(src->tex_buf[0]); /* src/vectorscope.c:119-120 */
	bfree(src->tex_buf[1]);
	bfree(src->pic.rgba);
	pthread_mutex_destroy(&src->tex_mutex);
	bfree(src);
}

static void vss_get_defaults_v1(obs_data_t *settings)
{
	cm_get_defaults(settings);
	obs_data_set_default_int(settings, "intensity", 25);          /* src/vectorscope.c:158 */
	obs_data_set_default_int(settings, "graticule", 1 | SCOPE_GRATICULE_IQ); /* :159 */
	obs_data_set_default_int(settings, "graticule_skintone_color", SCOPE_SKIN_TONE_LINE); /* :160 */
}

static void vss_get_defaults(obs_data_t *settings)
{
	vss_get_defaults_v1(settings);
	obs_data_set_default_int(settings, "color_type", 1); /* color_type_uv, :166; version 1 sources stay white */
}

/* src/vectorscope.c:169-196 minus the colour picker of the skin-tone line */
static obs_properties_t *vss_get_properties(void *data)
{
	struct vss_source *src = data;
	obs_properties_t *props = obs_properties_create();
	cm_get_properties(&src->cm, props);
	obs_properties_add_int(props, "intensity", obs_module_text("Intensity"), 1, 255, 1);
	obs_property_t *prop = obs_properties_add_list(props, "color_type", obs_module_text("VS.Prop.ColorType"),
						       OBS_COMBO_TYPE_LIST, OBS_COMBO_FORMAT_INT);
	obs_property_list_add_int(prop, obs_module_text("VS.Prop.ColorType.White"), 0);
	obs_property_list_add_int(prop, obs_module_text("VS.Prop.ColorType.UV"), 1);
	prop = obs_properties_add_list(props, "graticule", obs_module_text("Graticule"), OBS_COMBO_TYPE_LIST,
				       OBS_COMBO_FORMAT_INT);
	obs_property_list_add_int(prop, obs_module_text("None"), 0);
	obs_property_list_add_int(prop, obs_module_text("Amber"), 1);
	obs_property_list_add_int(prop, obs_module_text("Amber, IQ"), 1 + SCOPE_GRATICULE_IQ);
	obs_property_list_add_int(prop, obs_module_text("Green"), 2);
	obs_property_list_add_int(prop, obs_module_text("Green, IQ"), 2 + SCOPE_GRATICULE_IQ);
	return props;
}

void vss_mouse_wheel(void *data, const struct obs_mouse_event *event, int x_delta, int y_delta)
{
	UNUSED_PARAMETER(event);
	UNUSED_PARAMETER(x_delta);
	struct vss_source *src = data;
	src->zoom *= expf((float)y_delta * 5e-4f); /* src/vectorscope.c:479 */
	if (src->zoom < 1.0f)
		src->zoom = 1.0f;
}

/* src/vectorscope.c:382-471: upload, zoom transform and PSDrawBare become one GPU pass, the label quads and the bracket /
 * skin-tone lines a second one (scope_display_graticule; under a zoom they are scaled about 127.5 like the sprite, :395-408) */
static void vss_render_locked(struct vss_source *src)
{
	const int r_tex_buf = src->w_tex_buf ^ 1;
	if (!src->tex_buf[r_tex_buf])
		return; /* :411 */
	uint8_t *rgba = picture_reserve(&src->pic, VS_SIZE, VS_SIZE);
	if (!rgba)
		return;
	PROFILE_START(PROF_VSS_RENDER);
	PROFILE_START(PROF_DRAW);
	const int st = scope_display_vectorscope_zoom(cm_gpu_render(), src->tex_buf[r_tex_buf], src->intensity, src->color_type,
						      src->tex_cs[r_tex_buf], src->zoom, 0, rgba);
	PROFILE_END(PROF_DRAW);
	log_failure("scope_display_vectorscope", st);
	PROFILE_START(PROF_DRAW_GRATICULE);
	if (st == SCOPE_OK && src->graticule) { /* :447-464 */
		struct scope_label labels[SCOPE_VSS_LABELS];
		struct scope_vertex lines[SCOPE_VSS_LINE_VERTICES_MAX];
		const uint32_t n = scope_graticule_vectorscope(src->tex_cs[r_tex_buf], (uint32_t)src->graticule,
							       src->graticule_skintone_color, labels, lines);
		if (src->zoom > 1.01f) { /* the matrix pushed at :399-404: translate by ofst, scale by zoom */
			const float ofst = 127.5f * (1.0f - src->zoom);
			for (uint32_t i = 0; i < SCOPE_VSS_LABELS; i++) {
				labels[i].x = labels[i].x * src->zoom + ofst;
				labels[i].y = labels[i].y * src->zoom + ofst;
				labels[i].w *= src->zoom;
				labels[i].h *= src->zoom;
			}
			for (uint32_t i = 0; i < n; i++) {
				lines[i].x = lines[i].x * src->zoom + ofst;
				lines[i].y = lines[i].y * src->zoom + ofst;
			}
		}
		log_failure("scope_display_graticule",
			    scope_display_graticule(cm_gpu_render(), rgba, VS_SIZE, VS_SIZE, (const struct scope_graticule_label *)labels,
						    SCOPE_VSS_LABELS, (const struct scope_graticule_vertex *)lines, n, src->graticule_color, 0));
	}
	PROFILE_END(PROF_DRAW_GRATICULE);
	PROFILE_END(PROF_VSS_RENDER);
}

static void vss_render(void *data, gs_effect_t *effect)
{
	UNUSED_PARAMETER(effect);
	struct vss_source *src = data;
	pthread_mutex_lock(&src->tex_mutex); /* the table being drawn is not the worker's until this section ends (scope_sources.h) */
	vss_render_locked(src);
	pthread_mutex_unlock(&src->tex_mutex);
}

static uint32_t vss_get_width(void *data)
{
	UNUSED_PARAMETER(data);
	return VS_SIZE; /* src/vectorscope.c:205-215 */
}

void vss_surface_cb(void *data, struct cm_surface_data *surface_data)
{
	struct vss_source *src = data;

	if (!surface_data->yuv_data)
		return;

	if (!src->tex_buf[src->w_tex_buf])
		src->tex_buf[src->w_tex_buf] = bzalloc(VS_SIZE * VS_SIZE);

	/* was: vss_draw_vectorscope(src->tex_buf[src->w_tex_buf], surface_data), in the same profiler zone */
	PROFILE_START(PROF_DRAW_VECTORSCOPE);
	const int st = scope_vectorscope(cm_gpu(), as_scope(surface_data), 0, src->tex_buf[src->w_tex_buf]);
	PROFILE_END(PROF_DRAW_VECTORSCOPE);
	if (st != SCOPE_OK) {
		log_failure("scope_vectorscope", st);
		return; /* keep showing the previous table */
	}

	src->tex_cs[src->w_tex_buf] = surface_data->colorspace;
	TEX_FLIP(src);
}

static void scope_render_nothing(void *data, gs_effect_t *effect)
{
	/* the display shaders (data/vectorscope|waveform|histogram.effect) stay in the plugin;
	 * the stand-alone host has no graphics context to draw with */
	UNUSED_PARAMETER(data);
	UNUSED_PARAMETER(effect);
}

/* src/vectorscope.c:484-500: settings saved by plugin versions before "color_type" existed keep the white trace */
const struct obs_source_info colormonitor_vectorscope_v1 = {
	.id = "vectorscope_source",
	.type = OBS_SOURCE_TYPE_INPUT,
	.output_flags = OBS_SOURCE_VIDEO | OBS_SOURCE_CUSTOM_DRAW | OBS_SOURCE_INTERACTION | OBS_SOURCE_CAP_OBSOLETE,
	.get_name = vss_get_name,
	.create = vss_create,
	.destroy = vss_destroy,
	.update = vss_update,
	.get_defaults = vss_get_defaults_v1,
	.get_properties = vss_get_properties,
	.get_width = vss_get_width,
	.get_height = vss_get_width,
	.enum_active_sources = cm_enum_sources,
	.video_render = vss_render,
	.video_tick = cm_tick,
	.mouse_wheel = vss_mouse_wheel,
};

const struct obs_source_info colormonitor_vectorscope_v2 = {
	.id = "vectorscope_source",
	.version = 2,
	.type = OBS_SOURCE_TYPE_INPUT,
	.output_flags = OBS_SOURCE_VIDEO | OBS_SOURCE_CUSTOM_DRAW | OBS_SOURCE_INTERACTION,
	.get_name = vss_get_name,
	.create = vss_create,
	.destroy = vss_destroy,
	.update = vss_update,
	.get_defaults = vss_get_defaults,
	.get_properties = vss_get_properties,
	.get_width = vss_get_width,
	.get_height = vss_get_width,
	.enum_active_sources = cm_enum_sources,
	.video_render = vss_render,
	.video_tick = cm_tick,
	.mouse_wheel = vss_mouse_wheel,
};

/* -- waveform ----------------------------------------------------------------------------------------- */

static const char *wvs_get_name(void *unused)
{
	UNUSED_PARAMETER(unused);
	return "Waveform";
}

static void wvs_update(void *data, obs_data_t *settings)
{
	struct wvs_source *src = data;
	cm_update(&src->cm, settings);
	src->components = (uint32_t)obs_data_get_int(settings, "components");
	/* src/waveform.c:100-102 */
	src->cm.flags = ((src->components & COMP_RGB) ? CM_FLAG_CONVERT_RGB : 0) |
			((src->components & COMP_YUV) ? CM_FLAG_CONVERT_YUV : 0);
	src->display = (int)obs_data_get_int(settings, "display");
	src->intensity = (int)obs_data_get_int(settings, "intensity");
	if (src->intensity < 1)
		src->intensity = 1;
	src->graticule_lines = (int)obs_data_get_int(settings, "graticule_lines"); /* src/waveform.c:108 */
}

static void *wvs_create(obs_data_t *settings, obs_source_t *source)
{
	struct wvs_source *src = bzalloc(sizeof(struct wvs_source));
	cm_create(&src->cm, settings, source);
	cm_request(&src->cm, wvs_surface_cb, src);
	pthread_mutex_init(&src->tex_mutex, NULL);
	wvs_update(src, settings);
	return src;
}

static void wvs_destroy(void *data)
{
	struct wvs_source *src = data;
	cm_destroy(&src->cm);
	bfree(src->tex_buf[0]); /* src/waveform.c:87-88 */
	bfree(src->tex_buf[1]);
	bfree(src->pic.rgba);
	pthread_mutex_destroy(&src->tex_mutex);
	bfree(src);
}

/* src/waveform.c:133-170 */
static obs_properties_t *wvs_get_properties(void *data)
{
	struct wvs_source *src = data;
	obs_properties_t *props = obs_properties_create();
	cm_get_properties(&src->cm, props);
	obs_property_t *prop = obs_properties_add_list(props, "display", obs_module_text("Display"), OBS_COMBO_TYPE_LIST,
						       OBS_COMBO_FORMAT_INT);
	obs_property_list_add_int(prop, obs_module_text("Overlay"), 0);
	obs_property_list_add_int(prop, obs_module_text("Stack"), 1);
	obs_property_list_add_int(prop, obs_module_text("Parade"), 2);
	prop = obs_properties_add_list(props, "components", obs_module_text("Components"), OBS_COMBO_TYPE_LIST,
				       OBS_COMBO_FORMAT_INT);
	obs_property_list_add_int(prop, obs_module_text("RGB"), COMP_RGB);
	obs_property_list_add_int(prop, obs_module_text("Luma"), COMP_Y);
	obs_property_list_add_int(prop, obs_module_text("Chroma"), COMP_UV);
	obs_property_list_add_int(prop, obs_module_text("YUV"), COMP_YUV);
	obs_properties_add_int(props, "intensity", obs_module_text("Intensity"), 1, 255, 1);
	prop = obs_properties_add_list(props, "graticule_lines", obs_module_text("Graticule"), OBS_COMBO_TYPE_LIST,
				       OBS_COMBO_FORMAT_INT);
	static const int steps[] = {0, 1, 2, 4, 5, 10}; /* none, every 100 / 50 / 25 / 20 / 10 % */
	for (size_t i = 0; i < sizeof steps / sizeof steps[0]; i++)
		obs_property_list_add_int(prop, "Graticule.Step", steps[i]);
	return props;
}

static void wvs_get_defaults(obs_data_t *settings)
{
	cm_get_defaults(settings);
	obs_data_set_default_int(settings, "components", COMP_RGB); /* src/waveform.c:115 */
	obs_data_set_default_int(settings, "display", 0);
	obs_data_set_default_int(settings, "intensity", 51); /* src/waveform.c:114 */
	obs_data_set_default_int(settings, "graticule_lines", 5); /* :116 */
}

/* src/waveform.c:334-392 minus graticule: upload + DrawOverlay/Stack/Parade become one GPU pass */
static void wvs_render_locked(struct wvs_source *src)
{
	const int r = src->w_tex_buf ^ 1;
	if (!src->tex_buf[r] || !src->tex_buf_width[r])
		return;
	uint32_t w, h;
	scope_display_size(src->tex_buf_width[r], WV_SIZE, src->components, src->display, &w, &h);
	uint8_t *rgba = picture_reserve(&src->pic, w, h);
	if (!rgba || !w || !h)
		return;
	PROFILE_START(PROF_WVS_RENDER);
	PROFILE_START(PROF_DRAW);
	const int st = scope_display_waveform(cm_gpu_render(), src->tex_buf[r], src->tex_buf_width[r], src->components, src->display,
					      src->intensity, 0, rgba);
	PROFILE_END(PROF_DRAW);
	log_failure("scope_display_waveform", st);
	PROFILE_START(PROF_DRAW_GRATICULE);
	if (st == SCOPE_OK && src->graticule_lines > 0) { /* src/waveform.c:381-389 */
		uint32_t nc = 0;
		for (uint32_t c = src->components & 0x77u; c; c >>= 1)
			nc += c & 1u;
		struct scope_vertex lines[2 * 3 * 64];
		const uint32_t max = sizeof lines / sizeof lines[0];
		uint32_t n = scope_graticule_waveform(src->graticule_lines, src->display, nc, src->tex_buf_width[r], lines, max);
		if (n > max)
			n = max;
		log_failure("scope_display_graticule",
			    scope_display_graticule(cm_gpu_render(), rgba, w, h, NULL, 0, (const struct scope_graticule_vertex *)lines, n,
						    0x80FFBF00 /* amber, :311 */, 0));
	}
	PROFILE_END(PROF_DRAW_GRATICULE);
	PROFILE_END(PROF_WVS_RENDER);
}

static void wvs_render(void *data, gs_effect_t *effect)
{
	UNUSED_PARAMETER(effect);
	struct wvs_source *src = data;
	pthread_mutex_lock(&src->tex_mutex); /* the table being drawn is not the worker's until this section ends (scope_sources.h) */
	wvs_render_locked(src);
	pthread_mutex_unlock(&src->tex_mutex);
}

static uint32_t wvs_get_width(void *data)
{
	struct wvs_source *src = data; /* src/waveform.c:181-189 */
	uint32_t w, h;
	pthread_mutex_lock(&src->tex_mutex);
	const uint32_t tw = src->tex_buf_width[src->w_tex_buf ^ 1];
	pthread_mutex_unlock(&src->tex_mutex);
	scope_display_size(tw, WV_SIZE, src->components, src->display, &w, &h);
	return w;
}

static uint32_t wvs_get_height(void *data)
{
	struct wvs_source *src = data; /* src/waveform.c:191-199 */
	uint32_t w, h;
	pthread_mutex_lock(&src->tex_mutex);
	const uint32_t tw = src->tex_buf_width[src->w_tex_buf ^ 1];
	pthread_mutex_unlock(&src->tex_mutex);
	scope_display_size(tw, WV_SIZE, src->components, src->display, &w, &h);
	return h;
}

/* src/waveform.c:207-218 */
static void ensure_tex_buf_size(struct wvs_source *src, const uint32_t width, int ix)
{
	if (src->tex_buf[ix] && src->tex_buf_width[ix] == width)
		return;
	if (!width)
		return;
	bfree(src->tex_buf[ix]);
	src->tex_buf[ix] = bzalloc((size_t)width * WV_SIZE * 4);
	src->tex_buf_width[ix] = width;
}

void wvs_surface_cb(void *data, struct cm_surface_data *surface_data)
{
	struct wvs_source *src = data;

	if ((src->components & COMP_RGB) && !surface_data->rgb_data)
		return;
	if ((src->components & COMP_YUV) && !surface_data->yuv_data)
		return;
	if (!surface_data->width)
		return;

	ensure_tex_buf_size(src, surface_data->width, src->w_tex_buf);

	/* was: wvs_draw_waveform(src, src->tex_buf[src->w_tex_buf], surface_data), in the same profiler zone */
	PROFILE_START(PROF_DRAW_WAVEFORM);
	const int st = scope_waveform(cm_gpu(), as_scope(surface_data), src->components, 0, src->tex_buf[src->w_tex_buf]);
	PROFILE_END(PROF_DRAW_WAVEFORM);
	if (st != SCOPE_OK) {
		log_failure("scope_waveform", st);
		return;
	}
	TEX_FLIP(src);
}

const struct obs_source_info colormonitor_waveform = {
	.id = "waveform_source",
	.type = OBS_SOURCE_TYPE_INPUT,
	.output_flags = OBS_SOURCE_VIDEO | OBS_SOURCE_CUSTOM_DRAW,
	.get_name = wvs_get_name,
	.create = wvs_create,
	.destroy = wvs_destroy,
	.update = wvs_update,
	.get_defaults = wvs_get_defaults,
	.get_properties = wvs_get_properties,
	.get_width = wvs_get_width,
	.get_height = wvs_get_height,
	.enum_active_sources = cm_enum_sources,
	.video_render = wvs_render,
	.video_tick = cm_tick,
};

/* -- histogram ---------------------------------------------------------------------------------------- */

static const char *his_get_name(void *unused)
{
	UNUSED_PARAMETER(unused);
	return "Histogram";
}

static void his_update(void *data, obs_data_t *settings)
{
	struct his_source *src = data;
	cm_update(&src->cm, settings);
	src->components = (uint32_t)obs_data_get_int(settings, "components");
	src->cm.flags = ((src->components & COMP_RGB) ? CM_FLAG_CONVERT_RGB : 0) |
			((src->components & COMP_YUV) ? CM_FLAG_CONVERT_YUV : 0); /* src/histogram.c:119-121 */
	src->logscale = obs_data_get_bool(settings, "logscale");
	/* src/histogram.c:131-156: level_mode 0 none, 1 pixel count, 2 share of the frame.  Keys, types and conversions are the
	 * reference's, so a settings blob saved by the real plugin reads the same here: the ratio is a double PERCENTAGE kept in
	 * tenths ((int)(v * 10 + 0.5), :146-147), the graticule steps are doubles (:140-141, :149-150). */
	const int level_mode = (int)obs_data_get_int(settings, "level_mode");
	if (level_mode == 0) {
		/* the two level values only: the step of the mode before stays (:132-135) */
		src->level_ratio_value = 0;
		src->level_fixed_value = 0;
	} else if (level_mode == 1) {
		src->level_fixed_value = (int)obs_data_get_int(settings, "level_fixed_value");
		src->graticule_horizontal_step = (float)obs_data_get_double(settings, "graticule_horizontal_step_fixed");
		src->level_ratio_value = 0;
	} else if (level_mode == 2) {
		src->level_ratio_value = (int)(obs_data_get_double(settings, "level_ratio_value") * 10.0 + 0.5);
		src->graticule_horizontal_step = (float)obs_data_get_double(settings, "graticule_horizontal_step_ratio");
		src->level_fixed_value = 0;
	} else {
		/* a bad blob changes nothing and is only reported (:153-155) */
		blog(LOG_ERROR, "histogram: Invalid level_mode %d", level_mode);
	}
	src->display = (int)obs_data_get_int(settings, "display");
	src->level_height = (int)obs_data_get_int(settings, "level_height"); /* src/histogram.c:123 */
	if (src->level_height < 1)
		src->level_height = 1;
	src->graticule_vertical_lines = (int)obs_data_get_int(settings, "graticule_vertical_lines"); /* :158-159 */
}

static void *his_create(obs_data_t *settings, obs_source_t *source)
{
	struct his_source *src = bzalloc(sizeof(struct his_source));
	cm_create(&src->cm, settings, source);
	cm_request(&src->cm, his_surface_cb, src);
	pthread_mutex_init(&src->tex_mutex, NULL);
	his_update(src, settings);
	return src;
}

static void his_destroy(void *data)
{
	struct his_source *src = data;
	cm_destroy(&src->cm);
	bfree(src->tex_buf[0]); /* src/histogram.c:97-98 */
	bfree(src->tex_buf[1]);
	bfree(src->pic.rgba);
	pthread_mutex_destroy(&src->tex_mutex);
	bfree(src);
}

/* src/histogram.c:233-302, the settings this path reads */
static obs_properties_t *his_get_properties(void *data)
{
	struct his_source *src = data;
	obs_properties_t *props = obs_properties_create();
	cm_get_properties(&src->cm, props);
	obs_property_t *prop = obs_properties_add_list(props, "display", obs_module_text("Display"), OBS_COMBO_TYPE_LIST,
						       OBS_COMBO_FORMAT_INT);
	obs_property_list_add_int(prop, obs_module_text("Overlay"), 0);
	obs_property_list_add_int(prop, obs_module_text("Stack"), 1);
	obs_property_list_add_int(prop, obs_module_text("Parade"), 2);
	prop = obs_properties_add_list(props, "components", obs_module_text("Components"), OBS_COMBO_TYPE_LIST,
				       OBS_COMBO_FORMAT_INT);
	obs_property_list_add_int(prop, obs_module_text("RGB"), COMP_RGB);
	obs_property_list_add_int(prop, obs_module_text("Luma"), COMP_Y);
	obs_property_list_add_int(prop, obs_module_text("Chroma"), COMP_UV);
	obs_property_list_add_int(prop, obs_module_text("YUV"), COMP_YUV);
	obs_properties_add_int(props, "level_height", obs_module_text("Height"), 50, 2048, 1);
	obs_properties_add_bool(props, "logscale", obs_module_text("Log scale"));
	prop = obs_properties_add_list(props, "level_mode", obs_module_text("Level"), OBS_COMBO_TYPE_LIST, OBS_COMBO_FORMAT_INT);
	obs_property_list_add_int(prop, obs_module_text("Auto"), 0);
	obs_property_list_add_int(prop, obs_module_text("Pixels"), 1);
	obs_property_list_add_int(prop, obs_module_text("Ratio"), 2);
	obs_properties_add_int(props, "level_fixed_value", obs_module_text("Pixels"), 1, 1 << 30, 1);
	obs_properties_add_int(props, "level_ratio_value", obs_module_text("Ratio"), 1, 1000, 1);
	return props;
}

static void his_get_defaults(obs_data_t *settings)
{
	cm_get_defaults(settings);
	obs_data_set_default_int(settings, "components", COMP_RGB); /* src/histogram.c:167 */
	obs_data_set_default_bool(settings, "logscale", false);
	obs_data_set_default_int(settings, "level_mode", 0);
	obs_data_set_default_int(settings, "display", 0);
	obs_data_set_default_int(settings, "level_height", 200); /* src/histogram.c:168 */
	obs_data_set_default_int(settings, "graticule_vertical_lines", 5); /* :169 */
	obs_data_set_default_int(settings, "level_fixed_value", 1000);     /* :170 */
	obs_data_set_default_double(settings, "level_ratio_value", 10.0);  /* :171: a percentage */
}

static uint32_t his_get_width(void *data)
{
	struct his_source *src = data; /* src/histogram.c:304-312 */
	uint32_t w, h;
	scope_display_size(HI_SIZE, (uint32_t)src->level_height, src->components, src->display, &w, &h);
	return w;
}

static uint32_t his_get_height(void *data)
{
	struct his_source *src = data; /* src/histogram.c:314-322 */
	uint32_t w, h;
	scope_display_size(HI_SIZE, (uint32_t)src->level_height, src->components, src->display, &w, &h);
	return h;
}

/* src/histogram.c:522-578 minus graticule */
static void his_render_locked(struct his_source *src)
{
	const int r = src->w_tex_buf ^ 1;
	if (!src->tex_buf[r])
		return;
	uint32_t w, h;
	scope_display_size(HI_SIZE, (uint32_t)src->level_height, src->components, src->display, &w, &h);
	uint8_t *rgba = picture_reserve(&src->pic, w, h);
	if (!rgba || !w || !h)
		return;
	PROFILE_START(PROF_HIS_RENDER);
	PROFILE_START(PROF_DRAW);
	const int st = scope_display_histogram(cm_gpu_render(), (const float *)src->tex_buf[r], src->hi_max[r], src->components,
					       src->display, (uint32_t)src->level_height, 0, rgba);
	PROFILE_END(PROF_DRAW);
	log_failure("scope_display_histogram", st);
	if (st == SCOPE_OK) { /* src/histogram.c:570-575 */
		uint32_t nc = 0;
		for (uint32_t c = src->components & 0x77u; c; c >>= 1)
			nc += c & 1u;
		struct scope_vertex lines[2048];
		const uint32_t max = sizeof lines / sizeof lines[0];
		uint32_t n = scope_graticule_histogram(src->graticule_vertical_lines, src->graticule_horizontal_step, src->logscale,
						       src->level_fixed_value, src->level_ratio_value, src->display, nc,
						       (uint32_t)src->level_height, lines, max);
		if (n > max)
			n = max;
		if (n)
			log_failure("scope_display_graticule",
				    scope_display_graticule(cm_gpu_render(), rgba, w, h, NULL, 0, (const struct scope_graticule_vertex *)lines, n,
							    0x80FFBF00 /* amber, :495 */, 0));
	}
	PROFILE_END(PROF_HIS_RENDER);
}

static void his_render(void *data, gs_effect_t *effect)
{
	UNUSED_PARAMETER(effect);
	struct his_source *src = data;
	pthread_mutex_lock(&src->tex_mutex); /* the table being drawn is not the worker's until this section ends (scope_sources.h) */
	his_render_locked(src);
	pthread_mutex_unlock(&src->tex_mutex);
}

void his_surface_cb(void *data, struct cm_surface_data *surface_data)
{
	struct his_source *src = data;

	if ((src->components & COMP_RGB) && !surface_data->rgb_data)
		return;
	if ((src->components & COMP_YUV) && !surface_data->yuv_data)
		return;
	if (!surface_data->width)
		return;

	if (!src->tex_buf[src->w_tex_buf])
		src->tex_buf[src->w_tex_buf] = bzalloc(sizeof(float) * HI_SIZE * 4);

	/* was: his_draw_histogram(src, tex_buf, hi_max, surface_data), in the same profiler zone */
	PROFILE_START(PROF_DRAW_HISTOGRAM);
	const int st = scope_histogram(cm_gpu(), as_scope(surface_data), src->components, src->logscale,
				       src->level_fixed_value, src->level_ratio_value, 0, src->tex_buf[src->w_tex_buf],
				       src->hi_max[src->w_tex_buf]);
	PROFILE_END(PROF_DRAW_HISTOGRAM);
	if (st != SCOPE_OK) {
		log_failure("scope_histogram", st);
		return;
	}
	TEX_FLIP(src);
}

const struct obs_source_info colormonitor_histogram = {
	.id = "histogram_source",
	.type = OBS_SOURCE_TYPE_INPUT,
	.output_flags = OBS_SOURCE_VIDEO | OBS_SOURCE_CUSTOM_DRAW,
	.get_name = his_get_name,
	.create = his_create,
	.destroy = his_destroy,
	.update = his_update,
	.get_defaults = his_get_defaults,
	.get_properties = his_get_properties,
	.get_width = his_get_width,
	.get_height = his_get_height,
	.enum_active_sources = cm_enum_sources,
	.video_render = his_render,
	.video_tick = cm_tick,
};

/* -- ROI: one capture shared by several scopes ----------------------------------------------------------- */

static const char *roi_get_name(void *unused)
{
	UNUSED_PARAMETER(unused);
	return "ROI";
}

static void roi_update(void *data, obs_data_t *settings)
{
	struct roi_source *src = data;
	cm_update(&src->cm, settings);
	src->n_interleave = (int)obs_data_get_int(settings, "interleave"); /* src/roi.c:108 */
	/* the rectangle arrives through mouse interaction in the plugin (src/roi.c:343-521); a host sets it here */
	src->cm.x0 = (int)obs_data_get_int(settings, "roi_x0");
	src->cm.y0 = (int)obs_data_get_int(settings, "roi_y0");
	src->cm.x1 = (int)obs_data_get_int(settings, "roi_x1");
	src->cm.y1 = (int)obs_data_get_int(settings, "roi_y1");
}

static void roi_get_defaults(obs_data_t *settings)
{
	cm_get_defaults(settings);
	obs_data_set_default_int(settings, "interleave", 1); /* src/roi.c:99 */
	obs_data_set_default_int(settings, "roi_x0", -1);    /* no rectangle: the whole capture */
	obs_data_set_default_int(settings, "roi_y0", -1);
	obs_data_set_default_int(settings, "roi_x1", -1);
	obs_data_set_default_int(settings, "roi_y1", -1);
}

/* The same for a host that holds the full-size RGB target: scale, crop and K0 run on the GPU. */
bool roi_target_render_full(struct roi_source *src, const uint8_t *target_rgb, uint32_t linesize, uint32_t target_width,
			    uint32_t target_height)
{
	src->interleave_rendered = true;
	if (src->i_interleave != 0 && src->n_interleave > 0)
		return true; /* this frame is skipped */

	cm_render_target(&src->cm, target_rgb, linesize, target_width, target_height);

	if (src->n_interleave <= 0)
		return true;
	return false;
}

bool roi_target_render(struct roi_source *src, const uint8_t *video_data, uint32_t linesize, uint32_t width,
		       uint32_t height)
{
	src->interleave_rendered = true;
	if (src->i_interleave != 0 && src->n_interleave > 0)
		return true; /* this frame is skipped */

	cm_stage_surface(&src->cm, video_data, linesize, width, height);

	if (src->n_interleave <= 0)
		return true;
	return false;
}

static void *roi_create(obs_data_t *settings, obs_source_t *source)
{
	struct roi_source *src = bzalloc(sizeof(struct roi_source));
	src->cm.flags = CM_FLAG_ROI | CM_FLAG_RAW_TEXTURE; /* src/roi.c:35 */
	cm_create(&src->cm, settings, source);
	pthread_mutex_init(&src->sources_mutex, NULL);
	cm_request(&src->cm, roi_surface_cb, src);
	roi_update(src, settings);
	return src;
}

static void roi_destroy(void *data)
{
	struct roi_source *src = data;
	cm_destroy(&src->cm);
	pthread_mutex_destroy(&src->sources_mutex);
	bfree(src);
}

void roi_register_source(struct roi_source *src, struct cm_source *cm)
{
	pthread_mutex_lock(&src->sources_mutex);
	if (src->n_sources < ROI_MAX_SOURCES)
		src->sources[src->n_sources++] = cm;
	pthread_mutex_unlock(&src->sources_mutex);
	cm->roi = src;
}

void roi_unregister_source(struct roi_source *src, struct cm_source *cm)
{
	pthread_mutex_lock(&src->sources_mutex);
	for (size_t i = 0; i < src->n_sources; i++) {
		if (src->sources[i] == cm) {
			memmove(&src->sources[i], &src->sources[i + 1], (src->n_sources - i - 1) * sizeof(src->sources[0]));
			src->n_sources--;
			break;
		}
	}
	pthread_mutex_unlock(&src->sources_mutex);
	cm->roi = NULL;
}

void roi_update_flags(struct roi_source *src)
{
	/* src/roi.c:534-540: the ROI captures whatever planes its scopes need */
	uint32_t flags = CM_FLAG_ROI | CM_FLAG_RAW_TEXTURE;
	pthread_mutex_lock(&src->sources_mutex);
	for (size_t i = 0; i < src->n_sources; i++)
		flags |= src->sources[i]->flags & (CM_FLAG_CONVERT_RGB | CM_FLAG_CONVERT_YUV);
	pthread_mutex_unlock(&src->sources_mutex);
	src->cm.flags = flags;
}

void roi_surface_cb(void *data, struct cm_surface_data *surface_data)
{
	struct roi_source *src = data;

	pthread_mutex_lock(&src->sources_mutex);
	for (size_t i = 0; i < src->n_sources; i++) {
		struct cm_source *cm = src->sources[i];
		if (cm->callback)
			cm->callback(cm->callback_data, surface_data);
	}
	pthread_mutex_unlock(&src->sources_mutex);
}

static void roi_tick(void *data, float unused)
{
	struct roi_source *src = data;

	/* src/roi.c:523-532 */
	if (src->interleave_rendered && src->i_interleave++ >= src->n_interleave)
		src->i_interleave = 0;
	src->interleave_rendered = false;

	if (src->i_interleave == 0 || src->n_interleave <= 0)
		cm_tick(&src->cm, unused);

	roi_update_flags(src);
}

/* src/roi.c:79-96 */
static obs_properties_t *roi_get_properties(void *data)
{
	struct roi_source *src = data;
	obs_properties_t *props = obs_properties_create();
	cm_get_properties(&src->cm, props);
	obs_properties_add_int(props, "interleave", obs_module_text("Interleave"), 0, 1, 1);
	return props;
}

static uint32_t roi_get_size(void *data)
{
	UNUSED_PARAMETER(data);
	return 0;
}

const struct obs_source_info colormonitor_roi = {
	.id = "colormonitor_roi",
	.type = OBS_SOURCE_TYPE_INPUT,
	.output_flags = OBS_SOURCE_VIDEO | OBS_SOURCE_CUSTOM_DRAW | OBS_SOURCE_INTERACTION,
	.get_name = roi_get_name,
	.create = roi_create,
	.destroy = roi_destroy,
	.update = roi_update,
	.get_defaults = roi_get_defaults,
	.get_properties = roi_get_properties,
	.get_width = roi_get_size,
	.get_height = roi_get_size,
	.enum_active_sources = cm_enum_sources,
	.video_render = scope_render_nothing,
	.video_tick = roi_tick,
};

/* -- module entry (src/plugin-main.c:58-108, the sources on this path) ------------------------------------ */

bool obs_module_load(void)
{
	if (cm_gpu_open(0) != SCOPE_OK)
		return false; /* no GPU: the module refuses to load rather than fall back to a CPU loop */
	obs_register_source(&colormonitor_vectorscope_v1); /* src/plugin-main.c:82-85: the obsolete version stays loadable */
	obs_register_source(&colormonitor_vectorscope_v2);
	obs_register_source(&colormonitor_waveform);
	obs_register_source(&colormonitor_histogram);
	obs_register_source(&colormonitor_roi);
	blog(LOG_INFO, "[obs-color-monitor] scope accumulation on HIP device 0 (abi %d)", scope_abi_version());
	return true;
}

void obs_module_unload(void)
{
	cm_gpu_close();
}
