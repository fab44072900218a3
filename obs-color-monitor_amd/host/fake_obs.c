/*
 * fake_obs.c -- stand-in for the libobs calls obs_shim.h declares, plus a tiny "OBS" that the tests
 * and the stand-alone host drive: a registry of obs_source_info tables and a settings object.
 * Stand-alone host only; inside the real plugin libobs provides all of this.
 */
#include "obs_shim.h"

#include <pthread.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MAX_KEYS 32
struct obs_data {
	char key[MAX_KEYS][32];
	long long val[MAX_KEYS];
	double dval[MAX_KEYS]; /* what obs_data_get_double returns: libobs keeps numbers as int or double and converts on read */
	bool user[MAX_KEYS];
	int n;
};

static const struct obs_source_info *g_infos[16];
static int g_n_infos;
static int g_log_level = LOG_INFO;

void blog(int level, const char *format, ...)
{
	if (level > g_log_level)
		return;
	va_list ap;
	va_start(ap, format);
	vfprintf(stderr, format, ap);
	fputc('\n', stderr);
	va_end(ap);
}

void *bzalloc(size_t size)
{
	return calloc(1, size ? size : 1);
}

void bfree(void *ptr)
{
	free(ptr);
}

static int find_key(obs_data_t *d, const char *name, bool add)
{
	for (int i = 0; i < d->n; i++)
		if (!strcmp(d->key[i], name))
			return i;
	if (!add || d->n == MAX_KEYS)
		return -1;
	snprintf(d->key[d->n], sizeof d->key[0], "%s", name);
	d->val[d->n] = 0;
	d->dval[d->n] = 0.0;
	d->user[d->n] = false;
	return d->n++;
}

long long obs_data_get_int(obs_data_t *d, const char *name)
{
	const int i = d ? find_key(d, name, false) : -1;
	return i < 0 ? 0 : d->val[i];
}

double obs_data_get_double(obs_data_t *d, const char *name)
{
	const int i = d ? find_key(d, name, false) : -1;
	return i < 0 ? 0.0 : d->dval[i];
}

bool obs_data_get_bool(obs_data_t *d, const char *name)
{
	return obs_data_get_int(d, name) != 0;
}

void obs_data_set_default_int(obs_data_t *d, const char *name, long long val)
{
	const int i = find_key(d, name, true);
	if (i >= 0 && !d->user[i]) {
		d->val[i] = val;
		d->dval[i] = (double)val;
	}
}

void obs_data_set_default_double(obs_data_t *d, const char *name, double val)
{
	const int i = find_key(d, name, true);
	if (i >= 0 && !d->user[i]) {
		d->val[i] = (long long)val;
		d->dval[i] = val;
	}
}

void obs_data_set_default_bool(obs_data_t *d, const char *name, bool val)
{
	obs_data_set_default_int(d, name, val);
}

void obs_register_source(const struct obs_source_info *info)
{
	if (g_n_infos < 16)
		g_infos[g_n_infos++] = info;
}

/* ---- what the driving side (tests, host_demo) uses ---------------------------------------------- */

obs_data_t *fake_obs_data_create(void)
{
	return calloc(1, sizeof(struct obs_data));
}

void fake_obs_data_release(obs_data_t *d)
{
	free(d);
}

void fake_obs_data_set_int(obs_data_t *d, const char *name, long long val)
{
	const int i = find_key(d, name, true);
	if (i >= 0) {
		d->val[i] = val;
		d->dval[i] = (double)val;
		d->user[i] = true;
	}
}

void fake_obs_data_set_double(obs_data_t *d, const char *name, double val)
{
	const int i = find_key(d, name, true);
	if (i >= 0) {
		d->val[i] = (long long)val;
		d->dval[i] = val;
		d->user[i] = true;
	}
}

/* the latest version registered under an id, like obs_get_latest_input_type_id (src/plugin-main.c:50) */
const struct obs_source_info *fake_obs_find_source(const char *id)
{
	const struct obs_source_info *best = NULL;
	for (int i = 0; i < g_n_infos; i++)
		if (!strcmp(g_infos[i]->id, id) && (!best || g_infos[i]->version > best->version))
			best = g_infos[i];
	return best;
}

const struct obs_source_info *fake_obs_find_source_version(const char *id, uint32_t version)
{
	for (int i = 0; i < g_n_infos; i++)
		if (!strcmp(g_infos[i]->id, id) && g_infos[i]->version == version)
			return g_infos[i];
	return NULL;
}

int fake_obs_source_count(void)
{
	return g_n_infos;
}

void fake_obs_reset(void)
{
	g_n_infos = 0;
}

/* ---- properties: just enough of a list for a host (or a test) to enumerate what a source exposes --------------- */

#define MAX_PROPS 24
struct obs_property {
	char name[32];
	int kind; /* 0 int, 1 bool, 2 list */
	int min, max, step;
	long long item[12];
	int n_items;
};
struct obs_properties {
	struct obs_property p[MAX_PROPS];
	int n;
};

obs_properties_t *obs_properties_create(void)
{
	return calloc(1, sizeof(struct obs_properties));
}

void obs_properties_destroy(obs_properties_t *props)
{
	free(props);
}

static obs_property_t *add_prop(obs_properties_t *props, const char *name, int kind)
{
	if (!props || props->n == MAX_PROPS)
		return NULL;
	obs_property_t *p = &props->p[props->n++];
	snprintf(p->name, sizeof p->name, "%s", name);
	p->kind = kind;
	return p;
}

obs_property_t *obs_properties_add_int(obs_properties_t *props, const char *name, const char *description, int min, int max,
				       int step)
{
	(void)description;
	obs_property_t *p = add_prop(props, name, 0);
	if (p) {
		p->min = min;
		p->max = max;
		p->step = step;
	}
	return p;
}

obs_property_t *obs_properties_add_bool(obs_properties_t *props, const char *name, const char *description)
{
	(void)description;
	return add_prop(props, name, 1);
}

obs_property_t *obs_properties_add_list(obs_properties_t *props, const char *name, const char *description,
					enum obs_combo_type type, enum obs_combo_format format)
{
	(void)description;
	(void)type;
	(void)format;
	return add_prop(props, name, 2);
}

size_t obs_property_list_add_int(obs_property_t *p, const char *name, long long val)
{
	(void)name;
	if (!p || p->n_items == 12)
		return 0;
	p->item[p->n_items] = val;
	return (size_t)p->n_items++;
}

const char *obs_module_text(const char *lookup)
{
	return lookup; /* no locale files here */
}

int fake_obs_properties_count(const obs_properties_t *props)
{
	return props ? props->n : 0;
}

/* kind: 0 int, 1 bool, 2 list; for a list *min / *max are its smallest / largest item */
const char *fake_obs_property(const obs_properties_t *props, int i, int *kind, int *min, int *max, int *n_items)
{
	if (!props || i < 0 || i >= props->n)
		return NULL;
	const struct obs_property *p = &props->p[i];
	*kind = p->kind;
	*min = p->min;
	*max = p->max;
	*n_items = p->n_items;
	if (p->kind == 2 && p->n_items) {
		*min = *max = (int)p->item[0];
		for (int k = 1; k < p->n_items; k++) {
			*min = p->item[k] < *min ? (int)p->item[k] : *min;
			*max = p->item[k] > *max ? (int)p->item[k] : *max;
		}
	}
	return p->name;
}

/* ---- profiler: counts the zones so that a test can see them open and close in pairs ------------------------------ */

#define MAX_ZONES 24
static struct {
	const char *name;
	long starts, ends;
} g_zones[MAX_ZONES];
static int g_n_zones;
static pthread_mutex_t g_zone_mutex = PTHREAD_MUTEX_INITIALIZER;

static void zone_count(const char *name, int end)
{
	pthread_mutex_lock(&g_zone_mutex);
	int i;
	for (i = 0; i < g_n_zones; i++)
		if (!strcmp(g_zones[i].name, name))
			break;
	if (i == g_n_zones && g_n_zones < MAX_ZONES) {
		g_zones[g_n_zones].name = name;
		g_zones[g_n_zones].starts = g_zones[g_n_zones].ends = 0;
		g_n_zones++;
	}
	if (i < g_n_zones) {
		if (end)
			g_zones[i].ends++;
		else
			g_zones[i].starts++;
	}
	pthread_mutex_unlock(&g_zone_mutex);
}

void profile_start(const char *name)
{
	zone_count(name, 0);
}

void profile_end(const char *name)
{
	zone_count(name, 1);
}

/* starts of zone `name` (-1: never seen); *balanced = every start has had its end */
long fake_obs_profile_zone(const char *name, int *balanced)
{
	long n = -1;
	pthread_mutex_lock(&g_zone_mutex);
	for (int i = 0; i < g_n_zones; i++)
		if (!strcmp(g_zones[i].name, name)) {
			n = g_zones[i].starts;
			if (balanced)
				*balanced = g_zones[i].starts == g_zones[i].ends;
		}
	pthread_mutex_unlock(&g_zone_mutex);
	return n;
}
