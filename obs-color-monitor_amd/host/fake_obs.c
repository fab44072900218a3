/*
 * fake_obs.c -- stand-in for the libobs calls obs_shim.h declares, plus a tiny "OBS" that the tests
 * and the stand-alone host drive: a registry of obs_source_info tables and a settings object.
 * Stand-alone host only; inside the real plugin libobs provides all of this.
 */
#include "obs_shim.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MAX_KEYS 32
struct obs_data {
	char key[MAX_KEYS][32];
	long long val[MAX_KEYS];
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
	d->user[d->n] = false;
	return d->n++;
}

long long obs_data_get_int(obs_data_t *d, const char *name)
{
	const int i = d ? find_key(d, name, false) : -1;
	return i < 0 ? 0 : d->val[i];
}

bool obs_data_get_bool(obs_data_t *d, const char *name)
{
	return obs_data_get_int(d, name) != 0;
}

void obs_data_set_default_int(obs_data_t *d, const char *name, long long val)
{
	const int i = find_key(d, name, true);
	if (i >= 0 && !d->user[i])
		d->val[i] = val;
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
		d->user[i] = true;
	}
}

const struct obs_source_info *fake_obs_find_source(const char *id)
{
	for (int i = 0; i < g_n_infos; i++)
		if (!strcmp(g_infos[i]->id, id))
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
