/* miniweed.c -- a small weed HOST for the bench / C tests of this repo (no part of the product, no part of the oracle).
 *
 * The layer seam and the plugin seam only ever see a weed host through the function pointers of the weed ABI (libweed/weed.h:224-237, the bootstrap of
 * libweed/weed-effects.h:170-186).  LiVES hands them libweed's; tools/seam_host.c, which drives the seams from C threads the way src/nodemodel.c drives them,
 * needs a host of its own that can travel to the GPU box -- this file: plants as arrays of typed leaves, the six accessors, weed_bootstrap.  It implements what
 * this repo's two libraries call (leaf get / set / num_elements / delete, plant_new, malloc / free at bootstrap), not the whole of libweed (no leaf flags, no
 * locking: a plant is used by one thread at a time here).  Written from the ABI in include/lives_gpu_weed_abi.h. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../include/lives_gpu_weed_abi.h"
#include "miniweed.h"

typedef struct {
  char key[40];
  weed_seed_t seed;
  weed_size_t n, cap;
  union { int32_t *i; double *d; int64_t *l; void **p; char **s; void *raw; } v;
} mw_leaf;
struct _weed_leaf {            /* weed_plant_t */
  int nleaves, cap;
  mw_leaf *leaves;
};

static size_t esize(weed_seed_t seed) {
  switch (seed) {
  case WEED_SEED_INT: case WEED_SEED_BOOLEAN: return 4;
  case WEED_SEED_DOUBLE: case WEED_SEED_INT64: return 8;
  default: return sizeof(void *);       /* strings (owned copies), funcptr, voidptr, plantptr */
  }
}
static mw_leaf *find(weed_plant_t *p, const char *key) {
  if (!p || !key) return NULL;
  for (int i = 0; i < p->nleaves; i++) if (!strcmp(p->leaves[i].key, key)) return &p->leaves[i];
  return NULL;
}
static void drop_values(mw_leaf *l) {
  if (l->seed == WEED_SEED_STRING) for (weed_size_t i = 0; i < l->n; i++) free(l->v.s[i]);
  l->n = 0;
}

weed_error_t mw_leaf_set(weed_plant_t *p, const char *key, weed_seed_t seed, weed_size_t n, weed_voidptr_t values) {
  if (!p || !key || strlen(key) >= sizeof(((mw_leaf *)0)->key)) return WEED_ERROR_NOSUCH_LEAF;
  mw_leaf *l = find(p, key);
  if (l && l->n && l->seed != seed) return WEED_ERROR_WRONG_SEED_TYPE;
  if (!l) {
    if (p->nleaves == p->cap) {
      const int nc = p->cap ? p->cap * 2 : 16;
      mw_leaf *nl = (mw_leaf *)realloc(p->leaves, (size_t)nc * sizeof(mw_leaf));
      if (!nl) return WEED_ERROR_MEMORY_ALLOCATION;
      p->leaves = nl; p->cap = nc;
    }
    l = &p->leaves[p->nleaves++];
    memset(l, 0, sizeof *l);
    strcpy(l->key, key);
  }
  drop_values(l);
  l->seed = seed;
  if (n > l->cap) {
    void *nv = realloc(l->v.raw, (size_t)n * esize(seed));
    if (!nv) return WEED_ERROR_MEMORY_ALLOCATION;
    l->v.raw = nv; l->cap = n;
  }
  if (seed == WEED_SEED_STRING) {
    for (weed_size_t i = 0; i < n; i++) { const char *s = ((char **)values)[i]; l->v.s[i] = strdup(s ? s : ""); }
  } else if (n) memcpy(l->v.raw, values, (size_t)n * esize(seed));
  l->n = n;
  return WEED_SUCCESS;
}
weed_error_t mw_leaf_get(weed_plant_t *p, const char *key, weed_size_t idx, weed_voidptr_t value) {
  mw_leaf *l = find(p, key);
  if (!l) return WEED_ERROR_NOSUCH_LEAF;
  if (idx >= l->n) return WEED_ERROR_NOSUCH_ELEMENT;
  if (!value) return WEED_SUCCESS;
  if (l->seed == WEED_SEED_STRING) {               /* libweed copies the characters into the caller's buffer (weed.c, _weed_leaf_get) */
    strcpy(*(char **)value, l->v.s[idx]);
    return WEED_SUCCESS;
  }
  memcpy(value, (char *)l->v.raw + (size_t)idx * esize(l->seed), esize(l->seed));
  return WEED_SUCCESS;
}
weed_size_t mw_leaf_num_elements(weed_plant_t *p, const char *key) {
  mw_leaf *l = find(p, key);
  return l ? l->n : 0;
}
weed_error_t mw_leaf_delete(weed_plant_t *p, const char *key) {
  mw_leaf *l = find(p, key);
  if (!l) return WEED_ERROR_NOSUCH_LEAF;
  drop_values(l);
  free(l->v.raw);
  *l = p->leaves[--p->nleaves];
  return WEED_SUCCESS;
}
weed_plant_t *mw_plant_new(int32_t type) {
  weed_plant_t *p = (weed_plant_t *)calloc(1, sizeof *p);
  if (p) mw_leaf_set(p, WEED_LEAF_TYPE, WEED_SEED_INT, 1, &type);
  return p;
}
void mw_plant_free(weed_plant_t *p) {
  if (!p) return;
  for (int i = 0; i < p->nleaves; i++) { drop_values(&p->leaves[i]); free(p->leaves[i].v.raw); }
  free(p->leaves);
  free(p);
}
const char *mw_string(weed_plant_t *p, const char *key) {        /* host-side shortcut: the stored characters themselves */
  mw_leaf *l = find(p, key);
  return (l && l->seed == WEED_SEED_STRING && l->n) ? l->v.s[0] : NULL;
}

/* ---- bootstrap (libweed/weed-effects.h:170-186; what libweed/weed-host-utils.c's weed_bootstrap answers): a HOST_INFO plant whose leaves hold the host's
   functions; the default getter reads element 0 of a leaf */
static weed_error_t mw_default_get(weed_plant_t *p, const char *key, void *value) { return mw_leaf_get(p, key, 0, value); }
weed_plant_t *mw_bootstrap(weed_default_getter_f *getter, int32_t plugin_weed_min, int32_t plugin_weed_max, int32_t plugin_filter_min, int32_t plugin_filter_max) {
  (void)plugin_weed_min; (void)plugin_filter_min;
  weed_plant_t *hi = mw_plant_new(WEED_PLANT_HOST_INFO);
  if (!hi || !getter) return NULL;
  *getter = mw_default_get;
  weed_funcptr_t f;
  int32_t v;
#define FN(leaf, fn) do { f = (weed_funcptr_t)(fn); mw_leaf_set(hi, leaf, WEED_SEED_FUNCPTR, 1, &f); } while (0)
  FN(WEED_LEAF_GET_FUNC, mw_leaf_get); FN(WEED_LEAF_SET_FUNC, mw_leaf_set); FN(WEED_LEAF_DELETE_FUNC, mw_leaf_delete); FN(WEED_PLANT_NEW_FUNC, mw_plant_new);
  FN(WEED_PLANT_FREE_FUNC, mw_plant_free); FN(WEED_LEAF_NUM_ELEMENTS_FUNC, mw_leaf_num_elements); FN(WEED_LEAF_MALLOC_FUNC, malloc); FN(WEED_LEAF_FREE_FUNC, free);
#undef FN
  v = plugin_weed_max; mw_leaf_set(hi, WEED_LEAF_WEED_API_VERSION, WEED_SEED_INT, 1, &v);
  v = plugin_filter_max; mw_leaf_set(hi, WEED_LEAF_FILTER_API_VERSION, WEED_SEED_INT, 1, &v);
  return hi;
}
