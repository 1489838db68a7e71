/* miniweed.h -- the small weed host of tools/miniweed.c (bench / C-test infrastructure) */
#ifndef MINIWEED_H
#define MINIWEED_H
#include "../include/lives_gpu_weed_abi.h"
weed_error_t mw_leaf_set(weed_plant_t *p, const char *key, weed_seed_t seed, weed_size_t n, weed_voidptr_t values);
weed_error_t mw_leaf_get(weed_plant_t *p, const char *key, weed_size_t idx, weed_voidptr_t value);
weed_size_t mw_leaf_num_elements(weed_plant_t *p, const char *key);
weed_error_t mw_leaf_delete(weed_plant_t *p, const char *key);
weed_plant_t *mw_plant_new(int32_t type);
void mw_plant_free(weed_plant_t *p);
const char *mw_string(weed_plant_t *p, const char *key);
weed_plant_t *mw_bootstrap(weed_default_getter_f *getter, int32_t plugin_weed_min, int32_t plugin_weed_max, int32_t plugin_filter_min, int32_t plugin_filter_max);
#endif
