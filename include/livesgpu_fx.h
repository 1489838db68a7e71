/* livesgpu_fx.h -- what livesgpu_fx.so (the weed plugin of this repo) exports.
 *
 * weed_setup() is the weed plugin ABI's one entry point (libweed/weed-plugin.h; a host finds the filter classes, their
 * init / process / deinit functions and everything else through the plant it returns): 33 filter classes under the
 * reference plugins' names (lives-plugins/weed-plugins/{simple_blend,multi_blends,mirrors,edge,softlight,blurzoom,
 * multi_transitions,slide_over,layout_blends,deinterlace,RGBdelay}.c, the generated script effects, gdk/compositor.c).
 *
 * livesgpu_fx_process_batch() is an extension of THIS plugin (a host looks it up with dlsym and keeps its per-instance
 * loop when it is not there): the n instances of ONE filter class that a plan step applies to n tracks
 * (src/effects-weed.c:1850-2425 calls weed_apply_instance once per track) in ONE launch, when the class has a batch
 * kernel -- the transitions of multi_transitions.c, the blends of simple_blend.c / multi_blends.c, softlight -- and the
 * channels share palette, size and rowstrides; anything else (another class, mixed classes, ARGB32 frames, sliced
 * channels, more than 16 instances, an instance that reads or writes the plane another one writes) runs process_func
 * instance by instance behind the same call, so the result never depends on which way was taken.  Channels whose
 * pixel_data is the plane of a pinned layer (lives_gpu_layer_pin, lives_gpu_layer.h) are used where they live in HBM.
 * INTEGRATION.md shows the host side.
 */
#ifndef LIVESGPU_FX_H
#define LIVESGPU_FX_H
#include "lives_gpu_weed_abi.h"
#ifdef __cplusplus
extern "C" {
#endif
#ifndef __WEED_PLUGIN_H__     /* the real weed-plugin.h declares it */
weed_plant_t *weed_setup(weed_bootstrap_f weed_boot);
#endif
weed_error_t livesgpu_fx_process_batch(weed_plant_t **instances, int n, weed_timecode_t tc);
#ifdef __cplusplus
}
#endif
#endif
