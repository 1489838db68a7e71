/* lives_gpu_weed_abi.h -- the slice of the Weed ABI that liblivesgpu.so talks.
 *
 * A LiVES build includes the real <weed/weed.h>, <weed/weed-palettes.h>, <weed/weed-effects.h>; this
 * header exists so that the library builds without the LiVES tree.  It only restates numeric ids,
 * leaf-name strings and function-pointer signatures of the public ABI (libweed/weed.h:150-262,:373-465;
 * libweed/weed-palettes.h:43-185; libweed/weed-effects.h:44-400).  tests/test_host_cpu.py
 * (test_weed_abi_constants_match_reference_headers) checks every value below against the reference headers when /root/reference is present.
 *
 * If the real weed headers were included first, nothing here is redefined.
 */
#ifndef LIVES_GPU_WEED_ABI_H
#define LIVES_GPU_WEED_ABI_H
#include <stdint.h>
#include <stddef.h>

#ifndef WEED_PALETTE_END   /* ---- palettes: weed-palettes.h:43-102 ---- */
#define WEED_PALETTE_NONE 0
#define WEED_PALETTE_ANY -1          /* libweed/weed-palettes.h:41 */
#define WEED_PALETTE_END 0
#define WEED_PALETTE_RGB24 1
#define WEED_PALETTE_BGR24 2
#define WEED_PALETTE_RGBA32 3
#define WEED_PALETTE_BGRA32 4
#define WEED_PALETTE_ARGB32 5
#define WEED_PALETTE_RGBFLOAT 64
#define WEED_PALETTE_RGBAFLOAT 65
#define WEED_PALETTE_YUV420P 512
#define WEED_PALETTE_YVU420P 513
#define WEED_PALETTE_YUV422P 522
#define WEED_PALETTE_YUV444P 544
#define WEED_PALETTE_YUVA4444P 545
#define WEED_PALETTE_UYVY 564
#define WEED_PALETTE_YUYV 565
#define WEED_PALETTE_YUV888 588
#define WEED_PALETTE_YUVA8888 589
#define WEED_PALETTE_YUV411 595
#define WEED_PALETTE_A8 1024
#define WEED_PALETTE_A1 1025
#define WEED_PALETTE_AFLOAT 1064
/* weed-palettes.h:154-185 */
#define WEED_YUV_SAMPLING_DEFAULT 0
#define WEED_YUV_SAMPLING_JPEG 0
#define WEED_YUV_SAMPLING_MPEG 1
#define WEED_YUV_CLAMPING_CLAMPED 0
#define WEED_YUV_CLAMPING_UNCLAMPED 1
#define WEED_YUV_SUBSPACE_YUV 0
#define WEED_YUV_SUBSPACE_YCBCR 1
#define WEED_YUV_SUBSPACE_BT709 2
#define WEED_GAMMA_UNKNOWN 0
#define WEED_GAMMA_LINEAR (-1)
#define WEED_GAMMA_SRGB 1
#define WEED_GAMMA_BT709 2
#endif

/* LiVES-side additions (src/colourspace.h:25-29, src/widget-helper-gtk.h:1136-1138, src/main.h) */
#define LIVES_GAMMA_MONITOR 1024
#define LIVES_GAMMA_FILE 1025
#define LIVES_GAMMA_VARIANT 2048
#define LIVES_LAYER_ALPHA_PREMULT 1
#define LIVES_INTERP_FAST 0      /* GDK_INTERP_NEAREST */
#define LIVES_INTERP_NORMAL 2    /* GDK_INTERP_BILINEAR */
#define LIVES_INTERP_BEST 3      /* GDK_INTERP_HYPER */
#define LIVES_DIRECTION_REVERSE (-1)
#define LIVES_DIRECTION_FORWARD 1

#ifndef WEED_SEED_INT      /* ---- core: weed.h:373-465 ---- */
typedef struct _weed_leaf weed_leaf_t;
typedef weed_leaf_t weed_plant_t;
typedef int32_t weed_error_t;
typedef uint32_t weed_size_t;
typedef uint32_t weed_seed_t;
typedef uint32_t weed_flags_t;
typedef void *weed_voidptr_t;
typedef void (*weed_funcptr_t)(void);
typedef int64_t weed_timecode_t;

#define WEED_TRUE 1
#define WEED_FALSE 0
#define WEED_SUCCESS 0
#define WEED_ERROR_MEMORY_ALLOCATION 1
#define WEED_ERROR_NOSUCH_LEAF 2
#define WEED_ERROR_NOSUCH_ELEMENT 3
#define WEED_ERROR_WRONG_SEED_TYPE 4
#define WEED_ERROR_IMMUTABLE 5
#define WEED_SEED_INT 1
#define WEED_SEED_DOUBLE 2
#define WEED_SEED_BOOLEAN 3
#define WEED_SEED_STRING 4
#define WEED_SEED_INT64 5
#define WEED_SEED_FUNCPTR 64
#define WEED_SEED_VOIDPTR 65
#define WEED_SEED_PLANTPTR 66
#define WEED_LEAF_TYPE "type"
#define WEED_LEAF_FLAGS "flags"

/* the six accessors a host hands to a plugin / exports from libweed (weed.h:224-237) */
typedef weed_plant_t *(*weed_plant_new_f)(int32_t plant_type);
typedef weed_error_t (*weed_leaf_set_f)(weed_plant_t *, const char *key, weed_seed_t seed_type,
                                        weed_size_t num_elems, weed_voidptr_t values);
typedef weed_error_t (*weed_leaf_get_f)(weed_plant_t *, const char *key, weed_size_t idx, weed_voidptr_t value);
typedef weed_size_t (*weed_leaf_num_elements_f)(weed_plant_t *, const char *key);
typedef weed_size_t (*weed_leaf_element_size_f)(weed_plant_t *, const char *key, weed_size_t idx);
typedef weed_seed_t (*weed_leaf_seed_type_f)(weed_plant_t *, const char *key);
typedef weed_flags_t (*weed_leaf_get_flags_f)(weed_plant_t *, const char *key);
typedef weed_error_t (*weed_plant_free_f)(weed_plant_t *);
typedef weed_error_t (*weed_leaf_delete_f)(weed_plant_t *, const char *key);
typedef char **(*weed_plant_list_leaves_f)(weed_plant_t *, weed_size_t *nleaves);
typedef void *(*weed_malloc_f)(size_t);
typedef void (*weed_free_f)(void *);
typedef void *(*weed_memset_f)(void *, int, size_t);
typedef void *(*weed_memcpy_f)(void *, const void *, size_t);
typedef void *(*weed_realloc_f)(void *, size_t);
typedef void *(*weed_calloc_f)(size_t, size_t);
typedef void *(*weed_memmove_f)(void *, const void *, size_t);
#endif

#ifndef WEED_PLANT_PLUGIN_INFO   /* ---- effects: weed-effects.h:44-162, :200-400 ---- */
#define WEED_API_VERSION_MIN 200
#define WEED_FILTER_API_VERSION 202
#define WEED_PLANT_PLUGIN_INFO 1
#define WEED_PLANT_FILTER_CLASS 2
#define WEED_PLANT_FILTER_INSTANCE 3
#define WEED_PLANT_CHANNEL_TEMPLATE 4
#define WEED_PLANT_PARAMETER_TEMPLATE 5
#define WEED_PLANT_CHANNEL 6
#define WEED_PLANT_PARAMETER 7
#define WEED_PLANT_GUI 8
#define WEED_PLANT_HOST_INFO 9
#define WEED_PARAM_INTEGER 1
#define WEED_PARAM_FLOAT 2
#define WEED_PARAM_TEXT 3
#define WEED_PARAM_SWITCH 4
#define WEED_PARAM_COLOR 5
#define WEED_COLORSPACE_RGB 1
#define WEED_COLORSPACE_RGBA 2
#define WEED_FILTER_HINT_STATEFUL (1 << 2)
#define WEED_FILTER_PREF_LINEAR_GAMMA (1 << 3)
#define WEED_FILTER_HINT_MAY_THREAD (1 << 6)
#define WEED_CHANNEL_REINIT_ON_SIZE_CHANGE (1 << 0)
#define WEED_PARAMETER_REINIT_ON_VALUE_CHANGE (1 << 0)   /* libweed/weed-effects.h:134 */
#define WEED_CHANNEL_CAN_DO_INPLACE (1 << 4)
#define WEED_ERROR_PLUGIN_INVALID 64
#define WEED_ERROR_FILTER_INVALID 65
#define WEED_ERROR_REINIT_NEEDED 67
#define WEED_ERROR_NOT_READY 68
#define WEED_LEAF_FILTER_CLASS "filter_class"
#define WEED_LEAF_IN_PARAMETERS "in_parameters"
#define WEED_LEAF_IN_CHANNELS "in_channels"
#define WEED_LEAF_OUT_CHANNELS "out_channels"
#define WEED_LEAF_STATE_UPDATED "state_updated"
#define WEED_LEAF_TEMPLATE "template"
#define WEED_LEAF_PIXEL_DATA "pixel_data"
#define WEED_LEAF_CURRENT_PALETTE "current_palette"
#define WEED_LEAF_ROWSTRIDES "rowstrides"
#define WEED_LEAF_OFFSET "offset"
#define WEED_LEAF_WIDTH "width"
#define WEED_LEAF_HEIGHT "height"
#define WEED_LEAF_VALUE "value"
#define WEED_LEAF_GAMMA_TYPE "gamma_type"
#define WEED_LEAF_YUV_CLAMPING "YUV_clamping"
#define WEED_LEAF_RANDOM_SEED "random_seed"           /* libweed/weed-effects.h:319 */
#define WEED_LEAF_GROUP "group"                       /* libweed/weed-effects.h:391 */
/* the compositor class (lives-plugins/weed-plugins/gdk/compositor.c:295-351): libweed/weed-effects.h:113, :135-136, :204, :279, :304, :339, :362, :390 */
#define WEED_FILTER_CHANNEL_SIZES_MAY_VARY (1 << 8)
#define WEED_PARAMETER_VARIABLE_SIZE (1 << 1)
#define WEED_PARAMETER_VALUE_PER_CHANNEL (1 << 2)
#define WEED_LEAF_DESCRIPTION "description"
#define WEED_LEAF_INNER_SIZE "inner_size"
#define WEED_LEAF_LAYOUT_SCHEME "layout_scheme"
#define WEED_LEAF_MAX_REPEATS "max_repeats"
#define WEED_LEAF_DISABLED "disabled"
#define WEED_LEAF_NEW_DEFAULT "new_default"
#define WEED_LEAF_CHOICES "choices"
#define WEED_LEAF_YUV_SAMPLING "YUV_sampling"
#define WEED_LEAF_YUV_SUBSPACE "YUV_subspace"
/* plugin bootstrap (weed-effects.h:170-186, :195-230; weed.h:493-496) */
#define WEED_LEAF_FILTERS "filters"
#define WEED_LEAF_HOST_INFO "host_info"
#define WEED_LEAF_VERSION "version"
#define WEED_LEAF_PLUGIN_INFO "plugin_info"
#define WEED_LEAF_NAME "name"
#define WEED_LEAF_AUTHOR "author"
#define WEED_LEAF_PALETTE_LIST "palette_list"
#define WEED_LEAF_INIT_FUNC "init_func"
#define WEED_LEAF_DEINIT_FUNC "deinit_func"
#define WEED_LEAF_PROCESS_FUNC "process_func"
#define WEED_LEAF_IN_PARAMETER_TEMPLATES "in_param_tmpls"
#define WEED_LEAF_OUT_PARAMETER_TEMPLATES "out_param_tmpls"
#define WEED_LEAF_IN_CHANNEL_TEMPLATES "in_chan_tmpls"
#define WEED_LEAF_OUT_CHANNEL_TEMPLATES "out_chan_tmpls"
#define WEED_LEAF_GUI "gui"
#define WEED_LEAF_DEFAULT "default"
#define WEED_LEAF_MIN "min"
#define WEED_LEAF_MAX "max"
#define WEED_LEAF_PARAM_TYPE "param_type"
#define WEED_LEAF_COLORSPACE "colorspace"
#define WEED_LEAF_IS_TRANSITION "is_transition"
#define WEED_LEAF_LABEL "label"
#define WEED_LEAF_USE_MNEMONIC "use_mnemonic"
#define WEED_LEAF_DECIMALS "decimals"
#define WEED_LEAF_FILTER_API_VERSION "filter_api_version"
#define WEED_LEAF_WEED_API_VERSION "weed_api_version"
#define WEED_LEAF_GET_FUNC "weed_leaf_get_func"
#define WEED_LEAF_SET_FUNC "weed_leaf_set_func"
#define WEED_LEAF_DELETE_FUNC "weed_leaf_delete_func"
#define WEED_PLANT_NEW_FUNC "weed_plant_new_func"
#define WEED_PLANT_FREE_FUNC "weed_plant_free_func"
#define WEED_LEAF_NUM_ELEMENTS_FUNC "weed_leaf_num_elements_func"
#define WEED_LEAF_MALLOC_FUNC "weed_malloc_func"
#define WEED_LEAF_FREE_FUNC "weed_free_func"
typedef weed_error_t (*weed_default_getter_f)(weed_plant_t *plant, const char *key, void *value);
typedef weed_plant_t *(*weed_bootstrap_f)(weed_default_getter_f *, int32_t plugin_weed_min_api_version,
                                          int32_t plugin_weed_max_api_version, int32_t plugin_filter_min_api_version,
                                          int32_t plugin_filter_max_api_version);
typedef weed_error_t (*weed_process_f)(weed_plant_t *filter_instance, weed_timecode_t timestamp);
typedef weed_error_t (*weed_init_f)(weed_plant_t *filter_instance);
typedef weed_error_t (*weed_deinit_f)(weed_plant_t *filter_instance);
#endif

#endif
