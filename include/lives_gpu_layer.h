/* lives_gpu_layer.h -- the weed_layer_t seam of liblivesgpu.so.
 *
 * Same prototypes, argument meaning and failure behaviour as src/colourspace.h:377-423: the layer (a weed
 * plant of type WEED_PLANT_LAYER = 128, src/layers.h:14) is mutated IN PLACE -- pixel_data, rowstrides,
 * width (macropixels), height, current_palette, gamma_type, YUV leaves -- old pixel memory is released
 * with the host's allocator, and on failure the function returns FALSE and leaves the layer untouched
 * (the memfail: contract, src/colourspace.c:13906-13927).
 *
 * The library never links libweed: the host hands over its accessors once with lives_gpu_bind_weed()
 * (LiVES: the extern function pointers of libweed/weed.h:340-351 and its frame allocator).
 *
 * Symbols are exported as lives_gpu_<name>; define LIVES_GPU_DROP_IN before including this header to
 * get the reference names as macros (what a LiVES build that replaces the CPU bodies would do; see
 * INTEGRATION.md).
 *
 * Pixel data stays HOST memory at this seam (that is what every other part of LiVES expects): each call
 * uploads the planes it needs, runs the gfx950 kernels, downloads into freshly allocated host memory and
 * synchronises once before it returns (PCIe-bound).  A layer pinned with lives_gpu_layer_pin() keeps its planes
 * in HBM instead: the same calls then only enqueue kernels on the resident buffers -- no copy, no
 * synchronisation -- until lives_gpu_layer_sync() / _unpin() brings the pixels home.
 *
 * A call this library does not serve returns FALSE with the layer's pixels untouched and the caller's CPU
 * body takes over; a PINNED layer is first synchronised and unpinned in that case, so the CPU body reads
 * current bytes.  The reference's own first steps of convert_layer_palette_full (range switch, un-premultiply)
 * may already have run by then: the layer is in the state the reference has after them.
 *
 * Coverage (anything else returns FALSE with the layer untouched, lgpu_last_error() says why):
 *   convert_layer_palette[_full]  RGB24/BGR24/RGBA32/BGRA32/ARGB32 <-> each other (selector tree
 *                                 src/colourspace.c:12370-12556, LUT8 gamma inline); YUV420P/YVU420P/YUV422P ->
 *                                 those five (:13400-13560, LUT16 gamma fused when a target gamma is given); RGB -> YUV888 /
 *                                 YUVA8888 / YUV(A)444(4)P / UYVY / YUYV / YUV411 / YUV420P / YVU420P / YUV422P and those packed / 4:4:4
 *                                 planar / UYVY / YUYV / YUV411 palettes -> RGB (RGB -> YUV changes the gamma on the way as :12311-12332 decides: the 16-bit LUT inline for
 *                                 UYVY / YUYV, a gamma pass first for the others); the clamped <-> unclamped switch; the YUV -> YUV pairs of
 *                                 lgpu_yuv_repack (lives_gpu.h), YUV411 to and from the other YUV palettes and 4:2:0 / 4:2:2 planar -> YUV888 / YUVA8888 included
 *   gamma_convert_layer / _variant / gamma_convert_sub_layer, alpha_premult (RGB with alpha, YUVA8888, YUVA4444P), resize_layer / resize_layer_full
 *   (tgt_gamma = the fused LUT8 post-pass), letterbox_layer, unletterbox_layer (packed RGB palettes and YUV420P / YVU420P / YUV422P / YUV444P
 *   planes; palette hints see INTEGRATION.md), compact_rowstrides, create_empty_pixel_data, weed_layer_clear_pixel_data, calc_rowstrides (with the
 *   fixed-rowstride rule when leaf_get_flags is bound)
 *
 * The reference NAMES (convert_layer_palette, resize_layer_full, ...) are exported as real symbols by the small shim
 * lives_amd/liblivesgpu_dropin.so (csrc/dropin.c: one forwarding function per name), kept out of liblivesgpu.so itself so that a process
 * which also carries the CPU bodies (LiVES during a staged migration, the parity tests) has no duplicate symbols.
 */
#ifndef LIVES_GPU_LAYER_H
#define LIVES_GPU_LAYER_H
#include "lives_gpu_weed_abi.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef weed_plant_t lives_gpu_layer_t;   /* weed_layer_t */
typedef int lives_gpu_boolean;            /* LiVES `boolean`: TRUE 1 / FALSE 0 */

typedef struct {
  weed_leaf_get_f leaf_get;
  weed_leaf_set_f leaf_set;
  weed_leaf_num_elements_f leaf_num_elements;
  weed_leaf_delete_f leaf_delete;
  void *(*pixel_alloc)(size_t bytes);     /* frame allocator (LiVES: lives_calloc_safety); NULL -> calloc */
  void (*pixel_free)(void *);             /* LiVES: lives_free; NULL -> free */
} lives_gpu_weed_api;

/* prefs the reference reads on this path (src/preferences.h: apply_gamma, alpha_post, pb_quality, screen_gamma) */
typedef struct {
  int apply_gamma;      /* default 1 */
  int alpha_post;       /* default 0 */
  int pb_quality;       /* 1 LOW, 2 MED (default), 3 HIGH (render; same results as MED) */
  double screen_gamma;  /* default 1.4 (DEF_SCREEN_GAMMA) */
  int device;           /* HIP device ordinal, default 0 */
} lives_gpu_prefs;

int lives_gpu_bind_weed(const lives_gpu_weed_api *api);
int lives_gpu_set_prefs(const lives_gpu_prefs *prefs);
/* optional: weed_leaf_get_flags lets calc_rowstrides / create_empty_pixel_data honour rowstrides flagged LIVES_FLAG_CONST_VALUE (decoder plugins
   with fixed strides, src/colourspace.c:11268-11275, :11358-11363); a separate call so that lives_gpu_weed_api keeps its layout */
int lives_gpu_bind_leaf_get_flags(weed_leaf_get_flags_f leaf_get_flags);

lives_gpu_boolean lives_gpu_convert_layer_palette(lives_gpu_layer_t *layer, int outpl, int op_clamping);
lives_gpu_boolean lives_gpu_convert_layer_palette_full(lives_gpu_layer_t *layer, int outpl, int oclamping, int osampling,
                                                       int osubspace, int tgt_gamma);
lives_gpu_boolean lives_gpu_convert_layer_palette_with_sampling(lives_gpu_layer_t *layer, int outpl, int out_sampling);   /* colourspace.h:397 */
lives_gpu_boolean lives_gpu_gamma_convert_layer(int gamma_type, lives_gpu_layer_t *layer);
lives_gpu_boolean lives_gpu_gamma_convert_layer_variant(double file_gamma, int tgt_gamma, lives_gpu_layer_t *layer);      /* colourspace.h:392 */
lives_gpu_boolean lives_gpu_gamma_convert_sub_layer(int gamma_type, double fileg, lives_gpu_layer_t *layer, int x, int y,
                                                    int width, int height, lives_gpu_boolean may_thread);
void lives_gpu_alpha_premult(lives_gpu_layer_t *layer, int direction);
lives_gpu_boolean lives_gpu_resize_layer(lives_gpu_layer_t *layer, int width, int height, int interp, int opal_hint, int oclamp_hint);
/* colourspace.h:409-411; tgt_gamma: the LUT8 post-pass fused into the resize (src/colourspace.c:14718-14720, :15119-15127) */
lives_gpu_boolean lives_gpu_resize_layer_full(lives_gpu_layer_t *layer, int width, int height, int interp, int opal_hint, int oclamp_hint,
                                              int osamp_hint, int osubs_hint, int tgt_gamma);
/* which of the reference's two resize bodies resize_layer / resize_layer_full (and the scaling inside letterbox_layer / unletterbox_layer) follow:
   LIVES_GPU_RESIZE_PIXBUF (default) -- the gdk-pixbuf body (src/colourspace.c:15262-15322), bit-exact to gdk_pixbuf_scale_simple 2.42.8 for the palettes of
     its switch (RGB24, BGR24, RGBA32, BGRA32, YUV888, YUVA8888): alpha-weighted colours on 4-byte palettes, rowstride ALIGN4(width * channels), RGB layers
     come back tagged WEED_GAMMA_SRGB, no gamma pass.  In front of it runs the reference's palette resolution (get_resizable + the pre-conversion,
     :14869-14912; see lives_gpu_get_resizable below): a frame in a palette outside the switch is first converted -- YUV420P with an RGBA32 hint becomes RGBA32,
     ARGB32 with an RGBA32 hint becomes RGBA32 -- and then scaled; FALSE (the layer as the failed step left it, a pinned one synchronised and unpinned)
     only where the reference has no route either (e.g. YUV420P with a YUV420P hint: its LIVES_FATAL) or where its own post-check fails (:14916-14923);
   LIVES_GPU_RESIZE_POLYPHASE -- the swscale body's place (:14940-15259), the choice of a host "built with USE_SWSCALE"; libswscale is un-vendored and
     unpinned, so the arithmetic is this library's own spec "lgpu-polyphase-v1" (DESIGN.md), every palette, fused target gamma.
   A separate call so that lives_gpu_prefs keeps its layout.  Returns 0, -1 on a bad value. */
enum { LIVES_GPU_RESIZE_POLYPHASE = 0, LIVES_GPU_RESIZE_PIXBUF = 1 };
/* The planner's capability queries of the same header range (src/colourspace.h:400-407; callers src/nodemodel.c:131, :143, :210, :2961), answered for the
   bodies of THIS library so that a host which swapped the bodies plans for what will run:
   get_resizable (src/colourspace.c:14577-14669) -- the palette a frame is scaled in, with weed_palette_is_resizable() = the switch of the resize body in force
     (PIXBUF: RGB24 / BGR24 / RGBA32 / BGRA32 / YUV888 / YUVA8888, the reference's own rule without swscale, :2619-2643; POLYPHASE: packed RGB and planar YUV).
     Returns LIVES_RESULT_SUCCESS 1 / LIVES_RESULT_FAIL 0 (also where the reference ends in LIVES_FATAL: no resizable route).  resize_layer_full runs it in front
     of the body exactly as the reference does (:14869-14912): a YUV420P frame with an RGBA32 hint is converted, then scaled;
   get_tgt_gamma (:14736-14740); can_inline_gamma (:12128-12145) -- TRUE where this library folds the target gamma into the conversion kernel (RGB <-> RGB,
     4:2:0 / 4:2:2 planar -> RGB, RGB -> UYVY / YUYV); pconv_can_inplace (:12148-12157) -- TRUE where the conversion keeps the layer's pixel_data (UYVY <-> YUYV only:
     every other conversion of this library writes new planes). */
int lives_gpu_get_resizable(int *ppalette, int *pxpal, int *oclamp_hint, int *opal, int *pxopal, lives_gpu_boolean upscale);
int lives_gpu_get_tgt_gamma(int ipal, int opal);
lives_gpu_boolean lives_gpu_can_inline_gamma(int inpl, int opal);
lives_gpu_boolean lives_gpu_pconv_can_inplace(int inpl, int outpl);
int lives_gpu_set_resize_backend(int backend);
int lives_gpu_get_resize_backend(void);
lives_gpu_boolean lives_gpu_letterbox_layer(lives_gpu_layer_t *layer, int nwidth, int nheight, int width, int height, int interp,
                                            int tpal, int tclamp);
lives_gpu_boolean lives_gpu_unletterbox_layer(lives_gpu_layer_t *layer, int opwidth, int opheight, int top, int bottom, int left, int right);   /* colourspace.h:418 */
lives_gpu_boolean lives_gpu_compact_rowstrides(lives_gpu_layer_t *layer);                     /* colourspace.h:420 */
lives_gpu_boolean lives_gpu_weed_layer_clear_pixel_data(lives_gpu_layer_t *layer);            /* colourspace.h:399 */
lives_gpu_boolean lives_gpu_create_empty_pixel_data(lives_gpu_layer_t *layer, lives_gpu_boolean black_fill, lives_gpu_boolean may_contig);
int *lives_gpu_calc_rowstrides(int width, int pal, lives_gpu_layer_t *layer, int *nplanes);
/* THREADVAR(rowstride_alignment_hint) of the calling thread (src/colourspace.c:11285-11297; set by src/player.c:1355-1356, src/transcode.c:42,:369,
   src/effects-weed.c:2005-2007): a value >= 4 is the alignment of the NEXT plane allocation of this thread, -1 = compact rows until reset, 0 = default (32).
   A host that keeps the CPU bodies beside the GPU ones forwards its thread variable before a seam call and reads it back after. */
void lives_gpu_set_rowstride_alignment_hint(int hint);
int lives_gpu_get_rowstride_alignment_hint(void);

/* ---- device residency (optional): a pinned layer keeps the authoritative copy of its planes in HBM across the calls
   above, so a chain of layer ops crosses PCIe once per direction.  Between pin and sync the HOST bytes of pixel_data are
   stale (pointers, sizes, rowstrides and every other leaf stay exact).  The host pins a layer when it enters a run of
   GPU-served steps (e.g. the CONVERT chain of one plan step, src/nodemodel.c:2027-2089) and syncs / unpins it before CPU
   code reads the pixels.  State travels in the private leaf "host_gpu_resident" (host_* convention). */
int lives_gpu_layer_pin(lives_gpu_layer_t *layer);      /* upload the planes once, mark the layer resident */
int lives_gpu_layer_sync(lives_gpu_layer_t *layer);     /* download the current planes into pixel_data; stays pinned */
int lives_gpu_layer_unpin(lives_gpu_layer_t *layer);    /* sync, release the device copies, clear the leaf */
/* A layer whose planes ALREADY lie in device memory the caller owns (a hardware decoder's output surfaces, a frame an earlier pass left in HBM): pinned with
   those buffers as its resident planes -- no upload, no copy.  The library reads them in place (in-place seam calls write them), never frees them; keep them
   valid until the layer's pixel_data has been replaced by a seam call or the layer is synchronised / unpinned / forgotten.  producer_stream: where their
   contents were produced (NULL = the null stream); producer_done != 0: that work is known to be complete, nobody has to wait for it. */
int lives_gpu_layer_pin_device(lives_gpu_layer_t *layer, const void *const *planes_d, int nplanes, void *producer_stream, int producer_done);
/* weed_layer_copy(NULL, slayer) (the deep copy, src/layers.c:755-846) copies host bytes, which are stale while slayer is pinned.  Called on its result: dlayer (same
   palette, size, rowstrides; host planes of its own) becomes a pinned layer whose device planes are device-to-device copies of slayer's -- a pending program of slayer
   runs first, nothing crosses PCIe.  slayer not pinned: LGPU_OK, nothing done.  (A shallow copy shares the host planes and with them the device copies.) */
int lives_gpu_layer_copy(lives_gpu_layer_t *dlayer, lives_gpu_layer_t *slayer);
/* The host's word that every pixel of the layer's frame has alpha 255 (decoded video; a frame that was RGB24 or YUV before it became RGBA32): resize_layer[_full] /
   letterbox_layer on it -- recorded or eager -- then run the scalers' all-opaque instantiations (LGPU_INTERP_OPAQUE: the same bytes on such a frame, enlargements
   25-30 % faster).  The word stays on the layer (private leaf "host_gpu_opaque") until the host takes it back (on = 0); a frame that is not opaque gets wrong colours. */
int lives_gpu_layer_set_opaque(lives_gpu_layer_t *layer, int on);
/* ---- deferred execution on pinned layers (on by default).  The host bytes of a pinned layer are stale until lives_gpu_layer_sync(), so the seam calls of one
   track's plan step on an RGBA32 / BGRA32 frame -- convert_layer_palette (R <-> B), resize_layer[_full] (gdk-pixbuf body), letterbox_layer, livesgpu_fx.so's
   "chroma blend" in place, gamma_convert_layer -- are RECORDED on the plane (every leaf changes as in the eager call) and run as ONE launch of the fused chain
   kernel: by themselves when anyone needs the pixels (another seam call, an effect, sync / unpin), or for all tracks of a tick together when the host calls
   lives_gpu_layers_flush(layers, n) after its plan steps have returned (programs of equal shape share the launch: this is lgpu_chain, reached through the
   reference's own calls).  Results are those of the eager calls, bit for bit (tests/test_deferred.py).  What a stage can refuse is checked when it is recorded;
   device failures at run time surface at the flush / sync that runs the program.  lives_gpu_set_deferred(0) launches every call by itself (returns the old value). */
int lives_gpu_set_deferred(int on);
int lives_gpu_layers_flush(lives_gpu_layer_t *const *layers, int nlayers);
/* counters since load: [0] stages recorded, [1] fused chain launches made for pending programs, [2] programs (tracks) those carried, [3] programs run stage by stage */
void lives_gpu_deferred_stats(unsigned long long out[4]);
/* (for livesgpu_fx.so) record an in-place "chroma blend" of the pending plane dst_host with the resident plane layer2_host; 1 = recorded, 0 = run the kernel */
int lives_gpu_deferred_blend_chroma(const void *dst_host, int orow, int width, int height, int palette, const void *layer2_host, int irow2, int bf);
/* the host is about to free or replace the pixel_data of a pinned layer itself (weed_layer_pixel_data_free, an error path): release the
   device copies without a download and clear the leaf.  Device copies are keyed by host plane pointer, so this (or unpin) MUST precede
   any release of a pinned layer's planes that does not go through this library. */
int lives_gpu_layer_forget(lives_gpu_layer_t *layer);
/* optional allocator pair for lives_gpu_weed_api.pixel_alloc / pixel_free: page-locked (hipHostMalloc) zeroed memory, so frames cross PCIe by DMA at
   link rate; pageable frames still work (they go through pinned staging chunks inside lgpu_upload / lgpu_download) */
void *lives_gpu_pinned_calloc(size_t bytes);
void lives_gpu_pinned_free(void *p);
/* device copy of a pinned layer's plane by its host plane pointer, or NULL (used by livesgpu_fx.so: effects on pinned layers read and write
   HBM directly, no PCIe traffic; the host bytes stay stale until lives_gpu_layer_sync()) */
void *lives_gpu_resident_lookup(const void *host_plane, size_t min_bytes);
/* The same for code that enqueues on the calling thread's stream instead of the null stream (every host thread that enters the seam has a stream of its
   own; livesgpu_fx.so runs its effects there, so effects of different tracks overlap on the device like the seam's own calls):
   acquire = the device copy, with the calling thread's stream ordered behind the plane's last writer (write != 0: and behind every reader since);
   release = tell the table that a read / a write of the plane has been enqueued on that stream.  Between the two the caller enqueues its work on
   lives_gpu_thread_stream().  lives_gpu_stream_follow(s) orders the calling thread's stream behind everything enqueued so far on s (NULL = the null
   stream): for state a caller keeps on the device across calls that may arrive on different threads. */
void *lives_gpu_thread_stream(void);
void lives_gpu_stream_follow(void *other_stream);
void *lives_gpu_resident_acquire(const void *host_plane, size_t min_bytes, int write);
void lives_gpu_resident_release(const void *host_plane, int write);
void lives_gpu_transfer_stats(unsigned long long *h2d_bytes, unsigned long long *d2h_bytes);   /* PCIe bytes moved by the seam so far */

#ifdef LIVES_GPU_DROP_IN
#define convert_layer_palette lives_gpu_convert_layer_palette
#define convert_layer_palette_full lives_gpu_convert_layer_palette_full
#define convert_layer_palette_with_sampling lives_gpu_convert_layer_palette_with_sampling
#define gamma_convert_layer_variant lives_gpu_gamma_convert_layer_variant
#define resize_layer_full lives_gpu_resize_layer_full
#define unletterbox_layer lives_gpu_unletterbox_layer
#define compact_rowstrides lives_gpu_compact_rowstrides
#define weed_layer_clear_pixel_data lives_gpu_weed_layer_clear_pixel_data
#define gamma_convert_layer lives_gpu_gamma_convert_layer
#define gamma_convert_sub_layer lives_gpu_gamma_convert_sub_layer
#define alpha_premult lives_gpu_alpha_premult
#define resize_layer lives_gpu_resize_layer
#define letterbox_layer lives_gpu_letterbox_layer
#define create_empty_pixel_data lives_gpu_create_empty_pixel_data
#define calc_rowstrides lives_gpu_calc_rowstrides
#define get_resizable lives_gpu_get_resizable
#define get_tgt_gamma lives_gpu_get_tgt_gamma
#define can_inline_gamma lives_gpu_can_inline_gamma
#define pconv_can_inplace lives_gpu_pconv_can_inplace
#endif

#ifdef __cplusplus
}
#endif
#endif
