/* lives_gpu.h -- C ABI of liblivesgpu.so: the MI355X (gfx950) per-frame effects engine for LiVES.
 *
 * Frame-level entry points.  Plain pointers and sizes, no C++ / torch types.  Every `*_d` pointer is a
 * DEVICE pointer (HBM); `stream` is a hipStream_t passed as void* (NULL = the default stream).  All
 * calls are asynchronous on `stream`; they return LGPU_OK (0) or a negative LGPU_E_* code and never
 * touch the frame on failure.  There is NO CPU fallback: without a HIP device every compute entry
 * point fails with LGPU_E_NODEVICE.
 *
 * Each function names the reference interface it replaces (file:line under the LiVES tree).  The
 * weed_layer_t-level seam (convert_layer_palette() & co.) that sits on top of these is declared in
 * lives_gpu_layer.h; the weed plugin seam is lives_amd/csrc/fx_plugin.c (weed_setup).
 *
 * Geometry conventions are the reference's: `width` in PIXELS, rowstrides in BYTES
 * (src/colourspace.c:11252 calc_rowstrides: ALIGN_CEIL(width * psize, 32) by default).
 */
#ifndef LIVES_GPU_H
#define LIVES_GPU_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define LGPU_ABI_VERSION 1

enum {
  LGPU_OK = 0,
  LGPU_E_NODEVICE = -1,   /* no HIP device / runtime error at init */
  LGPU_E_BADARG = -2,
  LGPU_E_UNSUPPORTED = -3,
  LGPU_E_HIP = -4,        /* a HIP call failed; see lgpu_last_error() */
  LGPU_E_NOMEM = -5,
  LGPU_E_STATE = -6,      /* the object is out of step after an earlier half-done call (lgpu_stepper): destroy it */
  LGPU_E_TIMEOUT = -7     /* a peer did not arrive in time (lgpu_dist_comm_create_timeout, lgpu_stepper_wait); lgpu_last_error() says what was being waited for */
};

/* ---- runtime ----------------------------------------------------------------------------------- */
int lgpu_abi_version(void);
/* bind the calling thread to `device` and upload the conversion tables (idempotent, thread-safe) */
int lgpu_init(int device);
const char *lgpu_last_error(void);
int lgpu_device_count(void);
int lgpu_current_device(int *device);     /* hipGetDevice / hipSetDevice of the calling thread */
int lgpu_set_device(int device);
/* device memory + copies for hosts without their own allocator (the layer seam uses these) */
int lgpu_malloc(void **ptr_d, size_t bytes);
/* diagnostics: the nth lgpu_malloc from now (1 = the next one) fails with LGPU_E_NOMEM; 0 disarms.  Used by the tests of the
   "failure leaves the layer untouched" contract (memfail:, src/colourspace.c:13906-13927) to fail a call after it has started. */
int lgpu_debug_fail_alloc(int nth);
int lgpu_free(void *ptr_d);
/* stream-ordered allocation from the device's default pool (hipMallocAsync / hipFreeAsync): the block may be used by work enqueued on `stream` after the
   call, and is handed out again only to work enqueued after the free.  No device synchronisation on either side (hipFree waits for the device to drain);
   what the layer seam's resident planes come from.  lgpu_debug_fail_alloc counts these too. */
int lgpu_malloc_ordered(void **ptr_d, size_t bytes, void *stream);
int lgpu_free_ordered(void *ptr_d, void *stream);
/* streams and events for hosts that enqueue from several threads.  Every entry point below takes a `stream` (NULL = the null stream).  nonblocking = 0:
   the stream is ordered against work on the null stream (and every launch on it pays for that); 1: it is not, and the caller orders hand-overs with
   events.  The layer seam (lives_gpu_layer.h) gives every host thread a non-blocking stream of its own and orders work on a resident plane across
   threads -- and against the null stream the weed plugin uses -- with an event at each hand-over (LiVES' plan steps run on pool threads: src/threading.c). */
int lgpu_stream_create(void **stream_out, int nonblocking);
int lgpu_stream_destroy(void *stream);
int lgpu_event_create(void **event_out);
int lgpu_event_destroy(void *event);
int lgpu_event_record(void *event, void *stream);
int lgpu_stream_wait_event(void *stream, void *event);
int lgpu_upload(void *dst_d, const void *src_h, size_t bytes, void *stream);
int lgpu_download(void *dst_h, const void *src_d, size_t bytes, void *stream);
/* page-locked, zeroed host memory for frames (DMA at link rate); lgpu_upload / lgpu_download also accept pageable memory: uploads of it go through
   two pinned staging chunks per host thread (22 GB/s against 2.7 GB/s plain), downloads go direct (the runtime's pageable path is the faster one there);
   a download is complete after lgpu_sync(stream) */
void *lgpu_pinned_calloc(size_t bytes);
void lgpu_pinned_free(void *p);
int lgpu_copy(void *dst_d, const void *src_d, size_t bytes, void *stream);      /* device to device */
int lgpu_fill(void *dst_d, int byte, size_t bytes, void *stream);
int lgpu_sync(void *stream);
int lgpu_stream_query(void *stream);      /* 1 = everything enqueued so far has completed, 0 = not yet (never blocks), < 0 = error */
/* rows of row_bytes bytes between two pitched device buffers (what compact_rowstrides :14439 and the cut of unletterbox_layer :15612-15615 do) */
int lgpu_copy_rows(void *dst_d, int orow, const void *src_d, int irow, int row_bytes, int rows, void *stream);
/* n repetitions of a plen-byte (1..8) pattern at the start of every row: a palette's black as blank_pixel / blank_row paint it
   (src/colourspace.c:11123-11210), for weed_layer_clear_pixel_data on device-resident planes */
int lgpu_fill_pattern(void *dst_d, int rowstride, const uint8_t *pattern, int plen, int n, int rows, void *stream);

/* ---- host-side table builders (pure CPU, usable without a device) -------------------------------- */
/* conversion tables; replaces init_RGB_to_YUV_tables / init_YUV_to_RGB_tables (src/colourspace.c:851-1105).
   which: bit0 = unclamped, bit1 = BT.709.  rgb2yuv[9*256], yuv2rgb[5*256] (either may be NULL). */
int lgpu_conversion_tables(int which, int32_t *rgb2yuv, int32_t *yuv2rgb);
/* replaces create_gamma_lut8 (src/colourspace.c:655-736), same return convention: 1 = LUT written,
   0 = no conversion needed (the reference returns NULL) */
int lgpu_gamma_lut8(double file_gamma, int gamma_from, int gamma_to, double screen_gamma, uint8_t lut[256]);
/* chroma blend, translucent pixels (simple_blend.c:137-145): constants with (c * k2[a]) >> 16 == (uint8_t)((float)c * alpha)
   and (c * k1[a]) >> 16 == (uint8_t)((float)c * (1 - alpha)) for every byte c, alpha = (float)a / 255.; proven for all
   operand pairs while the table is built (HOST function).  a = 255: 65536 (the reference does not scale opaque pixels) */
int lgpu_alpha_scalers(uint32_t k2[256], uint32_t k1[256]);
/* create_gamma_lut (src/colourspace.c:738-808): 65536 x uint16; returns 1 if a LUT was produced (HOST function) */
int lgpu_gamma_lut16(double file_gamma, int gamma_from, int gamma_to, double screen_gamma, uint16_t *lut16);
/* rowstride rule; replaces calc_rowstrides (src/colourspace.c:11252-11366) for an explicit alignment
   (0 = RS_ALIGN_DEF 32, -1 = compact).  Returns the number of planes, fills rowstrides[4]. */
int lgpu_calc_rowstrides(int width, int palette, int alignment, int rowstrides[4]);

/* ---- K1: packed RGB <-> RGB swizzles ---------------------------------------------------------------
   replaces convert_swap3_frame ... convert_swapprepost_frame (src/colourspace.c:9259-10577) as picked
   by the selector tree of convert_layer_palette_full (:12370-12556). */
enum {
  LGPU_SWAP3, LGPU_SWAP4, LGPU_SWAP3ADDPOST, LGPU_SWAP3ADDPRE, LGPU_SWAP3POSTALPHA, LGPU_SWAP3PREALPHA,
  LGPU_ADDPOST, LGPU_ADDPRE, LGPU_SWAP3DELPOST, LGPU_DELPOST, LGPU_DELPRE, LGPU_SWAP3DELPRE, LGPU_SWAPPREPOST
};
/* lut8: HOST pointer to 256 bytes or NULL (travels as a kernel argument).  src_d == dst_d is allowed
   when input and output pixel sizes are equal. */
int lgpu_swizzle(int op, int alpha_first, const uint8_t *src_d, int irow, uint8_t *dst_d, int orow,
                 int width, int height, const uint8_t *lut8, void *stream);

/* ---- K6: gamma LUT apply, in place; replaces gamma_convert_layer_thread (src/colourspace.c:14034-14060)
   on the sub-rectangle of gamma_convert_sub_layer (:14069-14143).  psize 3 or 4. */
int lgpu_gamma_apply(uint8_t *pix_d, int rowstride, int x, int y, int width, int height, int psize,
                     int alpha_first, const uint8_t *lut8, void *stream);

/* ---- K9: alpha pre/un-multiply, in place; replaces alpha_premult (src/colourspace.c:11968-12105) for the
   packed 4-byte RGB palettes.  un = 1 is LIVES_DIRECTION_REVERSE (the `unal` table). */
int lgpu_alpha_premult(uint8_t *pix_d, int rowstride, int width, int height, int alpha_first, int un,
                       void *stream);

/* K9b: alpha_premult on YUVA8888 (589, planes_d[0]) / YUVA4444P (545, four planes), in place (src/colourspace.c:11995-12096).  clamped = the layer's
   YUV_clamping is CLAMPED: the reference then goes through the tables unalcy / alcy / unalcuv / alcuv of init_unal (:1141-1160; lgpu_premult_yuv_tables
   builds them on the host, each [256 alpha][256 value]); unclamped layers take the RGB arithmetic on Y, U and V. */
int lgpu_premult_yuv_tables(uint8_t *unalcy, uint8_t *alcy, uint8_t *unalcuv, uint8_t *alcuv);
int lgpu_alpha_premult_yuva(uint8_t *const planes_d[4], const int rowstrides[4], int width, int height, int palette, int clamped, int un,
                            void *stream);

/* ---- K2: planar YUV 4:2:0 / 4:2:2 -> packed RGB; replaces convert_yuv420p_to_rgb_frame
   (src/colourspace.c:3260-3904; the BGR / ARGB twins :3927-5114 share the maths).
   out_order 0 = RGB(A), 1 = BGR(A), 2 = ARGB; opsize 3 or 4; which_tables as lgpu_conversion_tables;
   pb_quality 1 (LOW) / 2 (MED, default) / 3 (HIGH, the render setting: its float rounding in _spc_rnd :832-835 is
   bit-identical to MED once clamped, see yuv.hip).
   lut8 (HOST, may be NULL) fuses the gamma_convert_layer() pass that follows in BASELINE config 2.
   flags: LGPU_YUV_FIX_EDGES = write the evident intent into the last row instead of replicating the
   reference's row-0-luma behaviour. */
#define LGPU_YUV_FIX_EDGES 1
int lgpu_yuv420p_to_rgb(const uint8_t *y_d, const uint8_t *u_d, const uint8_t *v_d, const int istrides[3],
                        long u_size, long v_size, uint8_t *dst_d, int orow, int width, int height,
                        int opsize, int out_order, int is_422, int which_tables, int pb_quality,
                        const uint8_t *lut8, int flags, void *stream);
/* the same conversion for a batch of frames of one geometry (the frames of the live tracks of a multitrack timeline, <= LGPU_CHAIN_MAX_TRACKS)
   in ONE launch: at 1920x1080 a single frame is bounded by the launch floor, a batch by the kernel. */
typedef struct { const uint8_t *y_d, *u_d, *v_d; uint8_t *dst_d; } lgpu_yuv_frame;
int lgpu_yuv420p_to_rgb_batch(int nframes, const lgpu_yuv_frame *frames, const int istrides[3], long u_size, long v_size, int orow,
                              int width, int height, int opsize, int out_order, int is_422, int which_tables, int pb_quality,
                              const uint8_t *lut8, int flags, void *stream);
/* launch shape of the 4:2:0 kernel for aligned rows (yuv.hip, k_yuv420p_to_rgb_s): chroma columns per lane cell (1, 2 or 4; 0 sends every launch
   through the one-column kernel), threads per workgroup (256 / 512 / 1024), resident groups of 256 threads per CU; -1 keeps a value.  Results do not
   depend on it (tests walk every setting); process-wide, not meant to change while conversions are in flight. */
int lgpu_yuv420_tuning(int cell_columns, int block, int groups_per_cu);
/* Launch-shape / ablation switches by name ("PBH_TH", "PBH_ALIGNED", "PB_NO_PAIRS", "GCK_TH", "CHAIN_SPARE_WGS", ...: the LGPU_<NAME> environment variables
   without the prefix).  The environment is read ONCE, at the library's first launch; afterwards only this call changes a switch (value < 0 clears it), so no
   launch path ever calls getenv() beside a host that calls setenv().  Results do not depend on any of them (the tests walk them).  lgpu_tuning_get: the
   current value, -1 when unset or unknown. */
int lgpu_tuning_set(const char *name, int value);
int lgpu_tuning_get(const char *name);
/* test hook: the four clamped YUVA premultiply tables (order of lgpu_premult_yuv_tables, 4 x 65,536 bytes) as the DEVICE arithmetic of lgpu_alpha_premult_yuva evaluates
   them -- the kernel is table-free; tests compare this with the host tables byte for byte */
int lgpu_debug_premult_yuv_tables_device(uint8_t *out_host);
/* test hook: entries in the scaler's table cache (bounded: LGPU_PB_CACHE_MAX / lgpu_tuning_set("PB_CACHE_MAX", n), default 64; least recently used goes first) */
int lgpu_debug_pixbuf_cache_entries(void);
/* test hook: the scaler's five-operation reciprocal (pixbuf.hip: pb_recip) against the IEEE division 1.0 / (double)a for every integer a of [lo, hi), hi <= 2^24;
   *mismatches = how many differ (0 over the whole range: tests/test_pixbuf_scale.py) */
int lgpu_debug_recip_check(uint32_t lo, uint32_t hi, unsigned long long *mismatches);
/* the same conversion with the reference's 16-bit indexed gamma LUT fused in, as convert_yuv420p_to_rgb_frame does when it
   is handed a target gamma (:3274-3283; xyuv2rgb_with_gamma :2386-2390): c = lut16[CLAMP16biti(sum >> 8)] >> 8.
   lut16_d: DEVICE pointer to 65536 uint16 (build on the host with lgpu_gamma_lut16, upload once, reuse). */
int lgpu_yuv420p_to_rgb_lut16(const uint8_t *y_d, const uint8_t *u_d, const uint8_t *v_d, const int istrides[3],
                              long u_size, long v_size, uint8_t *dst_d, int orow, int width, int height,
                              int opsize, int out_order, int is_422, int which_tables, int pb_quality,
                              const uint16_t *lut16_d, int flags, void *stream);

/* ---- K8: letterbox; replaces the canvas fill + blit of letterbox_layer (src/colourspace.c:15343-15567;
   fill :11109-11119).  Writes every pixel of the nwidth x nheight canvas exactly once. */
int lgpu_letterbox(const uint8_t *src_d, int irow, int width, int height, uint8_t *dst_d, int orow,
                   int nwidth, int nheight, int psize, const uint8_t black_pixel[4], void *stream);
/* same with explicit offsets: the chroma planes of planar palettes sit at (int)(offs * plane_ratio) (:15553-15556) */
int lgpu_letterbox_at(const uint8_t *src_d, int irow, int width, int height, uint8_t *dst_d, int orow,
                      int nwidth, int nheight, int psize, const uint8_t black_pixel[4], int offs_x, int offs_y,
                      void *stream);
/* the bars alone: black on every canvas pixel outside the width x height rectangle at (ox, oy) -- for callers that let lgpu_resize write the inner frame
   straight into the canvas (dst_d + oy * orow + ox * psize), so that the resized frame never exists on its own (what lives_gpu_letterbox_layer does) */
int lgpu_letterbox_bars(uint8_t *dst_d, int orow, int nwidth, int nheight, int psize, const uint8_t black_pixel[4], int ox, int oy, int width,
                        int height, void *stream);

/* ---- K7: resize.  Replaces the sws_scale() call of resize_layer_full (src/colourspace.c:14711, setup
   :14940-15259).  PARITY UNPINNED: libswscale is neither vendored nor version-pinned by the reference;
   this follows the repo's own spec "lgpu-polyphase-v1" (DESIGN.md).  interp = LiVESInterpType
   (GdkInterpType values: 0 NEAREST/"fast", 2 BILINEAR, 3 HYPER/"best").  psize 4, 3 or 1.
   lut8 (HOST, may be NULL) is the fused post-pass of :14718-14720. */
int lgpu_resize(const uint8_t *src_d, int irow, int sw, int sh, uint8_t *dst_d, int orow, int dw, int dh,
                int psize, int interp, const uint8_t *lut8, void *stream);
/* ---- K7, pixbuf arithmetic: the reference's OTHER resize body -- resize_layer_full without swscale scales through
   lives_pixbuf_scale_simple == gdk_pixbuf_scale_simple (src/colourspace.c:15262-15322, call :15295), and the compositor scales its layers with the
   same call (lives-plugins/weed-plugins/gdk/compositor.c:263-265).  PINNED: bit-exact to gdk-pixbuf 2.42.8's output (tests/golden/pixbuf_scale.npz,
   made from the runtime library by oracle/ref/gen_golden_pixbuf.py).  channels 3 = pixbuf without alpha (RGB24 / BGR24 / YUV888 layers), 4 = with
   alpha (RGBA32 / BGRA32 / YUVA8888: colours are weighted by alpha, colourspace.c:14219-14225).  interp = LiVESInterpType: 0 NEAREST, 2 BILINEAR,
   3 HYPER.  Same size = plain copy, as the library does.  LGPU_E_UNSUPPORTED for reductions beyond ~30:1 (the library's two-step scaler). */
int lgpu_pixbuf_scale(const uint8_t *src_d, int irow, int sw, int sh, uint8_t *dst_d, int orow, int dw, int dh, int channels, int interp,
                      void *stream);
/* the per-phase integer weight tables that call uses (host, for tests / inspection): table = [16 y phases][16 x phases][n_y][n_x], each summing to
   65536; xoff / yoff = 16.16 position of destination pixel 0.  table may be NULL to query the sizes. */
/* the same for nframes (1 .. LGPU_CHAIN_MAX_TRACKS) frames of ONE geometry in ONE launch: src_d / dst_d are host arrays of device pointers.  The unit of work of the
   path is a batch of independent frames, one per live clip track (src/effects-weed.c:1850-2425 runs an instance once per track per tick; compositor.c:262-266
   scales every layer of a frame): the weight tables, the per-launch cost of thousands of small workgroups and the launch itself are paid once.  Results are those
   of nframes single calls, bit for bit.  Alignment requirements apply to the least aligned frame. */
/* what lgpu_pixbuf_scale would answer for this geometry, nothing launched (the weight table is built and cached): LGPU_OK / LGPU_E_BADARG / LGPU_E_UNSUPPORTED */
int lgpu_pixbuf_scale_check(int sw, int sh, int dw, int dh, int channels, int interp, void *stream);
int lgpu_pixbuf_scale_batch(const uint8_t *const *src_d, uint8_t *const *dst_d, int nframes, int irow, int sw, int sh, int orow, int dw, int dh, int channels,
                            int interp, void *stream);
int lgpu_pixbuf_weights(int interp, int sw, int sh, int dw, int dh, int *n_x, int *n_y, int *xoff, int *yoff, int32_t *table, size_t table_ints);
/* the filter bank the kernel uses (host, for tests / inspection): kernel 0 triangle, 1 cubic(0,.6), 2 lanczos3 */
int lgpu_make_filter(int srcn, int dstn, int kernel, int *ntaps, int32_t *pos, int16_t *coef, int maxtaps);

/* ---- B1: 5x5 separable gaussian (no reference loop exists -- BASELINE config 4; build-defined spec) */
int lgpu_gauss5(const uint8_t *src_d, int irow, uint8_t *dst_d, int orow, int width, int height, int psize,
                void *stream);

/* ---- F1..F5: the built-in weed effect pixel loops ----------------------------------------------------- */
/* "chroma blend": lives-plugins/weed-plugins/simple_blend.c:117-150 */
int lgpu_blend_chroma(const uint8_t *src1_d, int irow1, const uint8_t *src2_d, int irow2, uint8_t *dst_d,
                      int orow, int width, int height, int psize, int alpha_first, int bf, void *stream);
/* "luma overlay" (1) / "luma underlay" (2) / "negative luma overlay" (3) / "averaged luma overlay" (4):
   simple_blend.c:151-194.  pal_order 0 RGB.., 1 BGR.., 2 ARGB. */
int lgpu_blend_luma(int type, const uint8_t *src1_d, int irow1, const uint8_t *src2_d, int irow2,
                    uint8_t *dst_d, int orow, int width, int height, int psize, int pal_order, int thresh,
                    void *stream);
/* blend_multiply .. blend_burn (type 0..6): lives-plugins/weed-plugins/multi_blends.c:24-168 */
int lgpu_blend_multi(int type, const uint8_t *src1_d, int irow1, const uint8_t *src2_d, int irow2,
                     uint8_t *dst_d, int orow, int width, int height, int is_bgr, int bf, void *stream);
/* "colorkey": lives-plugins/weed-plugins/scripts/colorkey.script <process> */
int lgpu_colorkey(const uint8_t *src0_d, int irow0, const uint8_t *src1_d, int irow1, uint8_t *dst_d,
                  int orow, int width, int height, int is_bgr, double delta, double opac,
                  int col_r, int col_g, int col_b, void *stream);
/* K4: packed RGB family -> YUV family (src/colourspace.c:5129-6440; pixel maths :2119-2192), PB_QUALITY_MED rounding.
   in_order 0 RGB, 1 BGR, 2 ARGB; in_alpha: 4-byte source pixels (implied for ARGB).
   out_fmt  0 packed YUV888 / YUVA8888 (out_alpha)     convert_{rgb,bgr,argb}_to_yuv_frame
            1 planar YUV444P / YUVA4444P (out_alpha)   convert_{rgb,bgr,argb}_to_yuvp_frame
            2 UYVY, 3 YUYV                             convert_{rgb,bgr,argb}_to_{uyvy,yuyv}_frame (no gamma LUT)
            4 YUV420P, 5 YUV422P                       convert_{rgb,bgr}_to_yuv420_frame
   dst_d / orow: one entry per destination plane.  which_tables: bit 0 unclamped, bit 1 BT.709 (4:2:0 / 4:2:2 only --
   the reference's other entry points are YCbCr only).  The reference's sampling is kept: U of the first and V of the
   second pixel of a pair, YUYV without the upper chroma clamp, 4:2:0 chroma row k = avg_chroma(row 2k+2, row 2k+1).
   LGPU_E_UNSUPPORTED for ARGB32 -> 4:2:0 / 4:2:2 (the reference reads the wrong bytes there). */
int lgpu_rgb_to_yuv(const uint8_t *src_d, int irow, int width, int height, int in_order, int in_alpha,
                    uint8_t *const dst_d[4], const int orow[4], int out_fmt, int out_alpha, int which_tables,
                    void *stream);
/* the same for out_fmt 2 (UYVY) / 3 (YUYV) with the 16-bit gamma LUT inline: rgb2uyvy_with_gamma / rgb2yuyv_with_gamma (src/colourspace.c:2146-2159, :2194-2207), what
   those entry points run when convert_layer_palette_full changes the gamma on the way (:12565-12580, :12646-12660, :12722-12736, :12796-12810, :12869-12883).
   lut16_d: the 65536 entries of lgpu_gamma_lut16() in device memory.  YCbCr tables only, as in the reference. */
int lgpu_rgb_to_yuv_lut16(const uint8_t *src_d, int irow, int width, int height, int in_order, int in_alpha, uint8_t *dst_d, int orow, int out_fmt,
                          int clamping_unclamped, const uint16_t *lut16_d, void *stream);
/* K3: YUV family -> packed RGB family (src/colourspace.c:2750-3258, :6616-7102, :7200-7498; pixel maths :2345-2459).
   in_fmt 0 packed YUV888 / YUVA8888 (in_alpha), 1 planar YUV444P / YUVA4444P (in_alpha), 2 UYVY, 3 YUYV (width in
   pixels); planar 4:2:0 / 4:2:2 sources: lgpu_yuv420p_to_rgb.  out_order 0 RGB, 1 BGR, 2 ARGB; out_alpha: 4-byte
   output (implied for ARGB); alpha = source alpha or 255.  which_tables as above (BT.709 for in_fmt 0 only).
   LGPU_E_UNSUPPORTED for planar -> ARGB32 / BGR24 (reference row arithmetic broken, :7475-7476, :7313). */
int lgpu_yuv_to_rgb(const uint8_t *const src_d[4], const int irow[4], int width, int height, int in_fmt, int in_alpha,
                    uint8_t *dst_d, int orow, int out_order, int out_alpha, int which_tables, void *stream);

/* K3b: YUV411 (u2 y0 y1 v2 y2 y3: 4 pixels in 6 bytes) -> RGB24 / RGBA32 / BGR24 / BGRA32 / ARGB32.
 * Replaces convert_yuv411_to_rgb_frame / _bgr_frame / _argb_frame (src/colourspace.c:8305-8620; dispatcher :13755-13795).
 * width_mp = macropixels per row (the layer's width leaf); the source is compact rows of width_mp * 6 bytes, as the reference
 * walks it; YCbCr subspace only.  Reference behaviour kept: the alpha byte of pixels 4j+2, 4j+3 (j < width_mp - 1)
 * is not written; the bgr variant stores the row's first pixel and last two pixels in R,G,B order. */
int lgpu_yuv411_to_rgb(const uint8_t *src_d, int width_mp, int height, uint8_t *dst_d, int orow, int out_order, int out_alpha,
                       int clamping_unclamped, void *stream);

/* K4b: RGB24 / RGBA32 / BGR24 / BGRA32 / ARGB32 -> YUV411.  Replaces convert_rgb_to_yuv411_frame / _bgr_ / _argb_
 * (src/colourspace.c:6499-6615, rgb2_411 :2322-2343; dispatcher :12627-12632, :12705, :12779, :12852, :12925).
 * width in pixels (width % 4 pixels on the right are dropped); dst_d receives compact rows of (width >> 2) * 6 bytes. */
int lgpu_rgb_to_yuv411(const uint8_t *src_d, int irow, int width, int height, int in_order, int in_alpha, uint8_t *dst_d,
                       int clamping_unclamped, void *stream);
/* diagnostics: cavgc, the clamped chroma-average table of init_average (src/colourspace.c:190-216), as this device computes it (the kernels use
   fmaf(fa[x] + fa[y], 0.4375f, 128.f) with fa(x) = (float)((x - 128) * 255. / 244.): one rounding of the reference's exactly representable double) */
int lgpu_chroma_average_table(uint8_t out[65536]);
/* K5: clamped <-> unclamped switch, in place: switch_yuv_clamping_and_subspace (src/colourspace.c:10929-11090) with the
   tables of init_YUV_to_YUV_tables (:1108-1139; one table set serves YCbCr and BT.709 -- the reference does no subspace
   maths).  palette 588 YUV888, 589 YUVA8888 (alpha untouched), 544 / 545 / 522 / 512 / 513 planar, 564 UYVY, 565 YUYV.
   As in the reference every byte of height * rowstride is mapped (row padding included; packed YUV888 is walked as one
   Y,U,V stream over the whole buffer, so byte roles follow the buffer offset mod 3). */
int lgpu_yuv_switch_clamping(uint8_t *const planes_d[4], const int rowstrides[4], int palette, int height, int to_unclamped,
                             void *stream);
/* K5b: YUV -> YUV repacks, the non-RGB half of convert_layer_palette_full's matrix (src/colourspace.c:12937-13750 dispatches
   to :7104-7198, :7500-7753, :7800-7971, :9198-9257, :10517-10639 and the K1 addpost / delpost pair).  Palettes are
   WEED_PALETTE_* numbers, width in pixels.  Taken: 444P / 4444P -> 888 / 8888 / 4444P / 444P / 420P / (compact rows) UYVY / YUYV;
   888 -> 444P / 8888; 8888 -> 888; 888 / 8888 -> (compact destination) 420P / 422P / UYVY / YUYV; 420P -> 422P / (compact chroma) UYVY / YUYV; UYVY <-> YUYV; UYVY / YUYV -> 444P / 4444P
   (equal plane strides) / 888 / 8888 / (compact rows) 420P / 422P.  LGPU_E_UNSUPPORTED for every other pair or layout: there the
   reference function overruns its buffers, mixes up its strides or leaves a result that depends on what the destination
   held before (DESIGN.md "YUV -> YUV"), and the caller keeps its CPU body.  Bytes the reference does not write are not
   written.  clamping_unclamped picks the chroma averaging table (init_average :190-216).
   K5d: 420P / YVU420P (chroma pointers in U, V order) / 422P -> 888 / 8888 (convert_quad_chroma_packed :10715-10808, convert_double_chroma_packed :10811-10873): the
   only pairs that read `sampling` (the source's WEED_YUV_SAMPLING_*: 0 = JPEG / default -> plain means, else the 3:1 / 1:3 forms).  The destination must be the zeroed
   frame create_empty_pixel_data() hands the reference: the last odd row's chroma and several alpha bytes are never written.  Even height for 4:2:0.
   K5c, the 4:1:1 pairs (palette 595; width in pixels, a multiple of 4): YUV411 -> 888 / 8888 / 444P / 4444P / UYVY / YUYV / 422P / 420P / YVU420P
   (src/colourspace.c:8622-9146) and 444P / 4444P / UYVY / YUYV / 888 / 8888 / 420P / 422P -> YUV411 (:7755-7798, :7973-8033, :8272-8303, :9148-9196).
   The reference functions take no destination rowstride and walk their 4:1:1 side as one stream: here too both sides are COMPACT STREAMS from the
   start of each plane and irow / orow are ignored, except irow[0] for the planar / packed 4:4:4 sources whose functions take one.  Their quirks
   are kept (DESIGN.md K5c-*): 4:4:4 planar -> 4:1:1 leaves only the frame's last macropixel, in the first slot; packed 4:4:4 -> 4:1:1 converts the rows
   that begin within the first width * height bytes; YUV411 -> 4:2:0 leaves all chroma in chroma row 0. */
int lgpu_yuv_repack(int in_pal, int out_pal, const uint8_t *const src_d[4], const int irow[4], uint8_t *const dst_d[4],
                    const int orow[4], int width, int height, int clamping_unclamped, int sampling, void *stream);
/* "softlight": lives-plugins/weed-plugins/softlight.c:62-141.  Planar YUV (palette 544 YUV444P, 545 YUVA4444P,
   522 YUV422P, 512 YUV420P, 513 YVU420P): gradient-magnitude highlight mixed into plane 0 (frame border copied), the
   other planes are copied.  unclamped != 0: luma range 0..255, else 16..235 (the channel's YUV_clamping leaf).
   src_d / dst_d / irow / orow: one entry per plane (3, or 4 for YUVA4444P).  Not in place (the filter is not
   CAN_DO_INPLACE). */
int lgpu_softlight(const uint8_t *const src_d[4], const int irow[4], uint8_t *const dst_d[4], const int orow[4],
                   int width, int height, int palette, int unclamped, void *stream);
/* "edge detect": lives-plugins/weed-plugins/edge.c:129-248.  palette 1..5 (RGB24, BGR24, RGBA32, BGRA32, ARGB32);
   mode 0 normal (edges keep the source colour) / 1 monochrome (white) / 2 supercolour (luma pass + one pass per colour
   byte).  Gradient magnitude of the 3x3-summed central differences, global Otsu threshold over a 1017-bin histogram
   (device-side, no host round trip), non-edges black.  dst_d may equal src_d (CAN_DO_INPLACE). */
int lgpu_edge(const uint8_t *src_d, int irow, uint8_t *dst_d, int orow, int width, int height, int palette, int mode,
              void *stream);
/* "blurzoom" (RadioacTV): lives-plugins/weed-plugins/blurzoom.c.  Stateful across frames (background luma, feedback
   plane, snapshot frame and countdown) -> an opaque handle per filter instance, created for one frame geometry
   (the reference's channel template is REINIT_ON_SIZE_CHANGE).  palette 3 RGBA32 / 4 BGRA32; 32 <= width < 8192.
   mode 0 normal / 1 strobe / 2 strobe2 / 3 trigger, pattern 0 blue / 1 green / 2 red / 3 white.  Per frame: background
   subtract + threshold, OR into the feedback plane, 4-neighbour blur (:149-168), zoom (the reference's serial pointer
   walk :171-192 as a gather over prefix-summed step tables), saturating palette add.  Modes 1 / 2 need compact source rows
   (the reference walks its snapshot with the source's row padding, :391-396).  Calls on one handle must be stream-ordered. */
typedef struct lgpu_blurzoom lgpu_blurzoom;
int lgpu_blurzoom_create(int width, int height, int palette, lgpu_blurzoom **out);
int lgpu_blurzoom_process(lgpu_blurzoom *bz, const uint8_t *src_d, int irow, uint8_t *dst_d, int orow, int mode, int pattern,
                          void *stream);
void lgpu_blurzoom_destroy(lgpu_blurzoom *bz);
/* geometric transitions: lives-plugins/weed-plugins/multi_transitions.c:86-233.  type 0 "iris rectangle", 1 "iris circle",
   2 "4 way split"; amount = the transition parameter 0..1; packed pixels of 3 or 4 bytes.  dst_d may equal src1_d for
   types 0 / 1 (the reference's out channel is CAN_DO_INPLACE there), not for type 2.  ("dissolve": lgpu_dissolve below; "rand replace"
   draws from a time-seeded global generator and stays on the CPU.) */
int lgpu_transition(int type, const uint8_t *src1_d, int irow1, const uint8_t *src2_d, int irow2, uint8_t *dst_d, int orow,
                    int width, int height, int psize, double amount, void *stream);
/* negate / posterise / ccorrect (lives-plugins/weed-plugins/scripts/{negate,posterise,ccorrect}.script): all three map every byte of a pixel
   through a table that depends on its position in the pixel only.  lgpu_fx_luts builds the tables on the host (kind 0 negate, 1 posterise
   with p0 = levels, 2 ccorrect with p0..p2 = red / green / blue factors; palette = WEED_PALETTE_* 1..5; returns psize, 0 = combination the
   reference does not offer), lgpu_byte_luts applies luts[psize][256] (host pointer, travels as a kernel argument).  src_d == dst_d allowed. */
int lgpu_fx_luts(int kind, int palette, double p0, double p1, double p2, uint8_t *luts_out);
int lgpu_byte_luts(const uint8_t *src_d, int irow, uint8_t *dst_d, int orow, int width, int height, int psize, const uint8_t *luts, void *stream);
/* "RGBdelay" / "YUVdelay": lives-plugins/weed-plugins/RGBdelay.c:135-416, stateful: a ring of up to 50 frames stays in HBM inside the
   handle.  palette 1 RGB24 / 2 BGR24 / 588 YUV888 (yuv_clamped = the channel's YUV_clamping leaf is CLAMPED); maxcache = parameter 0;
   on[3 * j + c] = the R / G / B (Y / U / V) switches and strength[j] the blend strength of frame -j, j = 0..50 (parameters 4j + 1 .. 4j + 4).
   src_d == dst_d = in place.  The host-ease branch (:180-183, :407-412) is not taken.  Calls on one handle must be stream-ordered;
   a change of geometry restarts the ring (the reference's in channel is REINIT_ON_SIZE_CHANGE). */
typedef struct lgpu_rgbdelay lgpu_rgbdelay;
int lgpu_rgbdelay_create(lgpu_rgbdelay **out);
int lgpu_rgbdelay_process(lgpu_rgbdelay *rd, const uint8_t *src_d, int irow, uint8_t *dst_d, int orow, int width, int height, int palette,
                          int yuv_clamped, int maxcache, const int *on, const double *strength, void *stream);
void lgpu_rgbdelay_destroy(lgpu_rgbdelay *rd);
/* "deinterlace": lives-plugins/weed-plugins/deinterlace.c:45-308.  Packed palettes (WEED_PALETTE_* 1..5, 588, 589, 564, 565;
   width in macropixels for UYVY / YUYV); src_d == dst_d = in place (the reference's out channel is CAN_DO_INPLACE).  Rows
   r - 1 and r are written for every odd r < height - 2, everything else is left alone (alpha of 4-byte pixels: row r only,
   out of place).  LGPU_E_UNSUPPORTED: planar palettes, ARGB32 out of place, a width that is not a multiple of 3 with rows
   too tight for the last partial triple. */
int lgpu_deinterlace(const uint8_t *src_d, int irow, uint8_t *dst_d, int orow, int width, int height, int palette, void *stream);
/* "triple split": lives-plugins/weed-plugins/layout_blends.c:24-113, RGB24 / BGR24.  start / end / border_width are the float parameters,
   symmetrical = "sym", split_rows = "vert", border_rgb[3] = "borderc".  src1_d == dst_d = in place (the middle band is then left alone). */
int lgpu_triple_split(const uint8_t *src1_d, int irow1, const uint8_t *src2_d, int irow2, uint8_t *dst_d, int orow, int width, int height,
                      int is_bgr, double start, int symmetrical, double end, int split_rows, double border_width, const int *border_rgb,
                      void *stream);
/* "slide over": lives-plugins/weed-plugins/slide_over.c:54-146.  amount = the transition parameter 0..255; direction 1..4 as
   sover_init stores it in "plugin_direction" (:40-51; 0 = random is drawn by the caller, :83-86); slide_lower / slide_upper =
   the "mlower" / "mupper" switches.  Packed pixels of 3 or 4 bytes, not in place. */
int lgpu_slide_over(const uint8_t *src1_d, int irow1, const uint8_t *src2_d, int irow2, uint8_t *dst_d, int orow,
                    int width, int height, int psize, int amount, int direction, int slide_lower, int slide_upper, void *stream);
/* "dissolve": multi_transitions.c:41-69 (mask) and :208-212 (select).  lgpu_dissolve_mask fills width * height floats on the host from the
   filter instance's "random_seed" (xorshift64 chain, libweed/weed-plugin-utils.c:666-704; once per instance); lgpu_dissolve shows src2 where
   mask < (float)amount.  mask_d: the same floats in device memory.  src1_d == dst_d = in place. */
int lgpu_dissolve_mask(uint64_t seed, int width, int height, float *mask_out);
int lgpu_dissolve(const uint8_t *src1_d, int irow1, const uint8_t *src2_d, int irow2, uint8_t *dst_d, int orow, int width, int height,
                  int psize, const float *mask_d, double amount, void *stream);
/* BASELINE config 4 in one launch: lgpu_gauss5 of frame 0, then lgpu_colorkey of the blurred frame against frame 1 -- the blurred frame never exists in memory.
   psize 3 (RGB24 / BGR24, the reference's palettes) or 4 (RGBA32 / BGRA32: the extension SURVEY 8d names for the headline size; the alpha byte stays the blurred
   frame's).  LGPU_E_UNSUPPORTED when width % 4 != 0 or a frame / rowstride is not 4- (psize 3) or 16-byte (psize 4) aligned: run the two entry points then. */
int lgpu_gauss5_colorkey(const uint8_t *src0_d, int irow0, const uint8_t *src1_d, int irow1, uint8_t *dst_d, int orow, int width, int height, int psize,
                         int is_bgr, double delta, double opac, int col_r, int col_g, int col_b, void *stream);
/* mirrorx (0) / mirrory (1) / mirrorxy (2): lives-plugins/weed-plugins/mirrors.c:26-122.  src_d may equal dst_d. */
int lgpu_mirror(int mode, const uint8_t *src_d, int irow, uint8_t *dst_d, int orow, int width, int height,
                int psize, void *stream);

/* ---- the fused per-track chain (BASELINE config 5 / north_star headline) ------------------------------
   convert (BGRA32 -> RGBA32 when swap_rb) -> resize -> [gauss5] -> chroma blend(bf) with layer2 -> gamma LUT,
   one launch for a batch of independent tracks.  Bit-identical to running the single entry points in
   that order (tests/test_chain_gpu.py).  All tracks share geometry and parameters (the shared
   transition parameter block of SURVEY 8e); per-track pointers travel as kernel arguments. */
#define LGPU_CHAIN_MAX_TRACKS 64
#define LGPU_INTERP_PIXBUF 0x100
#define LGPU_INTERP_OPAQUE 0x200   /* with LGPU_INTERP_PIXBUF: the caller STATES that every source pixel has alpha 255 (decoded video, a frame that has just been given
                                      its alpha channel).  The alpha weighting then cancels exactly (checked on the device for every sum the weights can make) and
                                      lighter instantiations with the same bytes run: the gaussian chain (do_blur, exact 2:1; lgpu_chain) and the two-dimensional
                                      filters of every other ratio (lgpu_pixbuf_scale[_batch], channels 4: pass interp | LGPU_INTERP_OPAQUE); ignored elsewhere.
                                      A frame that is not opaque gets wrong colours: state it only when known. */
#define LGPU_INTERP_NOBLEND 0x400  /* lgpu_chain_amounts only: the track has no layer 2 -- [R <-> B] -> scale [-> letterbox] -> gamma LUT, the plan steps of a track that is not
                                      blended with anything (layer2_d, irow2 and amounts are not read).  One launch like the blended form. */
typedef struct {
  const uint8_t *src_d;      /* sw x sh, 4 bytes / pixel */
  const uint8_t *layer2_d;   /* dw x dh, RGBA32 */
  uint8_t *dst_d;            /* dw x dh, RGBA32 */
} lgpu_chain_track;
typedef struct {
  int sw, sh, irow;          /* source geometry */
  int dw, dh, irow2, orow;   /* output / layer-2 geometry */
  int swap_rb;               /* 1: source is BGRA32 (K1 swap3postalpha fused into the load) */
  int interp;                /* LiVESInterpType; | LGPU_INTERP_PIXBUF: the resize stage follows the reference's gdk-pixbuf body (lgpu_pixbuf_scale, 4 channels with
                                alpha) instead of the polyphase spec -- pinned arithmetic; the exact aligned 2:1 case is one fused launch */
  int do_blur;               /* 1: 5x5 gaussian between resize and blend */
  int bf;                    /* chroma blend amount 0..255 */
  int use_lut;               /* 1: apply lut8 after the blend */
  uint8_t lut8[256];
  const int32_t *param_block_d; /* optional DEVICE pointer to the shared transition parameter block
                                   (int32[0] = blend amount); overrides `bf` when non-NULL.  This is the
                                   block rank 0 broadcasts over RCCL/xGMI in the multi-GPU batch:
                                   it is read by the kernel, stream-ordered, with no host round trip. */
} lgpu_chain_params;
int lgpu_chain(const lgpu_chain_params *params, const lgpu_chain_track *tracks, int ntracks, void *stream);
/* the same chain with letterbox_layer (src/colourspace.c:15343-15567) between the resize and the blend -- BASELINE config 3: the scaled dw x dh frame sits at
   (offs_x, offs_y) of an nwidth x nheight canvas of opaque black (LiVES centres it: offs = (n - size + 1) >> 1, :15522-15523); layer 2 and the destination are
   canvas-sized (irow2 / orow are their strides).  One launch on the pixbuf arithmetic for the exact aligned 2:1 case with an even offs_x; staged otherwise. */
typedef struct { int nwidth, nheight, offs_x, offs_y; } lgpu_canvas;
int lgpu_chain_canvas(const lgpu_chain_params *params, const lgpu_canvas *canvas, const lgpu_chain_track *tracks, int ntracks, void *stream);
/* lgpu_chain / lgpu_chain_canvas (canvas may be NULL) on the gdk-pixbuf arithmetic with a blend amount PER TRACK (amounts[ntracks], 0..255; params->bf and
   params->param_block_d are not used): the tracks of a tick share geometry and gamma table, not necessarily their transition amount -- still one launch.  What
   lives_gpu_layers_flush() emits for the recorded plan steps of a tick (lives_gpu_layer.h).  Two more shapes are served here only: interp | LGPU_INTERP_NOBLEND (no
   layer 2; amounts may be NULL), and sw == dw && sh == dh -- no resize stage: [R <-> B] [-> letterbox] [-> blend] [-> gamma LUT] on frames that already have their
   size (dst must not be src). */
int lgpu_chain_amounts(const lgpu_chain_params *params, const lgpu_canvas *canvas, const lgpu_chain_track *tracks, int ntracks, const uint8_t *amounts, void *stream);

/* ---- timing helper: HIP events on `stream` around `reps` launches of the last-configured chain; used by
   bench.py to measure the kernel's average launch duration on the stream it is launched on. */
int lgpu_chain_timed(const lgpu_chain_params *params, const lgpu_chain_track *tracks, int ntracks,
                     int reps, float *ms_total, void *stream);

/* measurement hook (bench.py: roofline.box_class): the chain's own algorithmic bytes as a bare stream on the same frames -- every source and layer-2 byte read once
   (16-byte non-temporal loads), every destination byte written once (non-temporal stores), no arithmetic, no re-reads; HIP events on `stream` around `reps` launches.
   Compact frames only; the destination frames are left dirty. */
int lgpu_debug_stream_probe(const lgpu_chain_params *params, const lgpu_chain_track *tracks, int ntracks, int reps, float *ms_total, void *stream);

/* ---- multi-GPU exchange: one process per GPU, tracks sharded track t -> rank t % world, no data-path collective.  What the
   ranks exchange goes over RCCL (xGMI), bound at run time (dlopen of librccl.so.1: no link-time dependency, a single-GPU host never loads
   it).  The communicator is RCCL's own: rank 0 makes the id, the host ships its 128 bytes to the other ranks by whatever channel it has,
   every rank calls lgpu_dist_comm_create.  All calls below are stream ordered and never synchronise the host. */
#define LGPU_DIST_ID_BYTES 128
int lgpu_dist_bind(const char *rccl_path);                                  /* optional: an explicit library path (NULL: the process's copy, then the system's) */
int lgpu_dist_unique_id(uint8_t id[LGPU_DIST_ID_BYTES]);                    /* ncclGetUniqueId */
int lgpu_dist_comm_create(const uint8_t id[LGPU_DIST_ID_BYTES], int rank, int world, void **comm);   /* ncclCommInitRank on the current device */
/* the same with a limit: ncclCommInitRank blocks for ever when a rank of the job never arrives; here the call returns LGPU_E_TIMEOUT after timeout_ms
   (lgpu_last_error(): which rank of how many waited how long) and the process can report and exit -- the abandoned initialisation is left behind, do not reuse the id.
   timeout_ms <= 0: no limit */
int lgpu_dist_comm_create_timeout(const uint8_t id[LGPU_DIST_ID_BYTES], int rank, int world, int timeout_ms, void **comm);
int lgpu_dist_comm_count(void *comm);                                       /* ncclCommCount: the ranks of the communicator, or a negative LGPU_E_* */
int lgpu_dist_comm_destroy(void *comm);
/* the shared transition parameter block (lgpu_chain_params.param_block_d: int32[4] in device memory) from `root` to every rank, in place */
int lgpu_params_broadcast(void *comm, int root, int32_t *param_block_d, void *stream);
/* max of a device status word over the ranks: the optional completion / error barrier */
int lgpu_status_allreduce(void *comm, int32_t *status_d, void *stream);
/* compositing fan-in: each rank's processed frames (nlocal = its share of ntracks, frame_bytes each, contiguous) to `root`, which gets the
   ntracks frames in track order in gathered_d; point-to-point sends inside one RCCL group */
int lgpu_fan_in(void *comm, int root, int rank, int world, int ntracks, const uint8_t *frames_d, size_t frame_bytes, uint8_t *gathered_d, void *stream);

/* the control rank's write of the block (four host values as kernel arguments: one tiny launch, no host -> device copy) */
int lgpu_params_set(int32_t *param_block_d, const int32_t values[4], void *stream);
/* the same for nblocks consecutive blocks: one launch / one ncclBroadcast of nblocks x 16 bytes */
int lgpu_params_set_n(int32_t *param_blocks_d, const int32_t *values, int nblocks, void *stream);
int lgpu_params_broadcast_n(void *comm, int root, int32_t *param_blocks_d, int nblocks, void *stream);
/* ---- one step of the batch from C (north_star "host code stays C"): the chain over this rank's tracks with the step's parameter block; the blocks
   travel AHEAD of the kernels on a side stream into a ring of 64 device blocks (the launch stream never waits for xGMI, the host never synchronises).
   lgpu_stepper_feed hands over the blocks of the next n steps in ONE exchange (root: one tiny launch + one ncclBroadcast of n x 16 bytes; every rank calls it with
   the same n; values = n x 4 ints, read on the root only): a render knows its schedule ahead, and a live parameter is one feed of latency either way -- per step
   the host then pays the chain launch and 1 / n of an exchange.  lgpu_chain_step with next_values != NULL is the one-block-ahead form of the same thing.
   comm == NULL: one GPU, nothing to exchange -- the blocks are written on the launch stream.  tools/worker.c is the render-worker loop on top of it.
   Errors: LGPU_E_BADARG comes from argument checks only -- EVERY check lgpu_chain makes on the same arguments is made before anything is fed or enqueued -- and
   leaves the stepper untouched (repeat the call); any other error leaves this rank out of step with its peers: lgpu_stepper_failed() turns 1, every later call
   returns LGPU_E_STATE, destroy the stepper.
   lgpu_stepper_wait(s, timeout_ms): the host-side wait a worker uses INSTEAD of a bare stream synchronise -- until everything fed and launched so far has
   completed, or LGPU_E_TIMEOUT after timeout_ms (0 = for ever) with lgpu_last_error() naming what hangs (the exchange on the side stream = a peer that never
   entered the same feed, or a launch) and the step / feed counters of this rank; a timed-out stepper is failed like any other. */
typedef struct lgpu_stepper lgpu_stepper;
int lgpu_stepper_create(void *comm, int root, int rank, void *launch_stream, const int32_t first_values[4], lgpu_stepper **out);
int lgpu_stepper_feed(lgpu_stepper *s, const int32_t *values, int n);
/* next_values: the block of the following step (read on the root only) when no feed has brought it yet; NULL on every rank after the last step or when
   lgpu_stepper_feed is used.  params->param_block_d is replaced by the stepper's block. */
int lgpu_chain_step(lgpu_stepper *s, const int32_t next_values[4], const lgpu_chain_params *params, const lgpu_chain_track *tracks, int ntracks);
/* a second launch stream: odd steps go there, so that the drain of one frame's launch overlaps the ramp-up of the next (frames of consecutive steps are independent).
   NULL switches it off.  The caller synchronises both streams before reading results. */
int lgpu_stepper_overlap(lgpu_stepper *s, void *second_launch_stream);
int lgpu_chain_check(const lgpu_chain_params *params, const lgpu_chain_track *tracks, int ntracks);     /* lgpu_chain's argument checks alone (no device work): LGPU_OK or LGPU_E_BADARG */
int lgpu_stepper_wait(lgpu_stepper *s, int timeout_ms);
int lgpu_stepper_failed(const lgpu_stepper *s);                           /* 1 after a half-done call, else 0 */
const int32_t *lgpu_stepper_block(const lgpu_stepper *s, int which);      /* ring slot which % 64 (tests); NULL for a negative index */
int lgpu_stepper_destroy(lgpu_stepper *s);

/* ---- batch forms of the CONVERT-step kernels and of the single-plane effects: nframes (1..LGPU_FX_MAX_FRAMES) frames of ONE geometry -- the same substep of the
   live tracks of a tick (src/nodemodel.c:1093 pconv, :1138 gamma, :1253 letterbox run once per track per tick, exactly like the effects) -- as ONE launch: the frame
   is the grid's z index, the frame pointers travel in the kernarg segment.  Arguments are those of the single-frame entry point with frame tables in place of the
   frame pointers; the results are those of nframes single calls, bit for bit (tests/test_batch_forms.py).  The vector forms are taken when EVERY frame is aligned.
   (K2 has lgpu_yuv420p_to_rgb_batch, the scaler lgpu_pixbuf_scale_batch, the two-input effects lgpu_fx_batch, the chain its tracks.) */
int lgpu_swizzle_batch(int op, int alpha_first, const uint8_t *const *src_d, int irow, uint8_t *const *dst_d, int orow, int width, int height,
                       const uint8_t *lut8, int nframes, void *stream);
int lgpu_gamma_apply_batch(uint8_t *const *pix_d, int rowstride, int x, int y, int width, int height, int psize, int alpha_first, const uint8_t *lut8,
                           int nframes, void *stream);
int lgpu_alpha_premult_batch(uint8_t *const *pix_d, int rowstride, int width, int height, int alpha_first, int un, int nframes, void *stream);
int lgpu_mirror_batch(int mode, const uint8_t *const *src_d, int irow, uint8_t *const *dst_d, int orow, int width, int height, int psize, int nframes, void *stream);
int lgpu_letterbox_batch(const uint8_t *const *src_d, int irow, int width, int height, uint8_t *const *dst_d, int orow, int nwidth, int nheight, int psize,
                         const uint8_t black_pixel[4], int nframes, void *stream);          /* centred, as lgpu_letterbox */
int lgpu_colorkey_batch(const uint8_t *const *src0_d, int irow0, const uint8_t *const *src1_d, int irow1, uint8_t *const *dst_d, int orow, int width, int height,
                        int is_bgr, double delta, double opac, int col_r, int col_g, int col_b, int nframes, void *stream);
/* K4 / K3 (lgpu_rgb_to_yuv / lgpu_yuv_to_rgb): the planar side's table holds nframes x 4 plane pointers, [frame * 4 + plane] */
int lgpu_rgb_to_yuv_batch(const uint8_t *const *src_d, int irow, int width, int height, int in_order, int in_alpha, uint8_t *const *dst_d, const int orow[4],
                          int out_fmt, int out_alpha, int which_tables, int nframes, void *stream);
int lgpu_yuv_to_rgb_batch(const uint8_t *const *src_d, const int irow[4], int width, int height, int in_fmt, int in_alpha, uint8_t *const *dst_d, int orow,
                          int out_order, int out_alpha, int which_tables, int nframes, void *stream);

/* ---- batched effects: ONE launch for the instances of one filter on the live tracks of a tick (weed_apply_instance runs once per track per tick,
   src/effects-weed.c:1850-2425).  The frames share geometry, rowstrides and parameters; only the planes differ.  Results are those of nframes single calls, bit
   for bit.  ops and their fields (everything else ignored):
     LGPU_FX_SOFTLIGHT      in0 / out: the 3 (4: YUVA4444P) planes; irow0[], orow[], width, height, palette; ip[0] = unclamped          (lgpu_softlight)
     LGPU_FX_TRANSITION     in0[0], in1[0], out[0]; irow0[0], irow1[0], orow[0], width, height; ip[0] = type, ip[1] = psize, dp[0] = amount, or frame_dp0[f] = the amount of frame f   (lgpu_transition)
     LGPU_FX_YUV411_TO_RGB  in0[0], out[0]; width = macropixels, height, orow[0]; ip[0] = out_order, ip[1] = out_alpha, ip[2] = unclamped      (lgpu_yuv411_to_rgb)
     LGPU_FX_GAUSS5_COLORKEY in0[0], in1[0], out[0]; irow0[0], irow1[0], orow[0], width, height; ip[0] = psize, ip[1] = is_bgr, ip[2] = key colour r | g << 8 | b << 16,
                            dp[0] = delta, dp[1] = opacity  (lgpu_gauss5_colorkey: BASELINE config 4 in one launch; LGPU_E_UNSUPPORTED outside its alignment range)
     LGPU_FX_BLEND_CHROMA   in0[0], in1[0], out[0] (out may be in0: in place); irow0[0], irow1[0], orow[0], width, height; ip[0] = psize, ip[1] = 0 (ARGB32: lgpu_blend_chroma),
                            dp[0] = blend amount, or frame_dp0[f]   (lgpu_blend_chroma)
     LGPU_FX_BLEND_LUMA     as above; ip[0] = type 1..4, ip[1] = psize, ip[2] = pal_order; dp[0] / frame_dp0[f] = threshold   (lgpu_blend_luma)
     LGPU_FX_BLEND_MULTI    as above, 3-byte pixels; ip[0] = type 0..6, ip[1] = is_bgr; dp[0] / frame_dp0[f] = blend amount   (lgpu_blend_multi)
   The scaler has its own batch entry (lgpu_pixbuf_scale_batch), the palette conversion K2 lgpu_yuv420p_to_rgb_batch, the chain takes its tracks directly.  The
   compositor (lgpu_composite) IS the fan-in of a tick's tracks into one frame: a tick has one of it, there is nothing to batch. */
#define LGPU_FX_MAX_FRAMES 16
enum { LGPU_FX_SOFTLIGHT = 1, LGPU_FX_TRANSITION = 2, LGPU_FX_YUV411_TO_RGB = 3, LGPU_FX_GAUSS5_COLORKEY = 4, LGPU_FX_BLEND_CHROMA = 5, LGPU_FX_BLEND_LUMA = 6, LGPU_FX_BLEND_MULTI = 7 };
typedef struct { const uint8_t *in0[4]; const uint8_t *in1[4]; uint8_t *out[4]; } lgpu_fx_frame;
typedef struct { int op, width, height, palette; int irow0[4], irow1[4], orow[4]; int ip[4]; double dp[2];
                 const double *frame_dp0; /* NULL, or nframes values that replace dp[0] frame by frame (the transitions and the blends; BADARG for the other ops) */ } lgpu_fx_params;
int lgpu_fx_batch(const lgpu_fx_params *params, const lgpu_fx_frame *frames, int nframes, void *stream);

/* ---- compositor fan-in (SURVEY 8f "next" 1): lives-plugins/weed-plugins/gdk/compositor.c:120-125 (paint_pixel),
   :167-189 (background, z order), :288-293 (paint loop).  One kernel: every output pixel starts from bgcol (R,G,B;
   alpha byte 0xFF) and takes the layers that cover it in paint order -- revz == 0: the last layer first, so layer 0 ends
   on top -- with dst.c = (uint8_t)(dst.c * (1. - alpha) + src.c * alpha) in double per colour byte.  Layers arrive
   already scaled -- by lgpu_pixbuf_scale, bit-exact to the gdk_pixbuf_scale_simple call of compositor.c:262-266 (GDK_INTERP_HYPER when either side grows,
   GDK_INTERP_BILINEAR otherwise; 4-byte palettes with alpha) -- at pixel offsets
   offs = (int)(offs_fraction * out_size).  Up to LGPU_COMP_MAX_LAYERS layers per call. */
#define LGPU_COMP_MAX_LAYERS 16
typedef struct {
  const uint8_t *src_d;      /* NULL: layer disabled (compositor.c:192-195) */
  int irow, width, height;
  int offs_x, offs_y;        /* position of the layer's top-left pixel in the output */
  double alpha;              /* per-layer opacity 0..1 */
} lgpu_comp_layer;
int lgpu_composite(uint8_t *dst_d, int orow, int owidth, int oheight, int psize, int is_bgr, const int bgcol[3],
                   const lgpu_comp_layer *layers, int nlayers, int revz, void *stream);

#ifdef __cplusplus
}
#endif
#endif
