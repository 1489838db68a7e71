/* lives_oracle.h -- CPU restatement of the LiVES per-frame hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (lives_amd/, include/) may include, link or
 * call this.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it -- as the
 * checker / the CPU number reported beside the GPU one, never as the thing shipped.
 *
 * Parity pin status (see DESIGN.md "Oracle"):
 *   pinned   : tables, gamma LUT8, al/unal, K1 swizzles, YUV420P->RGB, gamma apply  -> against
 *              oracle/_ref/libcsref.so (line-range slices of the reference's src/colourspace.c) and the
 *              committed fixtures under tests/golden/;
 *              chroma/luma blends, multi blends, mirrors, colour key -> against the reference's own
 *              plugins built unmodified into oracle/_ref/ (.so files) and the committed fixtures.
 *   UNPINNED : orc_resize (the reference calls FFmpeg libswscale, un-vendored, version unpinned --
 *              src/colourspace.c:14711) and orc_gauss5 (no reference loop exists): both follow the
 *              specification written in DESIGN.md.  "parity unpinned" for these two.
 */
#ifndef LIVES_ORACLE_H
#define LIVES_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* which: bit0 = unclamped, bit1 = BT.709.  rgb2yuv[9][256] = Y_R Y_G Y_B Cb_R Cb_G Cb_B Cr_R Cr_G Cr_B,
   yuv2rgb[5][256] = RGB_Y R_Cr G_Cb G_Cr B_Cb   (src/colourspace.c:851-1105) */
void orc_tables(int which, int32_t *rgb2yuv, int32_t *yuv2rgb);
/* create_gamma_lut8 (src/colourspace.c:655-736), including its deterministic quirks */
int orc_gamma_lut8(double fileg, int gamma_from, int gamma_to, double screen_gamma, uint8_t *lut);
/* init_unal (src/colourspace.c:1141-1160): value of unal[alpha][v] / al[alpha][v] */
int orc_unal(int alpha, int v);
int orc_al(int alpha, int v);

/* K1: the 13 packed RGB swizzles (src/colourspace.c:9259-10577).  op ids = order of definition. */
enum { ORC_SWAP3, ORC_SWAP4, ORC_SWAP3ADDPOST, ORC_SWAP3ADDPRE, ORC_SWAP3POSTALPHA, ORC_SWAP3PREALPHA,
       ORC_ADDPOST, ORC_ADDPRE, ORC_SWAP3DELPOST, ORC_DELPOST, ORC_DELPRE, ORC_SWAP3DELPRE, ORC_SWAPPREPOST };
int orc_swizzle(int op, int alpha_first, const uint8_t *src, int irow, uint8_t *dst, int orow,
                int width, int height, const uint8_t *lut8);

/* K2: convert_yuv420p_to_rgb_frame (src/colourspace.c:3260-3904), 1-thread semantics.
   out_order: 0 = RGB(A), 1 = BGR(A), 2 = ARGB.  u_size/v_size = readable bytes of the chroma planes.
   fix_edges: 0 = replicate every deterministic reference behaviour (garbage last row included),
              1 = compute the evident intent on row 0 (odd x) and on the last row. */
int orc_yuv420p_to_rgb(const uint8_t *y, const uint8_t *u, const uint8_t *v, const int istrides[3],
                       long u_size, long v_size, uint8_t *dst, int orow, int width, int height,
                       int opsize, int out_order, int is_422, int which_tables, int pb_quality,
                       const uint8_t *lut8, int fix_edges);

/* the same with the 16-bit indexed gamma LUT of create_gamma_lut (:738-808) fused in, as the reference does when it is
   handed a target gamma (:3274-3283, xyuv2rgb_with_gamma :2386-2390): c = lut16[CLAMP16biti(sum >> 8)] >> 8 */
int orc_yuv420p_to_rgb_lut16(const uint8_t *y, const uint8_t *u, const uint8_t *v, const int istrides[3],
                             long u_size, long v_size, uint8_t *dst, int orow, int width, int height,
                             int opsize, int out_order, int is_422, int which_tables, int pb_quality,
                             const uint16_t *lut16, int fix_edges);
int orc_gamma_lut16(double fileg, int gamma_from, int gamma_to, double screen_gamma, uint16_t *lut);

/* K6: gamma_convert_layer_thread (src/colourspace.c:14034-14060) */
void orc_gamma_apply(uint8_t *pix, int rowstride, int width, int height, int psize, int alpha_first,
                     const uint8_t *lut8);
/* K9: alpha_premult packed RGBA/BGRA (aoffs 3, coffs 0) / ARGB (aoffs 0, coffs 1); un = 1: unal (REVERSE) */
void orc_alpha_premult(uint8_t *pix, int rowstride, int width, int height, int alpha_first, int un);
/* K9b: the clamped-YUV premultiply tables of init_unal (:1141-1160) and alpha_premult for YUVA8888 (589) / YUVA4444P (545), :11995-12096 */
void orc_premult_yuv_tables(uint8_t *unalcy, uint8_t *alcy, uint8_t *unalcuv, uint8_t *alcuv);
int orc_alpha_premult_yuva(uint8_t *const planes[4], const int rows[4], int width, int height, int palette, int clamped, int un);
/* K8: letterbox blit into an opaque-black canvas (src/colourspace.c:15343-15567, fill :11109-11119) */
void orc_letterbox(const uint8_t *src, int irow, int width, int height, uint8_t *dst, int orow,
                   int nwidth, int nheight, int psize, const uint8_t *black_pixel);

/* F1: "chroma blend" simple_blend.c:117-150 (psize 3 or 4; alpha_first => ARGB quirk path) */
void orc_blend_chroma(const uint8_t *src1, int irow1, const uint8_t *src2, int irow2, uint8_t *dst, int orow,
                      int width, int height, int psize, int alpha_first, int bf);
/* F2: luma overlay (1) / underlay (2) / negative (3) / averaged (4 == 1 in the reference) simple_blend.c:151-194
   pal_order: 0 = RGB.., 1 = BGR.., 2 = ARGB */
void orc_blend_luma(int type, const uint8_t *src1, int irow1, const uint8_t *src2, int irow2, uint8_t *dst,
                    int orow, int width, int height, int psize, int pal_order, int thresh, int inplace);
/* F3: multi_blends.c:68-162 (RGB24 / BGR24), type 0..6 */
void orc_blend_multi(int type, const uint8_t *src1, int irow1, const uint8_t *src2, int irow2, uint8_t *dst,
                     int orow, int width, int height, int is_bgr, int bf);
/* F4: colour key, scripts/colorkey.script <process> */
void orc_colorkey(const uint8_t *src0, int irow0, const uint8_t *src1, int irow1, uint8_t *dst, int orow,
                  int width, int height, int is_bgr, double delta, double opac, int col_r, int col_g, int col_b,
                  int inplace);
/* extension: the key on 4-byte pixels, alpha from the first frame (own spec; BASELINE config 4's RGBA32 form) */
void orc_colorkey4(const uint8_t *src0, int irow0, const uint8_t *src1, int irow1, uint8_t *dst, int orow,
                   int width, int height, int is_bgr, double delta, double opac, int col_r, int col_g, int col_b);
/* F5: mirrors.c:26-122.  mode 0 = x, 1 = y, 2 = xy.  (OOB writes of the reference are not performed.) */
void orc_mirror(int mode, const uint8_t *src, int irow, uint8_t *dst, int orow, int width, int height, int psize);

/* K4: packed RGB family -> YUV family (src/colourspace.c:5129-6440, pixel maths :2119-2192, 1-thread semantics).
   in_order 0 RGB, 1 BGR, 2 ARGB; in_alpha: 4-byte source pixels (always for ARGB).
   out_fmt  0 packed YUV888 / YUVA8888 (out_alpha)      convert_{rgb,bgr,argb}_to_yuv_frame   :5700-6144
            1 planar YUV444P / YUVA4444P (out_alpha)    convert_{rgb,bgr,argb}_to_yuvp_frame  :5786-6239
            2 UYVY, 3 YUYV                              convert_{rgb,bgr,argb}_to_{uyvy,yuyv}_frame :5129-5690 (no gamma LUT)
            4 YUV420P, 5 YUV422P                        convert_{rgb,bgr}_to_yuv420_frame     :6250-6440
   which_tables: bit 0 unclamped, bit 1 BT.709 (4:2:0 / 4:2:2 only: the other entry points always use YCbCr).
   Packed/planar 4:4:4 leave the last pixel of an odd-width row unwritten
   (hsize = (hsize >> 1) << 1); the subsampled formats need an even width (4:2:0: and height).
   Reference-faithful oddities kept: UYVY/YUYV take U from the first and V from the SECOND pixel of a pair (no
   averaging); YUYV has no upper chroma clamp (:2183-2191); 4:2:0 chroma row k is avg_chroma(row 2k+2, row 2k+1) and the
   last chroma row is that of the last luma row alone (the in-place "average two rows" walk of :6302-6315 at compact
   strides).  Returns -1 for combinations the reference cannot do sanely (ARGB -> 4:2:0 reads the wrong bytes, :6353). */
int orc_rgb_to_yuv(const uint8_t *src, int irow, int width, int height, int in_order, int in_alpha,
                   uint8_t *const dst[4], const int orow[4], int out_fmt, int out_alpha, int which_tables);
/* K3: YUV family -> packed RGB family (src/colourspace.c:2750-3258, :6616-7102, :7200-7498; pixel maths :2345-2459).
   in_fmt 0 packed YUV888 / YUVA8888 (in_alpha), 1 planar YUV444P / YUVA4444P (in_alpha), 2 UYVY, 3 YUYV (width in
   pixels, even).  out_order 0 RGB, 1 BGR, 2 ARGB; out_alpha: 4-byte output (always for ARGB); alpha = source alpha
   or 255.  The reference's UYVY / YUYV / planar entry points always use the YCbCr tables (bit 1 of which_tables must
   be 0 for them).  Returns -1 where the reference's own row arithmetic is broken: planar -> ARGB (:7475-7476) and
   planar -> BGR24 (:7313 always steps 4 bytes). */
int orc_yuv_to_rgb(const uint8_t *const src[4], const int irow[4], int width, int height, int in_fmt, int in_alpha,
                   uint8_t *dst, int orow, int out_order, int out_alpha, int which_tables);
int orc_rgb_to_yuv_lut16(const uint8_t *src, int irow, int width, int height, int in_order, int in_alpha, uint8_t *dst, int orow, int out_fmt,
                         int unclamped, const uint16_t *lut16);
int orc_rgb_to_yuv411(const uint8_t *src, int irow, int width, int height, int in_order, int in_alpha, uint8_t *dst, int unclamped);
int orc_yuv411_to_rgb(const uint8_t *src, int width_mp, int height, uint8_t *dst, int orow, int out_order, int out_alpha, int unclamped);
/* init_average (src/colourspace.c:190-216): cavgc (clamped = 1) / cavgu chroma-averaging table entry */
int orc_cavg(int clamped, int x, int y);

/* K5: clamped <-> unclamped switch, in place (src/colourspace.c:10929-11090 switch_yuv_clamping_and_subspace, tables
   init_YUV_to_YUV_tables :1108-1139, selection :1163-1230: same table set for YCbCr and BT.709, no subspace maths).
   Every byte of height * rowstride is mapped (row padding included, as the reference walks the whole buffer); chroma
   planes of the subsampled planar palettes are walked for height * rowstride / 2 (4:2:2) or / 4 (4:2:0) bytes.
   palette: 588 YUV888, 589 YUVA8888 (alpha untouched), 544 / 545 planar 4:4:4(4), 522, 512, 513, 564 UYVY, 565 YUYV. */
void orc_yuv_yuv_tables(uint8_t *yc2u, uint8_t *uvc2u, uint8_t *yu2c, uint8_t *uvu2c);
int orc_switch_yuv_clamping(uint8_t *const planes[4], const int rowstrides[4], int palette, int height, int to_unclamped);
/* YUV -> YUV repacks (colourspace.c:7104-7198, :7500-7753, :7800-7971, :9198-9257, :10517-10575, :10612-10639 and the K1
   addpost / delpost pair); WEED_PALETTE_* numbers, width in pixels; -1 = pair / layout not taken */
int orc_yuv_repack(int in_pal, int out_pal, const uint8_t *const src[4], const int irow[4], uint8_t *const dst[4], const int orow[4],
                   int width, int height, int clamping_unclamped, int sampling);

/* F7: geometric transitions  lives-plugins/weed-plugins/multi_transitions.c:86-233: type 0 "iris rectangle", 1 "iris circle",
   2 "4 way split" (types 3 dissolve / 4 rand replace draw from the host's random generator and are not restated).
   Packed palettes, psize bytes per pixel; amount = the transition parameter 0..1; 1-thread semantics.  In place
   (dst == src1) is equivalent to out of place for types 0 / 1; type 2 is not in place (its out channel template says so). */
void orc_transition(int type, const uint8_t *src1, int irow1, const uint8_t *src2, int irow2, uint8_t *dst, int orow,
                    int width, int height, int psize, double amount);
/* dissolve (multi_transitions.c:41-69, :208-212): mask from the instance's random seed, then a per-pixel select */
void orc_dissolve_mask(uint64_t seed, int width, int height, float *mask);
void orc_dissolve(const uint8_t *src1, int irow1, const uint8_t *src2, int irow2, uint8_t *dst, int orow, int width, int height, int psize,
                  const float *mask, double amount);
/* slide over (slide_over.c:54-146): dirn 1..4 as stored by sover_init, transval 0..255 */
void orc_slide_over(const uint8_t *src1, int irow1, const uint8_t *src2, int irow2, uint8_t *dst, int orow, int width, int height,
                    int psize, int transval, int dirn, int mvlower, int mvupper);
/* triple split (layout_blends.c:24-113), RGB24 / BGR24; src1 == dst = in place */
void orc_triple_split(const uint8_t *src1, int irow1, const uint8_t *src2, int irow2, uint8_t *dst, int orow, int width, int height, int is_bgr,
                      double xstart, int sym, double xend, int vert, double bw, const int *bc);
/* deinterlace (deinterlace.c:45-308), packed palettes; src == dst = in place; -1 = not taken */
int orc_deinterlace(const uint8_t *src, int irow, uint8_t *dst, int orow, int width, int height, int palette);
/* negate / posterise / ccorrect (the three scripts of that name): per-byte-position tables + their application */
int orc_fx_luts(int kind, int palette, double p0, double p1, double p2, uint8_t *luts);
void orc_byte_luts(const uint8_t *src, int irow, uint8_t *dst, int orow, int width, int height, int psize, const uint8_t *luts);

/* F6a: "softlight"  lives-plugins/weed-plugins/softlight.c:62-141.  Planar YUV: the stencil runs on plane 0 (rows
   1..h-2, columns 1..w-2; the frame border is copied), the other planes are copied (:143-151).
   unclamped != 0: output range 0..255, else 16..235 (:97-103).  Needs width, height >= 3. */
void orc_softlight_y(const uint8_t *src, int irow, uint8_t *dst, int orow, int width, int height, int unclamped);
/* F6b: "edge detect"  lives-plugins/weed-plugins/edge.c:129-248.  mode 0 normal / 1 monochrome / 2 supercolour.
   pal: WEED palette id (1 RGB24, 2 BGR24, 3 RGBA32, 4 BGRA32, 5 ARGB32).  map16: caller's int16[width*height] scratch
   that must start zeroed (the reference's calloc'd sdata->map; its border cells are never written).  inplace: dst == src. */
void orc_edge(const uint8_t *src, int irow, uint8_t *dst, int orow, int width, int height, int pal, int mode,
              int16_t *map16, int inplace);

/* F6c: "blurzoom" (RadioacTV)  lives-plugins/weed-plugins/blurzoom.c:74-101 (background subtract), :106-145 (zoom tables),
   :149-168 (blur), :171-192 (zoom), :201-237 (palette), :345-421 (process).  Stateful across frames: background luma,
   feedback buffer, snapshot frame, snapshot countdown.  palette: WEED id 3 (RGBA32) or 4 (BGRA32).
   mode 0 normal / 1 strobe / 2 strobe2 / 3 trigger; pattern 0 blue / 1 green / 2 red / 3 white.  width >= 32.
   Modes 1 and 2 read the snapshot with the SOURCE row padding (:391-396): they need irow == 4 * width. */
typedef struct orc_blurzoom orc_blurzoom;
orc_blurzoom *orc_blurzoom_new(int width, int height, int palette);
int orc_blurzoom_process(orc_blurzoom *bz, const uint8_t *src, int irow, uint8_t *dst, int orow, int mode, int pattern);
void orc_blurzoom_free(orc_blurzoom *bz);
/* RGBdelay / YUVdelay (RGBdelay.c:36-431), stateful; palette 1 RGB24, 2 BGR24, 588 YUV888; on[3 * j + c] / strength[j] for the 51
   parameter groups; src == dst = in place */
typedef struct orc_rgbdelay orc_rgbdelay;
orc_rgbdelay *orc_rgbdelay_new(void);
int orc_rgbdelay_process(orc_rgbdelay *s, const uint8_t *src, int irow, uint8_t *dst, int orow, int width, int height, int palette,
                         int yuv_clamped, int maxcache, const int *on, const double *strength);
void orc_rgbdelay_free(orc_rgbdelay *s);

/* C1: "compositor" fan-in  lives-plugins/weed-plugins/gdk/compositor.c:120-125 (paint_pixel), :167-178 (background),
   :181-189 (z order), :288-293 (paint loop).  Layers arrive already scaled (the reference scales with gdk-pixbuf, which
   is un-vendored: scaling is the caller's lgpu-polyphase-v1 resize).  Per layer, in paint order (revz == 0: last
   layer first, so layer 0 ends on top): dst.c = (uint8_t)(dst.c * (1. - alpha) + src.c * alpha) in double for the
   three colour bytes; alpha byte of 4-byte palettes stays 0xFF.  bgcol is R,G,B; is_bgr swaps where R and B land. */
typedef struct { const uint8_t *src; int irow, width, height, offs_x, offs_y; double alpha; } orc_comp_layer;
void orc_composite(uint8_t *dst, int orow, int owidth, int oheight, int psize, int is_bgr, const int bgcol[3],
                   const orc_comp_layer *layers, int nlayers, int revz);

/* R1 (UNPINNED, spec "lgpu-polyphase-v1" in DESIGN.md) */
enum { ORC_INTERP_NEAREST = 0, ORC_INTERP_BILINEAR = 2, ORC_INTERP_HYPER = 3 };
int orc_make_filter(int srcn, int dstn, int kernel, int *ntaps, int32_t *pos, int16_t *coef, int maxtaps);
int orc_resize(const uint8_t *src, int irow, int sw, int sh, uint8_t *dst, int orow, int dw, int dh,
               int psize, int interp);
/* R1, pixbuf backend (PINNED on the gdk-pixbuf runtime library the reference's non-swscale resize body calls, src/colourspace.c:15295;
   oracle/orc_pixbuf.c).  channels 3 (no alpha) or 4 (alpha-weighted); interp as above.  0 ok, -1 bad args, -2 ratio not covered. */
int orc_pixbuf_scale(const uint8_t *src, int irow, int sw, int sh, uint8_t *dst, int orow, int dw, int dh, int channels, int interp);
int orc_pixbuf_scale_rows(const uint8_t *src, int irow, int sw, int sh, uint8_t *dst, int orow, int dw, int dh, int channels, int interp, int y0, int y1);
int *orc_pixbuf_weights(int interp, int sw, int sh, int dw, int dh, int *n_x, int *n_y, int *xoff, int *yoff);
void orc_pixbuf_free(void *p);
/* the palette resolution in front of resize_layer_full's body and the planner's queries (oracle/orc_resizable.c; src/colourspace.c:14500-14669, :14736-14740,
   :12128-12157), for a build without swscale; pinned by tests/golden/resizable.npz */
int orc_get_resizable(int *io);
int orc_get_tgt_gamma(int ipal, int opal);
int orc_can_inline_gamma(int inpl, int opal);
int orc_pconv_can_inplace(int inpl, int outpl);
/* B1 (UNPINNED, build-defined): separable [1 4 6 4 1]/16 per axis, edge replicate, one rounding */
void orc_gauss5(const uint8_t *src, int irow, uint8_t *dst, int orow, int width, int height, int psize);

/* the C5 / headline chain as the composition of the single ops above:
   BGRA32 -> RGBA32 (swap3postalpha) -> resize (interp) -> [gauss5] -> chroma blend(bf) with layer2 -> gamma LUT */
int orc_chain(const uint8_t *src, int irow, int sw, int sh, const uint8_t *layer2, int irow2,
              uint8_t *dst, int orow, int dw, int dh, int swap_rb, int interp, int do_blur, int bf,
              const uint8_t *lut8);

/* row-slice threaded drivers used by bench.py's cpu_baseline leg (reference slicing rule
   CEIL(height / n, 4), src/colourspace.c:9456-9485) */
int orc_chain_threaded(const uint8_t *src, int irow, int sw, int sh, const uint8_t *layer2, int irow2,
                       uint8_t *dst, int orow, int dw, int dh, int swap_rb, int interp, int do_blur, int bf,
                       const uint8_t *lut8, int nthreads);
double orc_bench_chain(int sw, int sh, int dw, int dh, int nthreads, int nframes, int do_blur);

#ifdef __cplusplus
}
#endif
#endif
