#!/usr/bin/env python3
"""Build oracle/_ref/libresizableref.so: the reference's palette-resolution prologue of resize_layer_full and the planner's
capability queries, as line-range slices of /root/reference/src/colourspace.c (built WITHOUT USE_SWSCALE: the gdk-pixbuf build).

TEST INFRASTRUCTURE ONLY.  Runs only in the build container (needs /root/reference).  The scratch translation unit goes to
oracle/_ref/ (git-ignored); what is committed is this recipe and the table it produces (tests/golden/resizable.npz, written by
gen_golden_resizable.py).

Sliced, unmodified:
  advp[] / init_advanced_palettes / get_advanced_palette      src/colourspace.c:1535-1717
  is_rgbchan / is_yuvchan / macropixel sizes / nplanes        :1728-1780
  weed_palette_is_rgb / _is_yuv / _has_alpha                  :1820-1845
  weed_palette_conv_resizable / weed_palette_is_resizable     :2596-2654   (the #else branch: no swscale)
  can_inline_gamma / pconv_can_inplace                        :12128-12157
  get_masq_pal / get_inter_pal / get_resizable                :14500-14669
  get_tgt_gamma                                               :14736-14740
The prelude below supplies typedefs, the constants of src/defs.h / src/colourspace.h those lines name, and -- the one behavioural
stand-in -- LIVES_FATAL as "set a flag" (the reference aborts there; the wrapper reports it as result -1).
WEED_CLAMPING_UNCLAMPED (:2620) is spelt WEED_YUV_CLAMPING_UNCLAMPED everywhere else in the reference; the value is written to two
locals nobody reads."""
import os
import subprocess
import sys

REF = os.environ.get("LIVES_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.normpath(os.path.join(HERE, "..", "_ref"))
# scratch translation units and the include-path symlinks stay outside the repo (build_ref.sh sets LIVES_REF_WORK); only the .so lands in oracle/_ref
WORK = os.environ.get("LIVES_REF_WORK", os.path.join(os.environ.get("TMPDIR", "/tmp"), "lives_ref_work"))


def lines(path, a, b):
    with open(os.path.join(REF, path), "r", errors="replace") as f:
        all_lines = f.readlines()
    return "/* ---- %s:%d-%d ---- */\n" % (path, a, b) + "".join(all_lines[a - 1:b]) + "\n"


PRELUDE = r'''
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
typedef int boolean;
#ifndef TRUE
#define TRUE 1
#define FALSE 0
#endif
typedef int lives_result_t;
#define LIVES_GLOBAL_INLINE
#define LIVES_LOCAL_INLINE static inline
#define lives_memset memset
#define lives_free(p) ((void)(p))
#define WEED_ADVANCED_PALETTES 1       /* what LiVES builds with (src/main.h) */
#include <weed/weed-palettes.h>
#define WEED_CLAMPING_UNCLAMPED WEED_YUV_CLAMPING_UNCLAMPED
#define weed_palette_is_planar(pal) (weed_palette_get_nplanes(pal) > 1)      /* libweed/weed-host-utils.h:356 */
#define pixel_size(pal) ((int)weed_palette_get_bytes_per_macropixel(pal))    /* src/colourspace.h:310 */
static int resizable_fatal;
#define LIVES_FATAL(msg) (resizable_fatal = 1)
#define lives_strdup_printf(...) ((char *)"")
#define _(s) (s)
#define weed_palette_get_name(p) ""
const weed_macropixel_t *get_advanced_palette(int weed_palette);
int weed_palette_get_nplanes(int pal);
double weed_palette_get_bytes_per_macropixel(int pal);
boolean weed_palette_is_rgb(int pal);
boolean weed_palette_is_yuv(int pal);
boolean weed_palette_has_alpha(int pal);
'''

WRAPPERS = r'''
/* ---- extern wrappers (this repo's code; they only forward) ---- */
static int inited;
static void ensure(void) { if (!inited) { init_advanced_palettes(); inited = 1; } }
/* io[5] in: palette, opal_hint, oclamp_hint, upscale ; out: io[0] resolved, io[1] xpalette, io[2] oclamp_hint, io[3] opal_hint, io[4] xopal_hint.
   Returns LIVES_RESULT_SUCCESS (1) / LIVES_RESULT_FAIL (0), or -1 where the reference reaches LIVES_FATAL. */
int rsref_get_resizable(int *io) {
  int palette = io[0], opal = io[1], oclamp = io[2], upscale = io[3], xpal = 0, xopal = 0;
  ensure();
  resizable_fatal = 0;
  int r = get_resizable(&palette, &xpal, &oclamp, &opal, &xopal, upscale);
  if (resizable_fatal) return -1;
  io[0] = palette; io[1] = xpal; io[2] = oclamp; io[3] = opal; io[4] = xopal;
  return r;
}
int rsref_can_inline_gamma(int inpl, int opal) { ensure(); return can_inline_gamma(inpl, opal); }
int rsref_pconv_can_inplace(int inpl, int outpl) { ensure(); return pconv_can_inplace(inpl, outpl); }
int rsref_get_tgt_gamma(int ipal, int opal) { ensure(); return get_tgt_gamma(ipal, opal); }
int rsref_is_resizable(int pal, int direction) { ensure(); return weed_palette_is_resizable(pal, WEED_YUV_CLAMPING_UNCLAMPED, direction); }
'''


def main():
    if not os.path.isdir(os.path.join(REF, "src")):
        print("build_resizable_slice.py: %s not present -- keeping prebuilt oracle/_ref" % REF, file=sys.stderr)
        return 0
    os.makedirs(OUT, exist_ok=True)
    os.makedirs(WORK, exist_ok=True)
    src = PRELUDE
    src += lines("src/defs.h", 211, 212) + lines("src/defs.h", 467, 469)
    src += lines("src/colourspace.h", 291, 301)
    c = "src/colourspace.c"
    for a, b in ((1535, 1717), (1728, 1780), (1820, 1845), (2596, 2654), (12128, 12157), (14500, 14669), (14736, 14740)):
        src += lines(c, a, b)
    src += WRAPPERS
    tu = os.path.join(WORK, "resizable_slice.c")
    with open(tu, "w") as f:
        f.write(src)
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-O1", "-w", "-I" + os.path.join(WORK, "inc"), "-o", os.path.join(OUT, "libresizableref.so"), tu])
    print(os.path.join(OUT, "libresizableref.so"))
    return 0


if __name__ == "__main__":
    sys.exit(main())
