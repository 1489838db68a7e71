/* orc_bench.c -- row-slice threaded CPU runner of the headline chain, for bench.py's cpu_baseline leg.
 *
 * TEST INFRASTRUCTURE ONLY (see lives_oracle.h).  Produces byte-identical output to orc_chain()
 * (tests/test_oracle_cpu.py::test_threaded_chain_equals_serial) while slicing the OUTPUT rows across
 * threads with the reference's rule: rows_per_thread = CEIL(height / n, 4), slice 0 on the calling
 * thread (src/colourspace.c:9456-9485; nfx_threads defaults to ncpus, src/startup.c:595-601).
 */
#include "lives_oracle.h"
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct {
  const uint8_t *src, *l2;
  uint8_t *dst;
  int irow, irow2, orow, sw, sh, dw, dh, swap_rb, do_blur, bf, pixbuf_interp;      /* pixbuf_interp >= 0: the resize stage on gdk-pixbuf's arithmetic (orc_pixbuf.c) */
  const uint8_t *lut8;
  int nth, ntv;
  const int32_t *hpos, *vpos;
  const int16_t *hco, *vco;
  int y0, y1;
} job_t;

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }

/* resized rows [ry0, ry1) of the (swapped) source into out (row pitch dw*4); exact orc_resize arithmetic */
static void resize_rows(const job_t *j, int ry0, int ry1, uint8_t *out) {
  const int dw = j->dw, sw = j->sw, sh = j->sh;
  if (j->pixbuf_interp >= 0) {          /* LGPU_INTERP_PIXBUF: rows [ry0, ry1) by the pinned scaler; the R <-> B swap commutes with it and is done on its output */
    orc_pixbuf_scale_rows(j->src, j->irow, sw, sh, out, dw * 4, dw, j->dh, 4, j->pixbuf_interp, ry0, ry1);
    if (j->swap_rb)
      for (size_t i = 0; i < (size_t)(ry1 - ry0) * dw; i++) { const uint8_t t = out[4 * i]; out[4 * i] = out[4 * i + 2]; out[4 * i + 2] = t; }
    return;
  }
  const int s0 = j->vpos[ry0], s1 = j->vpos[ry1 - 1] + j->ntv;      /* unclamped source row span */
  const int nrows = s1 - s0;
  int16_t *tmp = malloc(sizeof(int16_t) * (size_t)nrows * dw * 4);
  for (int r = 0; r < nrows; r++) {
    const int sy = clampi(s0 + r, 0, sh - 1);
    const uint8_t *s = j->src + (size_t)sy * j->irow;
    int16_t *t = tmp + (size_t)r * dw * 4;
    for (int x = 0; x < dw; x++) {
      int32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
      const int16_t *c = j->hco + (size_t)x * j->nth;
      for (int k = 0; k < j->nth; k++) {
        const uint8_t *p = s + (size_t)clampi(j->hpos[x] + k, 0, sw - 1) * 4;
        const int cf = c[k];
        if (j->swap_rb) { a0 += cf * p[2]; a2 += cf * p[0]; } else { a0 += cf * p[0]; a2 += cf * p[2]; }
        a1 += cf * p[1]; a3 += cf * p[3];
      }
      t[x * 4 + 0] = (int16_t)clampi((a0 + 64) >> 7, -32768, 32767);
      t[x * 4 + 1] = (int16_t)clampi((a1 + 64) >> 7, -32768, 32767);
      t[x * 4 + 2] = (int16_t)clampi((a2 + 64) >> 7, -32768, 32767);
      t[x * 4 + 3] = (int16_t)clampi((a3 + 64) >> 7, -32768, 32767);
    }
  }
  for (int y = ry0; y < ry1; y++) {
    uint8_t *d = out + (size_t)(y - ry0) * dw * 4;
    const int16_t *c = j->vco + (size_t)y * j->ntv;
    for (int x = 0; x < dw * 4; x++) {
      int32_t acc = 0;
      for (int k = 0; k < j->ntv; k++) {
        /* clamped source row -> row of tmp; rows outside [0, sh) were staged as replicas above */
        acc += (int32_t)c[k] * tmp[(size_t)(j->vpos[y] + k - s0) * dw * 4 + x];
      }
      d[x] = (uint8_t)clampi((acc + (1 << 20)) >> 21, 0, 255);
    }
  }
  free(tmp);
}

static void *run_slice(void *arg) {
  const job_t *j = (const job_t *)arg;
  const int dw = j->dw, dh = j->dh, y0 = j->y0, y1 = j->y1;
  if (y0 >= y1) return NULL;
  /* the blur needs two more resized rows on each side (edge replicate at the frame border) */
  const int ry0 = j->do_blur ? clampi(y0 - 2, 0, dh) : y0, ry1 = j->do_blur ? clampi(y1 + 2, 0, dh) : y1;
  uint8_t *rs = malloc((size_t)(ry1 - ry0) * dw * 4);
  resize_rows(j, ry0, ry1, rs);
  uint8_t *trk = rs + (size_t)(y0 - ry0) * dw * 4;
  uint8_t *bl = NULL;
  if (j->do_blur) {
    static const int kw[5] = {1, 4, 6, 4, 1};
    const int rowlen = dw * 4;
    uint16_t *hs = malloc(sizeof(uint16_t) * (size_t)(ry1 - ry0) * rowlen);
    for (int r = 0; r < ry1 - ry0; r++) {
      const uint8_t *s = rs + (size_t)r * rowlen;
      for (int x = 0; x < dw; x++)
        for (int c = 0; c < 4; c++) {
          int acc = 0;
          for (int k = -2; k <= 2; k++) acc += kw[k + 2] * s[clampi(x + k, 0, dw - 1) * 4 + c];
          hs[(size_t)r * rowlen + x * 4 + c] = (uint16_t)acc;
        }
    }
    bl = malloc((size_t)(y1 - y0) * rowlen);
    for (int y = y0; y < y1; y++)
      for (int x = 0; x < rowlen; x++) {
        int acc = 0;
        for (int k = -2; k <= 2; k++) acc += kw[k + 2] * hs[(size_t)(clampi(y + k, 0, dh - 1) - ry0) * rowlen + x];
        bl[(size_t)(y - y0) * rowlen + x] = (uint8_t)((acc + 128) >> 8);
      }
    free(hs);
    trk = bl;
  }
  orc_blend_chroma(trk, dw * 4, j->l2 + (size_t)y0 * j->irow2, j->irow2, trk, dw * 4, dw, y1 - y0, 4, 0, j->bf);
  if (j->lut8) orc_gamma_apply(trk, dw * 4, dw, y1 - y0, 4, 0, j->lut8);
  for (int y = y0; y < y1; y++) memcpy(j->dst + (size_t)y * j->orow, trk + (size_t)(y - y0) * dw * 4, (size_t)dw * 4);
  free(rs); free(bl);
  return NULL;
}

/* threaded chain; same result as orc_chain().  returns 0 / -1 */
int orc_chain_threaded(const uint8_t *src, int irow, int sw, int sh, const uint8_t *layer2, int irow2,
                       uint8_t *dst, int orow, int dw, int dh, int swap_rb, int interp, int do_blur, int bf,
                       const uint8_t *lut8, int nthreads) {
  const int pixbuf_interp = (interp & 0x100) ? (interp & 0xFF) : -1;
  interp &= 0xFF;
  const int kernel = (interp == ORC_INTERP_HYPER) ? ((dw > sw || dh > sh) ? 2 : 1) : 0;
  int32_t *hpos = malloc(sizeof(int32_t) * dw), *vpos = malloc(sizeof(int32_t) * dh);
  int16_t *hco = malloc(sizeof(int16_t) * (size_t)dw * 256), *vco = malloc(sizeof(int16_t) * (size_t)dh * 256);
  int nth = 0, ntv = 0, rc = -1;
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 256) nthreads = 256;
  if (orc_make_filter(sw, dw, kernel, &nth, hpos, hco, 256) || orc_make_filter(sh, dh, kernel, &ntv, vpos, vco, 256)) goto out;
  {
    job_t jobs[256];
    pthread_t th[256];
    /* CEIL(height / n, 4) rows per slice */
    int per = (dh + nthreads - 1) / nthreads;
    per = (per + 3) & ~3;
    int used = 0;
    for (int i = 0; i < nthreads; i++) {
      job_t *j = &jobs[i];
      j->src = src; j->l2 = layer2; j->dst = dst; j->irow = irow; j->irow2 = irow2; j->orow = orow;
      j->sw = sw; j->sh = sh; j->dw = dw; j->dh = dh; j->swap_rb = swap_rb; j->do_blur = do_blur; j->bf = bf; j->lut8 = lut8;
      j->pixbuf_interp = pixbuf_interp;
      j->nth = nth; j->ntv = ntv; j->hpos = hpos; j->vpos = vpos; j->hco = hco; j->vco = vco;
      j->y0 = i * per; j->y1 = (i + 1) * per > dh ? dh : (i + 1) * per;
      if (j->y0 >= dh) break;
      used++;
    }
    for (int i = 1; i < used; i++) pthread_create(&th[i], NULL, run_slice, &jobs[i]);
    run_slice(&jobs[0]);
    for (int i = 1; i < used; i++) pthread_join(th[i], NULL);
  }
  rc = 0;
out:
  free(hpos); free(vpos); free(hco); free(vco);
  return rc;
}

double orc_bench_chain2(int sw, int sh, int dw, int dh, int nthreads, int nframes, int do_blur, int interp);
static uint32_t xs32(uint32_t *s) { uint32_t x = *s; x ^= x << 13; x ^= x >> 17; x ^= x << 5; return *s = x; }

/* times `nframes` passes of the chain on synthetic frames (seed 0x11FE5); returns seconds, or < 0 */
double orc_bench_chain(int sw, int sh, int dw, int dh, int nthreads, int nframes, int do_blur) { return orc_bench_chain2(sw, sh, dw, dh, nthreads, nframes, do_blur, ORC_INTERP_HYPER); }

/* interp may carry 0x100 (LGPU_INTERP_PIXBUF) */
double orc_bench_chain2(int sw, int sh, int dw, int dh, int nthreads, int nframes, int do_blur, int interp) {
  const size_t sb = (size_t)sw * 4 * sh, db = (size_t)dw * 4 * dh;
  uint8_t *src = malloc(sb), *l2 = malloc(db), *dst = malloc(db), lut[256];
  uint32_t seed = 0x11FE5;
  struct timespec t0, t1;
  if (!src || !l2 || !dst) return -1.;
  for (size_t i = 0; i < sb; i++) src[i] = (uint8_t)(xs32(&seed) >> 24);
  for (size_t i = 0; i < db; i++) l2[i] = (uint8_t)(xs32(&seed) >> 24);
  for (size_t i = 3; i < db; i += 8) l2[i] = 255;          /* half of layer 2 opaque */
  orc_gamma_lut8(1.0, -1, 1, 1.4, lut);
  orc_chain_threaded(src, sw * 4, sw, sh, l2, dw * 4, dst, dw * 4, dw, dh, 1, interp, do_blur, 128, lut, nthreads);  /* warm-up */
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int f = 0; f < nframes; f++)
    if (orc_chain_threaded(src, sw * 4, sw, sh, l2, dw * 4, dst, dw * 4, dw, dh, 1, interp, do_blur, 128 + (f & 1), lut, nthreads)) return -1.;
  clock_gettime(CLOCK_MONOTONIC, &t1);
  free(src); free(l2); free(dst);
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
