/* tools/worker.c -- the render-worker loop of one rank in plain C on liblivesgpu.so's C ABI (north_star: "host code stays C"):
 *
 *     for every frame batch:  lgpu_chain_step()  =  wait for this batch's parameter block, send the next one beside the kernel, launch the chain
 *
 * One process per GPU.  World size 1 runs the whole host path -- RCCL communicator, side stream, events, broadcast -- on a single GPU (a one-rank
 * communicator), which is what the pool's one-GPU boxes can measure: host microseconds per step and frames / s at one 4K frame per step
 * (BASELINE config 5's per-GPU shape) with the exchange ON.  With WORLD_SIZE > 1 (RANK, LOCAL_RANK, LGPU_ID_FILE in the environment) rank 0
 * writes the 128-byte communicator id to LGPU_ID_FILE and the others read it: no Python, no MPI.
 *
 * build: gcc -O2 -o tools/_worker tools/worker.c -Iinclude -Llives_amd -llivesgpu -Wl,-rpath,$PWD/lives_amd
 * run  : tools/_worker [--tracks 1] [--steps 2000] [--exchange 1] [--pixbuf 1] [--ahead 16] [--overlap 1] [--sets N]      (--ahead n: the blocks of n steps per exchange, lgpu_stepper_feed;
 *                                                                                                  1 = one block ahead, through lgpu_chain_step's next_values)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include "lives_gpu.h"

#define CHECK(x) do { int rc_ = (x); if (rc_) { fprintf(stderr, "%s failed: %d (%s)\n", #x, rc_, lgpu_last_error()); exit(1); } } while (0)
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static int envi(const char *k, int d) { const char *v = getenv(k); return v ? atoi(v) : d; }

int main(int argc, char **argv) {
  int tracks = 1, steps = 2000, exchange = 1, pixbuf = 1, ahead = 16, overlap = 1, sets = 0;
  for (int i = 1; i + 1 < argc; i += 2) {
    if (!strcmp(argv[i], "--tracks")) tracks = atoi(argv[i + 1]);
    else if (!strcmp(argv[i], "--steps")) steps = atoi(argv[i + 1]);
    else if (!strcmp(argv[i], "--exchange")) exchange = atoi(argv[i + 1]);
    else if (!strcmp(argv[i], "--pixbuf")) pixbuf = atoi(argv[i + 1]);
    else if (!strcmp(argv[i], "--ahead")) ahead = atoi(argv[i + 1]);
    else if (!strcmp(argv[i], "--overlap")) overlap = atoi(argv[i + 1]);
    else if (!strcmp(argv[i], "--sets")) sets = atoi(argv[i + 1]);
  }
  const int rank = envi("RANK", 0), world = envi("WORLD_SIZE", 1), local = envi("LOCAL_RANK", rank);
  const int SW = 3840, SH = 2160, DW = 1920, DH = 1080;
  /* rotating buffer sets: 32 frames of each kind in all (1.6 GB) unless --sets says otherwise -- with two sets of one frame every byte stays in the 256 MiB memory-side cache
     from one step to the next (8.8 instead of 10.2 us per one-frame step), which a frame that was just uploaded does not (profiles/r04/sets_ab.txt) */
  const int NSETS = sets > 0 ? sets : (32 / tracks > 2 ? 32 / tracks : 2);
  CHECK(lgpu_init(local));
  void *stream, *stream2 = NULL;
  CHECK(lgpu_stream_create(&stream, 1));
  /* the second launch stream right behind the first: HIP deals its few hardware queues out to streams in creation order, and two launch streams that end up on
     the same queue do not overlap (measured: created after the communicator's streams they shared one) */
  if (overlap) CHECK(lgpu_stream_create(&stream2, 1));

  /* device-resident synthetic tracks, NSETS rotating sets */
  lgpu_chain_track *trk = calloc((size_t)NSETS * tracks, sizeof *trk);
  uint64_t *hostbuf = malloc((size_t)SW * SH * 4), rng = 0x11FE5ull + (uint64_t)rank;
  for (int i = 0; i < NSETS * tracks; i++) {
    void *s, *l, *d;
    CHECK(lgpu_malloc(&s, (size_t)SW * SH * 4)); CHECK(lgpu_malloc(&l, (size_t)DW * DH * 4)); CHECK(lgpu_malloc(&d, (size_t)DW * DH * 4));
    /* uniform random bytes (xorshift64), half of layer 2 opaque -- bench.py's synthetic frames; constant fills run measurably faster and are not used */
    for (size_t k = 0; k < (size_t)SW * SH / 2; k++) { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; hostbuf[k] = rng; }
    CHECK(lgpu_upload(s, hostbuf, (size_t)SW * SH * 4, stream)); CHECK(lgpu_sync(stream));
    for (size_t k = 0; k < (size_t)DW * DH / 2; k++) { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; hostbuf[k] = rng | ((rng & 0x100) ? 0xFF000000FF000000ull : 0); }
    CHECK(lgpu_upload(l, hostbuf, (size_t)DW * DH * 4, stream)); CHECK(lgpu_sync(stream));
    trk[i].src_d = s; trk[i].layer2_d = l; trk[i].dst_d = d;
  }
  lgpu_chain_params prm;
  memset(&prm, 0, sizeof prm);
  prm.sw = SW; prm.sh = SH; prm.irow = SW * 4; prm.dw = DW; prm.dh = DH; prm.irow2 = DW * 4; prm.orow = DW * 4;
  prm.swap_rb = 1; prm.interp = 3 | (pixbuf ? LGPU_INTERP_PIXBUF : 0); prm.bf = 128; prm.use_lut = 1;
  if (lgpu_gamma_lut8(1.0, -1, 1, 1.4, prm.lut8) != 1) { fprintf(stderr, "no gamma LUT\n"); return 1; }

  void *comm = NULL;
  if (exchange) {
    uint8_t id[LGPU_DIST_ID_BYTES];
    const char *idf = getenv("LGPU_ID_FILE");
    if (rank == 0) {
      CHECK(lgpu_dist_unique_id(id));
      if (world > 1) { FILE *f = fopen(idf, "wb"); fwrite(id, 1, sizeof id, f); fclose(f); char done[512]; snprintf(done, sizeof done, "%s.ok", idf); f = fopen(done, "w"); fclose(f); }
    } else {
      char done[512]; snprintf(done, sizeof done, "%s.ok", idf);
      while (access(done, F_OK)) usleep(1000);
      FILE *f = fopen(idf, "rb"); if (fread(id, 1, sizeof id, f) != sizeof id) return 1; fclose(f);
    }
    CHECK(lgpu_dist_comm_create(id, rank, world, &comm));
  }
  if (getenv("LGPU_WORKER_PROBE") && comm) {      /* host cost of the primitives a step is made of (idle queue, 1000 calls each) */
    void *side, *ev, *blk;
    int32_t pv[4] = {1, 2, 3, 4};
    CHECK(lgpu_stream_create(&side, 1)); CHECK(lgpu_event_create(&ev)); CHECK(lgpu_malloc(&blk, 16));
    const char *names[] = {"lgpu_event_record", "lgpu_stream_wait_event", "lgpu_params_set", "lgpu_params_broadcast", "lgpu_chain (1 track)"};
    for (int what = 0; what < 5; what++) {
      CHECK(lgpu_sync(stream)); CHECK(lgpu_sync(side));
      double acc = 0;
      for (int i = 0; i < 1000; i++) {
        const double a = now();
        if (what == 0) CHECK(lgpu_event_record(ev, stream));
        else if (what == 1) CHECK(lgpu_stream_wait_event(side, ev));
        else if (what == 2) CHECK(lgpu_params_set(blk, pv, side));
        else if (what == 3) CHECK(lgpu_params_broadcast(comm, 0, blk, side));
        else CHECK(lgpu_chain(&prm, trk, 1, stream));
        acc += now() - a;
        if ((i & 31) == 31) { CHECK(lgpu_sync(stream)); CHECK(lgpu_sync(side)); }
      }
      printf("probe %-24s %.2f us per call\n", names[what], acc / 1000 * 1e6);
    }
  }
  int32_t v[4] = {96, 0, 0, 0};
  if (ahead < 1) ahead = 1;
  if (ahead > 32) ahead = 32;
  lgpu_stepper *st;
  CHECK(lgpu_stepper_create(comm, 0, rank, stream, v, &st));
  if (overlap) CHECK(lgpu_stepper_overlap(st, stream2));      /* odd steps on the second launch stream */
#define SYNC() do { CHECK(lgpu_sync(stream)); if (stream2) CHECK(lgpu_sync(stream2)); } while (0)
  long fed = 1;                                  /* blocks handed over so far (the first one by lgpu_stepper_create) */
  int32_t sched[32 * 4];
  /* one step: feed the blocks of the next `ahead` steps when the last fed one is about to be used, then launch */
#define STEP(s_) do { \
    if (ahead > 1 && fed == (s_) + 1) { memset(sched, 0, sizeof sched); for (int j = 0; j < ahead; j++) sched[4 * j] = (96 + 7 * (int)(fed + j)) & 255; CHECK(lgpu_stepper_feed(st, sched, ahead)); fed += ahead; } \
    v[0] = (96 + 7 * ((s_) + 1)) & 255; \
    CHECK(lgpu_chain_step(st, ahead > 1 ? NULL : v, &prm, trk + ((s_) % NSETS) * tracks, tracks)); } while (0)
  long s = 0;
  for (; s < 300; s++) STEP(s);                  /* device wake-up */
  SYNC();
  const double t0 = now();
  for (; s < 300 + steps; s++) STEP(s);
  const double t_enq = now();
  SYNC();
  const double t1 = now();
  /* host cost alone: the same calls with the stream left to drain in between (enqueue never blocks on a full queue) */
  double host = 0;
  for (int i = 0; i < 256; i++, s++) {
    const double a = now();
    STEP(s);
    host += now() - a;
    if ((i & 15) == 15) SYNC();
  }
  SYNC();
  if (rank == 0)
    printf("{\"tool\": \"worker.c\", \"world\": %d, \"tracks_per_step\": %d, \"exchange\": \"%s\", \"resize\": \"%s\", \"blocks_per_exchange\": %d, \"launch_streams\": %d, \"buffer_sets\": %d, \"steps\": %d, \"us_per_step\": %.2f, "
           "\"frames_per_s_per_gpu\": %.0f, \"enqueue_us_per_step\": %.2f, \"host_us_per_step_idle_queue\": %.2f}\n",
           world, tracks, exchange ? (world > 1 ? "rccl broadcast" : "rccl broadcast (one-rank communicator)") : "none", pixbuf ? "pixbuf" : "polyphase", ahead, overlap ? 2 : 1, NSETS, steps,
           (t1 - t0) / steps * 1e6, tracks * steps / (t1 - t0), (t_enq - t0) / steps * 1e6, host / 256 * 1e6);
  CHECK(lgpu_stepper_destroy(st));
  if (comm) CHECK(lgpu_dist_comm_destroy(comm));
  return 0;
}
