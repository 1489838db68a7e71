/* N host threads, each running the CONVERT chain of a plan step (src/nodemodel.c:1065-1253: palette -> gamma -> resize -> letterbox) over its own
 * layers through the layer-op seam -- LiVES' pool threads (src/threading.c) in miniature, without an interpreter lock in the timed region.
 * Built and driven by tools/bench_seam_mt.py and tests/test_layer_seam2.py. */
#define _GNU_SOURCE
#include <pthread.h>
#include <stddef.h>
#include <time.h>

typedef int (*convert_f)(void *layer, int outpl, int clamping);
typedef int (*gamma_f)(int gamma, void *layer);
typedef int (*resize_f)(void *layer, int w, int h, int interp, int opal, int oclamp);
typedef int (*letterbox_f)(void *layer, int nw, int nh, int w, int h, int interp, int tpal, int tclamp);
typedef int (*sync_f)(void *layer);
typedef int (*forget_f)(void *layer);

typedef struct {
  convert_f convert; gamma_f gamma; resize_f resize; letterbox_f letterbox; sync_f layer_sync;
  forget_f consume;         /* what stands for the frame's consumer on the device (the next plan step, the display): the planes go back to the seam's pool */
  int outpl, gamma_type, w, h, nw, nh;
} mt_calls;

typedef struct {
  const mt_calls *c;
  void **layers;            /* [1 + n]: layer 0 is this thread's warm-up (its first call creates the thread's stream and staging buffers) */
  int n, rc;
  pthread_barrier_t *start, *enq;
  double t_enqueued, t_done;
} mt_job;

static double now(void);
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

static __thread double t_call[4];          /* seconds this thread spent in each of the four calls (diagnostics) */
double g_call[4];
static pthread_mutex_t g_call_mu = PTHREAD_MUTEX_INITIALIZER;
static int chain(const mt_calls *c, void *layer) {
  if (c->outpl < 0) {                       /* diagnostic: four in-place calls (no plane is allocated or freed) on an RGB layer */
    for (int k = 0; k < 4; k++) if (!c->gamma((k & 1) ? 1 : 2, layer)) return 2;
    return 0;
  }
  double t0 = now(), t1;
  if (!c->convert(layer, c->outpl, 0)) return 1;
  t1 = now(); t_call[0] += t1 - t0; t0 = t1;
  if (!c->gamma(c->gamma_type, layer)) return 2;
  t1 = now(); t_call[1] += t1 - t0; t0 = t1;
  if (!c->resize(layer, c->w, c->h, 3, 0, 0)) return 3;
  t1 = now(); t_call[2] += t1 - t0; t0 = t1;
  if (!c->letterbox(layer, c->nw, c->nh, c->w, c->h, 3, 0, 0)) return 4;
  t1 = now(); t_call[3] += t1 - t0;
  return 0;
}

static void *worker(void *arg) {
  mt_job *j = (mt_job *)arg;
  j->rc = chain(j->c, j->layers[0]);
  if (!j->rc && j->c->layer_sync(j->layers[0])) j->rc = 5;
  pthread_barrier_wait(j->start);
  for (int k = 0; k < 4; k++) t_call[k] = 0;
  const double t0 = now();
  for (int i = 1; i <= j->n && !j->rc; i++) {
    j->rc = chain(j->c, j->layers[i]);
    if (!j->rc && i < j->n && j->c->consume) j->c->consume(j->layers[i]);
  }
  j->t_enqueued = now() - t0;
  pthread_mutex_lock(&g_call_mu);
  for (int k = 0; k < 4; k++) g_call[k] += t_call[k];
  pthread_mutex_unlock(&g_call_mu);
  if (!j->rc && j->c->layer_sync(j->layers[j->n])) j->rc = 5;          /* waits for this thread's stream: the last layer comes home */
  j->t_done = now() - t0;
  return NULL;
}

/* layers: nthreads x (1 + per) pinned layers; returns 0 and the wall time from the common start to the last thread's sync in *seconds */
int mt_run_chains(const mt_calls *c, void **layers, int nthreads, int per, double *seconds, double *enqueue_max) {
  pthread_t th[64];
  mt_job job[64];
  pthread_barrier_t start;
  if (nthreads < 1 || nthreads > 64) return -1;
  for (int k = 0; k < 4; k++) g_call[k] = 0;
  pthread_barrier_init(&start, NULL, (unsigned)nthreads + 1);
  for (int i = 0; i < nthreads; i++) {
    job[i].c = c; job[i].layers = layers + (size_t)i * (1 + per); job[i].n = per; job[i].rc = 0; job[i].start = &start;
    pthread_create(&th[i], NULL, worker, &job[i]);
  }
  pthread_barrier_wait(&start);
  const double t0 = now();
  int rc = 0;
  double enq = 0;
  for (int i = 0; i < nthreads; i++) {
    pthread_join(th[i], NULL);
    if (job[i].rc) rc = job[i].rc;
    if (job[i].t_enqueued > enq) enq = job[i].t_enqueued;
  }
  *seconds = now() - t0;
  if (enqueue_max) *enqueue_max = enq;
  pthread_barrier_destroy(&start);
  return rc;
}

/* ---- a frame allocator in the manner of LiVES' bigblock pool (src/memory.c: preallocated blocks handed out and taken back without a system call) ----
 * three size classes, a free stack per class under a mutex; bound as lives_gpu_weed_api.pixel_alloc / pixel_free by tools/bench_seam_mt.py */
#include <stdint.h>
#include <stdlib.h>
#include <sys/mman.h>

typedef struct { size_t bsize; int nblocks, ntop; uint8_t *base; void **stack; pthread_mutex_t mu; } mt_class;
static mt_class g_cls[3];

int mt_pool_init(const size_t *bsize, const int *nblocks) {
  for (int k = 0; k < 3; k++) {
    mt_class *c = &g_cls[k];
    c->bsize = bsize[k]; c->nblocks = nblocks[k]; c->ntop = 0;
    c->base = (uint8_t *)mmap(NULL, c->bsize * (size_t)c->nblocks, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    c->stack = (void **)malloc(sizeof(void *) * (size_t)c->nblocks);
    if (c->base == MAP_FAILED || !c->stack) return -1;
    pthread_mutex_init(&c->mu, NULL);
    for (int i = c->nblocks - 1; i >= 0; i--) c->stack[c->ntop++] = c->base + (size_t)i * c->bsize;
  }
  return 0;
}
void *mt_pool_alloc(size_t n) {
  for (int k = 0; k < 3; k++) {
    mt_class *c = &g_cls[k];
    if (n > c->bsize) continue;
    void *p = NULL;
    pthread_mutex_lock(&c->mu);
    if (c->ntop > 0) p = c->stack[--c->ntop];
    pthread_mutex_unlock(&c->mu);
    if (p) return p;
  }
  return calloc(1, n ? n : 1);
}
void mt_pool_free(void *p) {
  for (int k = 0; k < 3; k++) {
    mt_class *c = &g_cls[k];
    if (c->base && (uint8_t *)p >= c->base && (uint8_t *)p < c->base + c->bsize * (size_t)c->nblocks) {
      pthread_mutex_lock(&c->mu);
      c->stack[c->ntop++] = p;
      pthread_mutex_unlock(&c->mu);
      return;
    }
  }
  free(p);
}
