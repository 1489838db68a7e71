// dist.cpp -- the multi-GPU exchange of the path as plain C entry points (north_star: "host code stays C ... RCCL broadcast of
// shared transition params over xGMI"): one process per GPU, tracks sharded track-per-rank with NO data-path collective; what is exchanged
// is the 16-byte transition parameter block (lgpu_chain_params.param_block_d) per frame batch, an optional status word, and -- for the
// multitrack render -- the fan-in of the processed frames to the compositing rank.
//
// RCCL is bound at run time (dlopen of librccl.so.1, the copy already loaded by the process if there is one), so liblivesgpu.so keeps no
// link-time dependency on it and a single-GPU host never loads it.  Only the handful of entry points below are used; their prototypes are
// restated from rccl.h (ROCm 7.2: /opt/rocm/include/rccl/rccl.h:187-933).
#include <dlfcn.h>
#include <stdint.h>
#include <string.h>
#include <mutex>
#include "../../include/lives_gpu.h"

namespace lgpu { void set_error(const char *fmt, ...); }

namespace {
typedef void *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
enum { ncclUint8 = 1, ncclInt32 = 2 };
enum { ncclMax = 2 };
struct Rccl {
  void *h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Broadcast)(const void *, void *, size_t, int, int, ncclComm_t, void *) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, void *) = nullptr;
  ncclResult_t (*Send)(const void *, size_t, int, int, ncclComm_t, void *) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, int, int, ncclComm_t, void *) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_r;
std::mutex g_mu;

int bind_rccl(const char *path) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_r.h) return LGPU_OK;
  const char *names[] = {path, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void *h = nullptr;
  for (const char *n : {"librccl.so.1", "librccl.so"}) if (!path && !h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);     // the process's own copy first
  for (const char *n : names) if (n && !h) h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
  if (!h) { lgpu::set_error("lgpu_dist: librccl.so.1 not found (%s)", dlerror()); return LGPU_E_UNSUPPORTED; }
  Rccl r;
  r.h = h;
#define SYM(field, name) *(void **)(&r.field) = dlsym(h, name); if (!r.field) { lgpu::set_error("lgpu_dist: %s missing in librccl", name); return LGPU_E_UNSUPPORTED; }
  SYM(GetUniqueId, "ncclGetUniqueId") SYM(CommInitRank, "ncclCommInitRank") SYM(CommDestroy, "ncclCommDestroy") SYM(Broadcast, "ncclBroadcast")
  SYM(AllReduce, "ncclAllReduce") SYM(Send, "ncclSend") SYM(Recv, "ncclRecv") SYM(GroupStart, "ncclGroupStart") SYM(GroupEnd, "ncclGroupEnd")
  SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
  g_r = r;
  return LGPU_OK;
}
int check(ncclResult_t rc, const char *what) {
  if (rc == 0) return LGPU_OK;
  lgpu::set_error("lgpu_dist: %s failed: %s", what, g_r.GetErrorString ? g_r.GetErrorString(rc) : "?");
  return LGPU_E_HIP;
}
}  // namespace

extern "C" {

int lgpu_dist_bind(const char *rccl_path) { return bind_rccl(rccl_path); }

int lgpu_dist_unique_id(uint8_t id[LGPU_DIST_ID_BYTES]) {
  int rc = bind_rccl(nullptr);
  if (rc) return rc;
  if (!id) return LGPU_E_BADARG;
  ncclUniqueId u;
  if ((rc = check(g_r.GetUniqueId(&u), "ncclGetUniqueId"))) return rc;
  memcpy(id, u.internal, sizeof u.internal);
  return LGPU_OK;
}

int lgpu_dist_comm_create(const uint8_t id[LGPU_DIST_ID_BYTES], int rank, int world, void **comm) {
  int rc = bind_rccl(nullptr);
  if (rc) return rc;
  if (!id || !comm || world < 1 || rank < 0 || rank >= world) { lgpu::set_error("lgpu_dist_comm_create: bad arguments"); return LGPU_E_BADARG; }
  ncclUniqueId u;
  memcpy(u.internal, id, sizeof u.internal);
  ncclComm_t c = nullptr;
  if ((rc = check(g_r.CommInitRank(&c, world, u, rank), "ncclCommInitRank"))) return rc;
  *comm = c;
  return LGPU_OK;
}

int lgpu_dist_comm_destroy(void *comm) {
  if (!comm) return LGPU_OK;
  if (!g_r.h) return LGPU_E_BADARG;
  return check(g_r.CommDestroy((ncclComm_t)comm), "ncclCommDestroy");
}

// the shared transition parameter block (int32[4], device memory, in place) from `root` to every rank, stream ordered: the chain kernel of
// the batch reads it from the same device words (lgpu_chain_params.param_block_d), no host round trip
int lgpu_params_broadcast(void *comm, int root, int32_t *param_block_d, void *stream) {
  if (!comm || !param_block_d) { lgpu::set_error("lgpu_params_broadcast: null argument"); return LGPU_E_BADARG; }
  if (!g_r.h) return LGPU_E_BADARG;
  return check(g_r.Broadcast(param_block_d, param_block_d, 4, ncclInt32, root, (ncclComm_t)comm, stream), "ncclBroadcast");
}

// max of a status word over the ranks (0 = every rank's batch went through): the optional completion / error barrier of SURVEY 8e
int lgpu_status_allreduce(void *comm, int32_t *status_d, void *stream) {
  if (!comm || !status_d) { lgpu::set_error("lgpu_status_allreduce: null argument"); return LGPU_E_BADARG; }
  if (!g_r.h) return LGPU_E_BADARG;
  return check(g_r.AllReduce(status_d, status_d, 1, ncclInt32, ncclMax, (ncclComm_t)comm, stream), "ncclAllReduce");
}

// compositing fan-in: rank r owns tracks r, r + world, ... (track t -> rank t % world); each rank hands over its
// `nlocal` processed frames of frame_bytes each (contiguous, in the order of its tracks); `root` receives the ntracks frames in TRACK order
// into gathered_d (ntracks * frame_bytes).  Point-to-point sends over the xGMI links into the root inside one group, no staging copy:
// frame i of rank r lands at slot r + i * world.
int lgpu_fan_in(void *comm, int root, int rank, int world, int ntracks, const uint8_t *frames_d, size_t frame_bytes, uint8_t *gathered_d, void *stream) {
  if (!comm || world < 1 || rank < 0 || rank >= world || root < 0 || root >= world || ntracks < 1 || !frames_d || !frame_bytes || (rank == root && !gathered_d)) {
    lgpu::set_error("lgpu_fan_in: bad arguments");
    return LGPU_E_BADARG;
  }
  if (!g_r.h) return LGPU_E_BADARG;
  int rc = check(g_r.GroupStart(), "ncclGroupStart");
  if (rc) return rc;
  const int nlocal = (ntracks - rank + world - 1) / world;
  if (rank != root)
    for (int i = 0; i < nlocal && !rc; i++) rc = check(g_r.Send(frames_d + (size_t)i * frame_bytes, frame_bytes, ncclUint8, root, (ncclComm_t)comm, stream), "ncclSend");
  else
    for (int t = 0; t < ntracks && !rc; t++) {
      const int r = t % world, i = t / world;
      if (r == root) rc = lgpu_copy(gathered_d + (size_t)t * frame_bytes, frames_d + (size_t)i * frame_bytes, frame_bytes, stream);
      else rc = check(g_r.Recv(gathered_d + (size_t)t * frame_bytes, frame_bytes, ncclUint8, r, (ncclComm_t)comm, stream), "ncclRecv");
    }
  const int rc2 = check(g_r.GroupEnd(), "ncclGroupEnd");
  return rc ? rc : rc2;
}


// ---- the per-step host path of the multi-GPU batch as ONE C call (north_star: "host code stays C") -------------------------------------------
// A render worker runs, per frame batch: [parameter block of this batch has arrived] -> chain kernel.  The block of batch s + 1 is broadcast on a
// side stream while the kernel of batch s runs (two device blocks, one event each), so the launch stream never waits for xGMI and the host never
// synchronises: lgpu_chain_step = stream wait + event record + (root: one tiny launch) + ncclBroadcast + event record + the chain launch.
enum { kStepRing = 16, kStepFence = kStepRing / 2 };
struct lgpu_stepper {
  void *comm;                 // RCCL communicator, or NULL: one GPU, nothing to exchange (the block is written on the launch stream)
  int root, rank;
  void *launch, *side;        // hipStream_t
  void *ready[kStepRing], *tail;      // events: block s % ring has arrived / the launch stream's tail at the last fence
  int32_t *blk[kStepRing];
  long step;
};

// The exchange of step `step` on the side stream.  Its block, ring slot step % 16, was last read by the kernel of step - 16; every 8th step the side stream is
// ordered behind the launch stream's tail (which then has passed the kernel of step - 2 at least), so the slot is free when it is rewritten, and the two
// cross-stream calls of that ordering are paid once per 8 steps instead of every step (they cost more host time than the rest of the step together when every
// launch sits behind a fresh cross-stream barrier: tools/worker.c, profiles/r03/worker_step.md).
static int stepper_prefetch(lgpu_stepper *s, long step, const int32_t values[4]) {
  int rc;
  int32_t *b = s->blk[step % kStepRing];
  if (!s->comm) return lgpu_params_set(b, values, s->launch);            // stream order does the rest
  if (step % kStepFence == 0 && ((rc = lgpu_event_record(s->tail, s->launch)) || (rc = lgpu_stream_wait_event(s->side, s->tail)))) return rc;
  if (s->rank == s->root && (rc = lgpu_params_set(b, values, s->side))) return rc;
  if ((rc = lgpu_params_broadcast(s->comm, s->root, b, s->side))) return rc;
  return lgpu_event_record(s->ready[step % kStepRing], s->side);
}

int lgpu_stepper_create(void *comm, int root, int rank, void *launch_stream, const int32_t first_values[4], lgpu_stepper **out) {
  if (!out || !first_values || root < 0 || rank < 0) { lgpu::set_error("lgpu_stepper_create: bad arguments"); return LGPU_E_BADARG; }
  lgpu_stepper *s = new lgpu_stepper();
  s->comm = comm; s->root = root; s->rank = rank; s->launch = launch_stream; s->step = 0;
  int rc = LGPU_OK;
  void *p = nullptr;
  if ((rc = lgpu_malloc(&p, kStepRing * 4 * sizeof(int32_t)))) { delete s; return rc; }
  for (int i = 0; i < kStepRing; i++) s->blk[i] = (int32_t *)p + 4 * i;
  if (comm) {
    if (!rc) rc = lgpu_stream_create(&s->side, 1);
    for (int i = 0; i < kStepRing && !rc; i++) rc = lgpu_event_create(&s->ready[i]);
    if (!rc) rc = lgpu_event_create(&s->tail);
  }
  if (!rc) rc = stepper_prefetch(s, 0, first_values);
  if (rc) { lgpu_stepper_destroy(s); return rc; }
  *out = s;
  return LGPU_OK;
}

// one batch: the chain over `tracks` with the parameter block of this step (params->param_block_d is ignored: the stepper's block is used), and -- unless this
// is the last step (next_values == NULL on every rank) -- the exchange of the next step's block enqueued beside it.  next_values matters on the root only.
int lgpu_chain_step(lgpu_stepper *s, const int32_t next_values[4], const lgpu_chain_params *params, const lgpu_chain_track *tracks, int ntracks) {
  if (!s || !params) { lgpu::set_error("lgpu_chain_step: null argument"); return LGPU_E_BADARG; }
  static const int32_t zero[4] = {0, 0, 0, 0};
  int rc;
  if (s->comm && (rc = lgpu_stream_wait_event(s->launch, s->ready[s->step % kStepRing]))) return rc;
  if (next_values && (rc = stepper_prefetch(s, s->step + 1, s->rank == s->root ? next_values : zero))) return rc;
  lgpu_chain_params p = *params;
  p.param_block_d = s->blk[s->step % kStepRing];
  s->step++;
  return lgpu_chain(&p, tracks, ntracks, s->launch);
}

const int32_t *lgpu_stepper_block(const lgpu_stepper *s, int which) { return s ? s->blk[which % kStepRing] : nullptr; }

int lgpu_stepper_destroy(lgpu_stepper *s) {
  if (!s) return LGPU_OK;
  if (s->launch || true) lgpu_sync(s->launch);
  if (s->side) { lgpu_sync(s->side); lgpu_stream_destroy(s->side); }
  for (int i = 0; i < kStepRing; i++) if (s->ready[i]) lgpu_event_destroy(s->ready[i]);
  if (s->tail) lgpu_event_destroy(s->tail);
  if (s->blk[0]) lgpu_free(s->blk[0]);
  delete s;
  return LGPU_OK;
}

}  // extern "C"
