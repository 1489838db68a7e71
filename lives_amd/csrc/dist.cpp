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
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
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
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;       // optional
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
  *(void **)(&r.CommCount) = dlsym(h, "ncclCommCount");
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

// ncclCommInitRank on a helper thread, the caller waits for it with a limit: version-independent (no ncclConfig_t, whose layout moves between RCCL releases)
int lgpu_dist_comm_create_timeout(const uint8_t id[LGPU_DIST_ID_BYTES], int rank, int world, int timeout_ms, void **comm) {
  if (timeout_ms <= 0) return lgpu_dist_comm_create(id, rank, world, comm);
  int rc = bind_rccl(nullptr);
  if (rc) return rc;
  if (!id || !comm || world < 1 || rank < 0 || rank >= world) { lgpu::set_error("lgpu_dist_comm_create_timeout: bad arguments"); return LGPU_E_BADARG; }
  int dev = 0;
  if ((rc = lgpu_current_device(&dev))) return rc;
  struct Job { std::mutex mu; std::condition_variable cv; bool done = false; ncclResult_t res = 0; ncclComm_t c = nullptr; };
  auto job = std::make_shared<Job>();
  ncclUniqueId u;
  memcpy(u.internal, id, sizeof u.internal);
  std::thread([job, u, rank, world, dev]() {
    (void)lgpu_set_device(dev);                          // the communicator belongs to the caller's device
    ncclComm_t c = nullptr;
    const ncclResult_t r = g_r.CommInitRank(&c, world, u, rank);
    std::lock_guard<std::mutex> lk(job->mu);
    job->res = r; job->c = c; job->done = true;
    job->cv.notify_all();
  }).detach();
  std::unique_lock<std::mutex> lk(job->mu);
  if (!job->cv.wait_for(lk, std::chrono::milliseconds(timeout_ms), [&] { return job->done; })) {
    lgpu::set_error("lgpu_dist_comm_create_timeout: rank %d of %d waited %d ms in ncclCommInitRank: a rank of the job has not arrived (not started, crashed before its "
                    "rendezvous, or another id / world size)", rank, world, timeout_ms);
    return LGPU_E_TIMEOUT;
  }
  if ((rc = check(job->res, "ncclCommInitRank"))) return rc;
  *comm = job->c;
  return LGPU_OK;
}

int lgpu_dist_comm_count(void *comm) {
  if (!comm || !g_r.h) { lgpu::set_error("lgpu_dist_comm_count: no communicator"); return LGPU_E_BADARG; }
  if (!g_r.CommCount) { lgpu::set_error("lgpu_dist_comm_count: ncclCommCount missing in librccl"); return LGPU_E_UNSUPPORTED; }
  int n = 0;
  const int rc = check(g_r.CommCount((ncclComm_t)comm, &n), "ncclCommCount");
  return rc ? rc : n;
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


// the same for `nblocks` consecutive blocks in one exchange (the schedule of a render is known ahead: lgpu_stepper_feed)
int lgpu_params_broadcast_n(void *comm, int root, int32_t *param_blocks_d, int nblocks, void *stream) {
  if (!comm || !param_blocks_d || nblocks < 1) { lgpu::set_error("lgpu_params_broadcast_n: bad argument"); return LGPU_E_BADARG; }
  if (!g_r.h) return LGPU_E_BADARG;
  return check(g_r.Broadcast(param_blocks_d, param_blocks_d, 4 * (size_t)nblocks, ncclInt32, root, (ncclComm_t)comm, stream), "ncclBroadcast");
}

// ---- the per-step host path of the multi-GPU batch as ONE C call (north_star: "host code stays C") -------------------------------------------
// A render worker runs, per frame batch: [parameter block of this batch has arrived] -> chain kernel.  Blocks travel AHEAD of the kernels that read them, on a
// side stream, into a ring of 64 device blocks: lgpu_stepper_feed() hands over the blocks of the next n steps (a render knows its schedule; a live parameter
// is one feed of latency either way) -- on the root one tiny launch that writes them, then ONE ncclBroadcast of n x 16 bytes, one event.  lgpu_chain_step()
// = [stream wait on the feed that brought this step's block, the first time a step of that feed comes up] + the chain launch: per step the host pays the
// launch and 1 / n of an exchange; the launch stream never waits for xGMI and the host never synchronises.  lgpu_chain_step with next_values is the n = 1 form.
enum { kStepRing = 64 };
struct lgpu_stepper {
  void *comm;                 // RCCL communicator, or NULL: one GPU, nothing to exchange (blocks are written on the launch stream)
  int root, rank;
  void *launch, *side;        // hipStream_t
  void *launch2, *tail2;      // lgpu_stepper_overlap: odd steps go to this stream, so that the drain of one frame's launch overlaps the ramp-up of the next
  void *ev[kStepRing];        // event of feed f at ev[f % ring] (a ring slot holds at most one unconsumed feed: feeds carry >= 1 block each and <= ring blocks are unconsumed)
  void *tail;                 // the launch stream's tail at the last fence
  int32_t *blk;               // ring of kStepRing blocks
  long fed, step, feeds;      // blocks handed over / consumed by a launch, feeds so far
  long feed_of[kStepRing];    // which feed brought the block in ring slot i
  long waited, waited2;       // the newest feed each launch stream has been ordered behind
  long fenced;                // every launch of a step < fenced is ordered before what the side stream does next
  int failed;                 // a step or feed failed half way: the stepper is out of step with its peers and refuses further calls
};

// n >= 1 blocks for steps fed .. fed + n - 1 (values: n x 4 ints, read on the root only; NULL elsewhere is fine).  Every rank calls it with the same n.
int lgpu_stepper_feed(lgpu_stepper *s, const int32_t *values, int n) {
  if (!s || n < 1 || n > kStepRing) { lgpu::set_error("lgpu_stepper_feed: bad arguments"); return LGPU_E_BADARG; }
  if (s->failed) { lgpu::set_error("lgpu_stepper_feed: an earlier call failed half way; the stepper is out of step with its peers: destroy it"); return LGPU_E_STATE; }
  if (s->fed + n - s->step > kStepRing) { lgpu::set_error("lgpu_stepper_feed: %ld blocks are waiting for their steps; the ring holds %d", s->fed - s->step, (int)kStepRing); return LGPU_E_BADARG; }
  const bool is_root = s->rank == s->root;
  if (is_root && !values) { lgpu::set_error("lgpu_stepper_feed: the root needs values"); return LGPU_E_BADARG; }
  int rc = LGPU_OK;
  void *st = s->comm ? s->side : s->launch;
  // the ring slots of blocks fed .. fed + n - 1 were last read by the kernels of steps fed - ring .. fed + n - 1 - ring: the side stream must be behind them.  A fence
  // orders it behind EVERY launch made so far, so one is due only about once per ring (two host calls, paid every ~60 steps instead of every step)
  if (s->comm && s->fenced < s->fed + n - kStepRing) {
    if ((rc = lgpu_event_record(s->tail, s->launch)) || (rc = lgpu_stream_wait_event(s->side, s->tail))) return rc;        // nothing has changed yet: the call can be repeated
    if (s->launch2 && ((rc = lgpu_event_record(s->tail2, s->launch2)) || (rc = lgpu_stream_wait_event(s->side, s->tail2)))) return rc;
    s->fenced = s->step;
  }
  if (!s->comm && s->launch2) {
    // no communicator: the blocks are written on the first launch stream -- behind whatever the second one still reads from the slots ...
    if ((rc = lgpu_event_record(s->tail2, s->launch2)) || (rc = lgpu_stream_wait_event(s->launch, s->tail2))) return rc;
  }
  for (int done = 0; done < n && !rc;) {        // at most two contiguous runs (the ring wraps)
    const long first = s->fed + done;
    const int slot = (int)(first % kStepRing), run = n - done < kStepRing - slot ? n - done : kStepRing - slot;
    if (is_root) rc = lgpu_params_set_n(s->blk + 4 * slot, values + 4 * done, run, st);
    if (!rc && s->comm) rc = lgpu_params_broadcast_n(s->comm, s->root, s->blk + 4 * slot, run, st);
    done += run;
  }
  if (!rc && s->comm) rc = lgpu_event_record(s->ev[s->feeds % kStepRing], s->side);
  if (!rc && !s->comm && s->launch2) {           // ... and the second launch stream behind the writes
    if (!(rc = lgpu_event_record(s->tail, s->launch))) rc = lgpu_stream_wait_event(s->launch2, s->tail);
  }
  if (rc) { s->failed = 1; return rc; }          // part of the exchange may be enqueued: this rank can no longer stay in step
  for (int i = 0; i < n; i++) s->feed_of[(s->fed + i) % kStepRing] = s->feeds;
  s->fed += n;
  s->feeds++;
  return LGPU_OK;
}

int lgpu_stepper_create(void *comm, int root, int rank, void *launch_stream, const int32_t first_values[4], lgpu_stepper **out) {
  if (!out || root < 0 || rank < 0) { lgpu::set_error("lgpu_stepper_create: bad arguments"); return LGPU_E_BADARG; }
  if (rank == root && !first_values) { lgpu::set_error("lgpu_stepper_create: the root needs the first block"); return LGPU_E_BADARG; }
  lgpu_stepper *s = new lgpu_stepper();
  memset(s, 0, sizeof *s);
  s->comm = comm; s->root = root; s->rank = rank; s->launch = launch_stream; s->waited = -1; s->waited2 = -1;
  int rc = LGPU_OK;
  void *p = nullptr;
  if ((rc = lgpu_malloc(&p, kStepRing * 4 * sizeof(int32_t)))) { delete s; return rc; }
  s->blk = (int32_t *)p;
  if (comm) {
    if (!rc) rc = lgpu_stream_create(&s->side, 1);
    for (int i = 0; i < kStepRing && !rc; i++) rc = lgpu_event_create(&s->ev[i]);
    if (!rc) rc = lgpu_event_create(&s->tail);
  }
  if (!rc) rc = lgpu_stepper_feed(s, first_values, 1);
  if (rc) { lgpu_stepper_destroy(s); return rc; }
  *out = s;
  return LGPU_OK;
}

// one batch: the chain over `tracks` with the parameter block of this step (params->param_block_d is ignored: the stepper's block is used).  next_values != NULL:
// the block of the following step is fed first (the one-block-ahead form; NULL on every rank when the blocks come through lgpu_stepper_feed or after the last step;
// the values matter on the root only).  Arguments are checked BEFORE anything is enqueued: a call that returns LGPU_E_BADARG has changed nothing and may be repeated.
int lgpu_chain_step(lgpu_stepper *s, const int32_t next_values[4], const lgpu_chain_params *params, const lgpu_chain_track *tracks, int ntracks) {
  if (!s || !params || !tracks || ntracks < 1 || ntracks > LGPU_CHAIN_MAX_TRACKS) { lgpu::set_error("lgpu_chain_step: bad arguments"); return LGPU_E_BADARG; }
  if (s->failed) { lgpu::set_error("lgpu_chain_step: an earlier call failed half way; the stepper is out of step with its peers: destroy it"); return LGPU_E_STATE; }
  // every argument check of lgpu_chain BEFORE anything is fed or enqueued (lgpu_chain_check is the list lgpu_chain itself runs): BADARG then really means "untouched"
  { const int bad = lgpu_chain_check(params, tracks, ntracks); if (bad) return bad; }
  if (s->step >= s->fed) { lgpu::set_error("lgpu_chain_step: no parameter block has been fed for step %ld", s->step); return LGPU_E_BADARG; }
  static const int32_t zero[4] = {0, 0, 0, 0};
  int rc;
  if (next_values && s->fed == s->step + 1 && (rc = lgpu_stepper_feed(s, s->rank == s->root ? next_values : zero, 1))) return rc;
  const long f = s->feed_of[s->step % kStepRing];
  const bool second = s->launch2 && (s->step & 1);
  void *st = second ? s->launch2 : s->launch;
  long &waited = second ? s->waited2 : s->waited;
  if (s->comm && f > waited) {
    if ((rc = lgpu_stream_wait_event(st, s->ev[f % kStepRing]))) { s->failed = 1; return rc; }
    waited = f;
  }
  lgpu_chain_params p = *params;
  p.param_block_d = s->blk + 4 * (s->step % kStepRing);
  rc = lgpu_chain(&p, tracks, ntracks, st);
  if (rc) { s->failed = 1; return rc == LGPU_E_BADARG ? LGPU_E_STATE : rc; }          // the exchange for this step has happened on every rank; this rank's launch has not (never BADARG: that code means "untouched")
  s->step++;
  return LGPU_OK;
}

// Frames of consecutive steps are independent (their own tracks, their own parameter block): with a second launch stream the odd steps go there, and the drain of one
// launch overlaps the ramp-up of the next -- what a one-frame-per-GPU worker is short of (a 4K frame is 12.2 us as a launch of its own, 8.7 us inside an 8-track launch).
// Without a communicator the blocks are written on the first launch stream: the second is then ordered behind it at every feed.  NULL switches the overlap off.
// The caller synchronises BOTH streams (or destroys the stepper) before it reads results.
int lgpu_stepper_overlap(lgpu_stepper *s, void *second_launch_stream) {
  if (!s) { lgpu::set_error("lgpu_stepper_overlap: null stepper"); return LGPU_E_BADARG; }
  if (s->failed) { lgpu::set_error("lgpu_stepper_overlap: the stepper is out of step: destroy it"); return LGPU_E_STATE; }
  int rc = LGPU_OK;
  if (second_launch_stream && !s->tail2 && (rc = lgpu_event_create(&s->tail2))) return rc;
  if (second_launch_stream && !s->tail && (rc = lgpu_event_create(&s->tail))) return rc;
  if (s->launch2 && s->launch2 != second_launch_stream) {
    // the stream being dropped (or replaced) may still run odd steps that read ring slots: the first launch stream -- and the side stream, which rewrites the slots when
    // there is a communicator -- go behind it now, since no later fence will look at it
    if ((rc = lgpu_event_record(s->tail2, s->launch2)) || (rc = lgpu_stream_wait_event(s->launch, s->tail2))) return rc;
    if (s->comm && (rc = lgpu_stream_wait_event(s->side, s->tail2))) return rc;
  }
  if (second_launch_stream) {
    // everything fed so far is visible to the new stream: order it behind the stream the blocks were written / received on
    void *from = s->comm ? s->side : s->launch;
    if ((rc = lgpu_event_record(s->tail2, from)) || (rc = lgpu_stream_wait_event(second_launch_stream, s->tail2))) return rc;
    s->waited2 = s->feeds - 1;
  }
  s->launch2 = second_launch_stream;
  return LGPU_OK;
}

int lgpu_stepper_failed(const lgpu_stepper *s) { return s && s->failed ? 1 : 0; }

// host-side wait with a limit (what a worker calls instead of a bare stream synchronise): polls the streams, never blocks inside the runtime
int lgpu_stepper_wait(lgpu_stepper *s, int timeout_ms) {
  if (!s) { lgpu::set_error("lgpu_stepper_wait: null stepper"); return LGPU_E_BADARG; }
  if (s->failed) { lgpu::set_error("lgpu_stepper_wait: the stepper is out of step: destroy it"); return LGPU_E_STATE; }
  const auto t0 = std::chrono::steady_clock::now();
  // which streams take part is a matter of state, not of pointer values: the first launch stream may be the NULL stream (lgpu_stepper_create accepts it, and
  // hipStreamQuery(NULL) is a valid query of it); the side stream only carries work with a communicator, the second launch stream only while the overlap is on
  void *streams[3] = {s->side, s->launch, s->launch2};
  const bool used[3] = {s->comm != nullptr && s->side != nullptr, true, s->launch2 != nullptr};
  const char *names[3] = {"the parameter exchange (side stream): a peer has not entered the same lgpu_stepper_feed / lgpu_chain_step", "a chain launch (first launch stream)",
                          "a chain launch (second launch stream)"};
  for (unsigned spin = 0;; spin++) {
    int busy = -1;
    for (int i = 0; i < 3 && busy < 0; i++) {
      if (!used[i]) continue;
      const int q = lgpu_stream_query(streams[i]);
      if (q < 0) { s->failed = 1; return q; }
      if (q == 0) busy = i;
    }
    if (busy < 0) return LGPU_OK;
    const long waited = (long)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
    if (timeout_ms > 0 && waited >= timeout_ms) {
      s->failed = 1;
      lgpu::set_error("lgpu_stepper_wait: rank %d (root %d) waited %ld ms for %s; %ld blocks fed in %ld feeds, %ld steps launched", s->rank, s->root, waited, names[busy],
                      s->fed, s->feeds, s->step);
      return LGPU_E_TIMEOUT;
    }
    if (spin < 2000) std::this_thread::yield(); else std::this_thread::sleep_for(std::chrono::microseconds(200));
  }
}

const int32_t *lgpu_stepper_block(const lgpu_stepper *s, int which) { return (s && which >= 0) ? s->blk + 4 * (which % kStepRing) : nullptr; }

int lgpu_stepper_destroy(lgpu_stepper *s) {
  if (!s) return LGPU_OK;
  lgpu_sync(s->launch);
  if (s->launch2) lgpu_sync(s->launch2);
  if (s->side) { lgpu_sync(s->side); lgpu_stream_destroy(s->side); }
  for (int i = 0; i < kStepRing; i++) if (s->ev[i]) lgpu_event_destroy(s->ev[i]);
  if (s->tail) lgpu_event_destroy(s->tail);
  if (s->tail2) lgpu_event_destroy(s->tail2);
  if (s->blk) lgpu_free(s->blk);
  delete s;
  return LGPU_OK;
}

}  // extern "C"
