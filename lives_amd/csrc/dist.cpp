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

}  // extern "C"
