// Multi-GPU exchange of liblspiv_hip.so: one process per GPU, RCCL over xGMI.
//
// The reference is a single process with no communication at all; frame pairs are independent
// (docs/user-guide/velocimetry/index.rst:12-13) and it already cuts the time axis into chunks with a one-frame halo
// (pyorc/velocimetry/ffpiv.py:140).  Sharding is that cut one level up: rank r owns a contiguous block of pairs and the
// ONLY exchange is one all-gather of the packed (4, t, y, x) float32 result block (ensemble mode: one sum all-reduce of
// corr_sum / corr_count).  These entry points are that exchange -- no PyTorch, no MPI:
//   * transport LSPIV_COMM_RCCL: librccl.so is dlopen'ed on first use (it is 570 MB; a 1-GPU process never loads it),
//     ncclCommInitRank on the calling thread's current device, collectives on the caller's stream;
//   * transport LSPIV_COMM_SHM: POSIX shared memory + a process-shared barrier, staged through host memory.  It exists
//     for plumbing tests only -- RCCL refuses two ranks on one GPU and cannot run without a GPU at all, while the
//     sharding logic above it has to be tested in both situations (2 processes on a 1-GPU box, and the CPU-only suite).
// The rendezvous is left to the caller: rank 0 obtains a 128-byte id (lspiv_comm_unique_id) and hands it to the other
// ranks through any side channel (bench.py and pyorc_amd.shard use a file on the node).
#include "../../include/lspiv.h"

#include <dlfcn.h>
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace lspiv_comm_detail {

int comm_fail(int code, const char* fmt, ...);   // sets the library's thread-local message (lspiv_api.hip)

// ---- the few RCCL symbols this library uses, resolved at run time -----------------------------------------------
typedef struct { char internal[128]; } NcclUniqueId;
typedef void* NcclComm;
enum { kNcclFloat32 = 7, kNcclFloat64 = 8, kNcclSum = 0, kNcclMax = 2 };   // ncclDataType_t / ncclRedOp_t values of rccl.h
struct Rccl {
  void* handle = nullptr;
  int (*GetVersion)(int*) = nullptr;
  int (*GetUniqueId)(NcclUniqueId*) = nullptr;
  int (*CommInitRank)(NcclComm*, int, NcclUniqueId, int) = nullptr;
  int (*CommDestroy)(NcclComm) = nullptr;
  int (*CommCount)(NcclComm, int*) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, NcclComm, hipStream_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;
std::atomic<bool> g_rccl_ready{false};

int load_rccl() {
  if (g_rccl_ready.load()) return LSPIV_OK;
  static const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
  void* h = nullptr;
  for (const char* n : names) {
    h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h) return comm_fail(LSPIV_ENODEV, "librccl.so not found (%s)", dlerror());
  Rccl r;
  r.handle = h;
#define LSPIV_SYM(field, name)                                                        \
  r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, name));                      \
  if (!r.field) return comm_fail(LSPIV_EHIP, "librccl.so lacks %s", name)
  LSPIV_SYM(GetVersion, "ncclGetVersion");
  LSPIV_SYM(GetUniqueId, "ncclGetUniqueId");
  LSPIV_SYM(CommInitRank, "ncclCommInitRank");
  LSPIV_SYM(CommDestroy, "ncclCommDestroy");
  LSPIV_SYM(CommCount, "ncclCommCount");
  LSPIV_SYM(AllGather, "ncclAllGather");
  LSPIV_SYM(AllReduce, "ncclAllReduce");
  LSPIV_SYM(GetErrorString, "ncclGetErrorString");
#undef LSPIV_SYM
  g_rccl = r;
  g_rccl_ready.store(true);
  return LSPIV_OK;
}

#define RCCL_TRY(expr)                                                                                   \
  do {                                                                                                   \
    int r_ = (expr);                                                                                     \
    if (r_ != 0) return comm_fail(LSPIV_EHIP, "%s failed: %s", #expr, g_rccl.GetErrorString(r_));        \
  } while (0)
#define HIPC_TRY(expr)                                                                                   \
  do {                                                                                                   \
    hipError_t e_ = (expr);                                                                              \
    if (e_ != hipSuccess) return comm_fail(e_ == hipErrorOutOfMemory ? LSPIV_ENOMEM : LSPIV_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

// ---- shared-memory transport (plumbing tests) --------------------------------------------------------------------
struct ShmCtl {
  std::atomic<int> arrived;
  std::atomic<int> generation;
  std::atomic<int> attached;
};

std::string shm_name(const unsigned char* id, const char* what, int rank) {
  char buf[96];
  snprintf(buf, sizeof(buf), "/lspiv_%02x%02x%02x%02x%02x%02x%02x%02x_%s%d", id[8], id[9], id[10], id[11], id[12], id[13],
           id[14], id[15], what, rank);
  return buf;
}

}  // namespace lspiv_comm_detail

using namespace lspiv_comm_detail;

struct lspiv_comm {
  int rank = 0, world = 1, transport = LSPIV_COMM_RCCL;
  int device = -1;
  NcclComm nccl = nullptr;
  int nccl_ranks = 0;
  // shm transport
  unsigned char id[LSPIV_COMM_ID_BYTES] = {0};
  ShmCtl* ctl = nullptr;
  int my_fd = -1;
  size_t my_cap = 0;
  // staging (host variants over RCCL, device variants over shm)
  void* d_stage = nullptr; size_t d_cap = 0;
  std::vector<char> h_stage;
};

namespace {

int shm_barrier(lspiv_comm* c) {
  ShmCtl* k = c->ctl;
  const int gen = k->generation.load(std::memory_order_acquire);
  if (k->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == c->world) {
    k->arrived.store(0, std::memory_order_relaxed);
    k->generation.store(gen + 1, std::memory_order_release);
    return LSPIV_OK;
  }
  const double limit = getenv("LSPIV_COMM_TIMEOUT_S") ? atof(getenv("LSPIV_COMM_TIMEOUT_S")) : 300.0;
  timespec t0;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (unsigned spins = 0; k->generation.load(std::memory_order_acquire) == gen; ++spins) {
    if (spins > 2000) usleep(50);
    if ((spins & 1023u) == 1023u) {
      timespec t1;
      clock_gettime(CLOCK_MONOTONIC, &t1);
      if ((double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec) > limit)
        return comm_fail(LSPIV_EHIP, "shm barrier timed out after %.0f s (a rank died?)", limit);
    }
  }
  return LSPIV_OK;
}

// publish `bytes` of this rank, then read every rank's contribution: allgather (op < 0) or element-wise reduction
int shm_exchange(lspiv_comm* c, const void* send, void* recv, size_t bytes, int dtype, int op) {
  if (bytes > c->my_cap) {
    if (ftruncate(c->my_fd, (off_t)bytes) != 0) return comm_fail(LSPIV_ENOMEM, "ftruncate(%zu) on the shm slot failed", bytes);
    c->my_cap = bytes;
  }
  if (pwrite(c->my_fd, send, bytes, 0) != (ssize_t)bytes) return comm_fail(LSPIV_EHIP, "shm write failed");
  int rc = shm_barrier(c);
  if (rc) return rc;
  std::vector<char> tmp;
  for (int r = 0; r < c->world; ++r) {
    int fd = c->my_fd;
    if (r != c->rank) {
      fd = shm_open(shm_name(c->id, "slot", r).c_str(), O_RDONLY, 0600);
      if (fd < 0) return comm_fail(LSPIV_EHIP, "shm slot of rank %d is gone", r);
    }
    if (op < 0) {
      if (pread(fd, (char*)recv + (size_t)r * bytes, bytes, 0) != (ssize_t)bytes) { if (r != c->rank) close(fd); return comm_fail(LSPIV_EHIP, "shm read failed"); }
    } else {
      tmp.resize(bytes);
      if (pread(fd, tmp.data(), bytes, 0) != (ssize_t)bytes) { if (r != c->rank) close(fd); return comm_fail(LSPIV_EHIP, "shm read failed"); }
      const size_t n = bytes / (dtype == LSPIV_F64 ? 8 : 4);
      // rank order 0, 1, ... on every rank: all ranks get the same bits
      if (dtype == LSPIV_F64) {
        double* o = (double*)recv; const double* x = (const double*)tmp.data();
        for (size_t i = 0; i < n; ++i) o[i] = r == 0 ? x[i] : (op == LSPIV_COMM_SUM ? o[i] + x[i] : (x[i] > o[i] ? x[i] : o[i]));
      } else {
        float* o = (float*)recv; const float* x = (const float*)tmp.data();
        for (size_t i = 0; i < n; ++i) o[i] = r == 0 ? x[i] : (op == LSPIV_COMM_SUM ? o[i] + x[i] : (x[i] > o[i] ? x[i] : o[i]));
      }
    }
    if (r != c->rank) close(fd);
  }
  return shm_barrier(c);   // nobody overwrites its slot before everybody has read it
}

int ensure_dev(lspiv_comm* c, size_t bytes) {
  if (bytes <= c->d_cap) return LSPIV_OK;
  if (c->d_stage) HIPC_TRY(hipFree(c->d_stage));
  c->d_stage = nullptr; c->d_cap = 0;
  HIPC_TRY(hipMalloc(&c->d_stage, bytes));
  c->d_cap = bytes;
  return LSPIV_OK;
}

size_t dtype_bytes(int dtype) { return dtype == LSPIV_F64 ? 8 : 4; }
int check_reduce_args(int dtype, int op) {
  if (dtype != LSPIV_F32 && dtype != LSPIV_F64) return comm_fail(LSPIV_EINVAL, "collectives take LSPIV_F32 / LSPIV_F64, got dtype %d", dtype);
  if (op != LSPIV_COMM_SUM && op != LSPIV_COMM_MAX) return comm_fail(LSPIV_EINVAL, "op %d not in {LSPIV_COMM_SUM, LSPIV_COMM_MAX}", op);
  return LSPIV_OK;
}

}  // namespace

extern "C" {

int lspiv_comm_unique_id(int transport, void* id) {
  if (!id) return comm_fail(LSPIV_EINVAL, "id is NULL");
  memset(id, 0, LSPIV_COMM_ID_BYTES);
  if (transport == LSPIV_COMM_RCCL) {
    int rc = load_rccl();
    if (rc) return rc;
    NcclUniqueId u;
    RCCL_TRY(g_rccl.GetUniqueId(&u));
    memcpy(id, &u, sizeof(u));
    return LSPIV_OK;
  }
  if (transport == LSPIV_COMM_SHM) {
    int fd = open("/dev/urandom", O_RDONLY);
    if (fd < 0 || read(fd, id, 32) != 32) { if (fd >= 0) close(fd); return comm_fail(LSPIV_EHIP, "cannot read /dev/urandom"); }
    close(fd);
    memcpy(id, "LSPIVSHM", 8);
    return LSPIV_OK;
  }
  return comm_fail(LSPIV_EINVAL, "transport %d not in {LSPIV_COMM_RCCL, LSPIV_COMM_SHM}", transport);
}

int lspiv_comm_init(int rank, int world, const void* id, int transport, lspiv_comm** comm) {
  if (!comm || !id) return comm_fail(LSPIV_EINVAL, "NULL argument");
  if (world < 1 || rank < 0 || rank >= world) return comm_fail(LSPIV_EINVAL, "bad rank/world %d/%d", rank, world);
  lspiv_comm* c = new lspiv_comm();
  c->rank = rank; c->world = world; c->transport = transport;
  memcpy(c->id, id, LSPIV_COMM_ID_BYTES);
  if (transport == LSPIV_COMM_RCCL) {
    int rc = load_rccl();
    if (rc) { delete c; return rc; }
    hipError_t e = hipGetDevice(&c->device);
    if (e != hipSuccess) { delete c; return comm_fail(LSPIV_ENODEV, "no HIP device for RCCL: %s", hipGetErrorString(e)); }
    NcclUniqueId u;
    memcpy(&u, id, sizeof(u));
    int r = g_rccl.CommInitRank(&c->nccl, world, u, rank);
    if (r != 0) { delete c; return comm_fail(LSPIV_EHIP, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, g_rccl.GetErrorString(r)); }
    r = g_rccl.CommCount(c->nccl, &c->nccl_ranks);
    if (r != 0 || c->nccl_ranks != world) {
      const int reported = c->nccl_ranks;   // read before the communicator object goes away
      g_rccl.CommDestroy(c->nccl);
      delete c;
      return comm_fail(LSPIV_EHIP, "RCCL reports %d ranks, expected %d", reported, world);
    }
  } else if (transport == LSPIV_COMM_SHM) {
    if (memcmp(id, "LSPIVSHM", 8) != 0) { delete c; return comm_fail(LSPIV_EINVAL, "id was not made by lspiv_comm_unique_id(LSPIV_COMM_SHM)"); }
    (void)hipGetDevice(&c->device);   // may fail: the shm transport also runs without a GPU (host variants)
    (void)hipGetLastError();
    const std::string ctl = shm_name(c->id, "ctl", 0);
    int fd = shm_open(ctl.c_str(), O_CREAT | O_RDWR, 0600);   // zero-filled on creation = the initial barrier state
    if (fd < 0 || ftruncate(fd, sizeof(ShmCtl)) != 0) { if (fd >= 0) close(fd); delete c; return comm_fail(LSPIV_EHIP, "shm_open(%s) failed", ctl.c_str()); }
    void* m = mmap(nullptr, sizeof(ShmCtl), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) { delete c; return comm_fail(LSPIV_EHIP, "mmap of the shm control block failed"); }
    c->ctl = static_cast<ShmCtl*>(m);
    c->my_fd = shm_open(shm_name(c->id, "slot", rank).c_str(), O_CREAT | O_RDWR, 0600);
    if (c->my_fd < 0) { munmap(m, sizeof(ShmCtl)); delete c; return comm_fail(LSPIV_EHIP, "shm_open of this rank's slot failed"); }
    c->ctl->attached.fetch_add(1);
    int rc = shm_barrier(c);   // every slot exists from here on
    if (rc) { lspiv_comm_destroy(c); return rc; }
  } else {
    delete c;
    return comm_fail(LSPIV_EINVAL, "transport %d not in {LSPIV_COMM_RCCL, LSPIV_COMM_SHM}", transport);
  }
  *comm = c;
  return LSPIV_OK;
}

int lspiv_comm_info(lspiv_comm* c, int* rank, int* world, int* transport, int* backend_ranks) {
  if (!c) return comm_fail(LSPIV_EINVAL, "comm is NULL");
  if (rank) *rank = c->rank;
  if (world) *world = c->world;
  if (transport) *transport = c->transport;
  if (backend_ranks) *backend_ranks = c->transport == LSPIV_COMM_RCCL ? c->nccl_ranks : (c->ctl ? c->ctl->attached.load() : 0);
  return LSPIV_OK;
}

int lspiv_comm_allgather_dev(lspiv_comm* c, const void* d_send, void* d_recv, int64_t count, int dtype, void* stream) {
  if (!c || !d_send || !d_recv || count < 0) return comm_fail(LSPIV_EINVAL, "bad argument");
  if (dtype != LSPIV_F32 && dtype != LSPIV_F64) return comm_fail(LSPIV_EINVAL, "collectives take LSPIV_F32 / LSPIV_F64");
  if (count == 0) return LSPIV_OK;
  hipStream_t s = (hipStream_t)stream;
  if (c->transport == LSPIV_COMM_RCCL) {
    RCCL_TRY(g_rccl.AllGather(d_send, d_recv, (size_t)count, dtype == LSPIV_F64 ? kNcclFloat64 : kNcclFloat32, c->nccl, s));
    return LSPIV_OK;
  }
  const size_t bytes = (size_t)count * dtype_bytes(dtype);
  c->h_stage.resize(bytes * (size_t)(c->world + 1));
  HIPC_TRY(hipMemcpyAsync(c->h_stage.data(), d_send, bytes, hipMemcpyDeviceToHost, s));
  HIPC_TRY(hipStreamSynchronize(s));
  int rc = shm_exchange(c, c->h_stage.data(), c->h_stage.data() + bytes, bytes, dtype, -1);
  if (rc) return rc;
  HIPC_TRY(hipMemcpyAsync(d_recv, c->h_stage.data() + bytes, bytes * c->world, hipMemcpyHostToDevice, s));
  HIPC_TRY(hipStreamSynchronize(s));
  return LSPIV_OK;
}

int lspiv_comm_allreduce_dev(lspiv_comm* c, const void* d_send, void* d_recv, int64_t count, int dtype, int op, void* stream) {
  if (!c || !d_send || !d_recv || count < 0) return comm_fail(LSPIV_EINVAL, "bad argument");
  int rc = check_reduce_args(dtype, op);
  if (rc) return rc;
  if (count == 0) return LSPIV_OK;
  hipStream_t s = (hipStream_t)stream;
  if (c->transport == LSPIV_COMM_RCCL) {
    RCCL_TRY(g_rccl.AllReduce(d_send, d_recv, (size_t)count, dtype == LSPIV_F64 ? kNcclFloat64 : kNcclFloat32,
                              op == LSPIV_COMM_SUM ? kNcclSum : kNcclMax, c->nccl, s));
    return LSPIV_OK;
  }
  const size_t bytes = (size_t)count * dtype_bytes(dtype);
  c->h_stage.resize(2 * bytes);
  HIPC_TRY(hipMemcpyAsync(c->h_stage.data(), d_send, bytes, hipMemcpyDeviceToHost, s));
  HIPC_TRY(hipStreamSynchronize(s));
  rc = shm_exchange(c, c->h_stage.data(), c->h_stage.data() + bytes, bytes, dtype, op);
  if (rc) return rc;
  HIPC_TRY(hipMemcpyAsync(d_recv, c->h_stage.data() + bytes, bytes, hipMemcpyHostToDevice, s));
  HIPC_TRY(hipStreamSynchronize(s));
  return LSPIV_OK;
}

int lspiv_comm_allgather(lspiv_comm* c, const void* send, void* recv, int64_t count, int dtype) {
  if (!c || !send || !recv || count < 0) return comm_fail(LSPIV_EINVAL, "bad argument");
  if (dtype != LSPIV_F32 && dtype != LSPIV_F64) return comm_fail(LSPIV_EINVAL, "collectives take LSPIV_F32 / LSPIV_F64");
  if (count == 0) return LSPIV_OK;
  const size_t bytes = (size_t)count * dtype_bytes(dtype);
  if (c->transport == LSPIV_COMM_SHM) return shm_exchange(c, send, recv, bytes, dtype, -1);
  int rc = ensure_dev(c, bytes * (size_t)(c->world + 1));
  if (rc) return rc;
  char* d = (char*)c->d_stage;
  HIPC_TRY(hipMemcpy(d, send, bytes, hipMemcpyHostToDevice));
  rc = lspiv_comm_allgather_dev(c, d, d + bytes, count, dtype, nullptr);
  if (rc) return rc;
  HIPC_TRY(hipStreamSynchronize(nullptr));
  HIPC_TRY(hipMemcpy(recv, d + bytes, bytes * c->world, hipMemcpyDeviceToHost));
  return LSPIV_OK;
}

int lspiv_comm_allreduce(lspiv_comm* c, const void* send, void* recv, int64_t count, int dtype, int op) {
  if (!c || !send || !recv || count < 0) return comm_fail(LSPIV_EINVAL, "bad argument");
  int rc = check_reduce_args(dtype, op);
  if (rc) return rc;
  if (count == 0) return LSPIV_OK;
  const size_t bytes = (size_t)count * dtype_bytes(dtype);
  if (c->transport == LSPIV_COMM_SHM) return shm_exchange(c, send, recv, bytes, dtype, op);
  rc = ensure_dev(c, 2 * bytes);
  if (rc) return rc;
  char* d = (char*)c->d_stage;
  HIPC_TRY(hipMemcpy(d, send, bytes, hipMemcpyHostToDevice));
  rc = lspiv_comm_allreduce_dev(c, d, d + bytes, count, dtype, op, nullptr);
  if (rc) return rc;
  HIPC_TRY(hipStreamSynchronize(nullptr));
  HIPC_TRY(hipMemcpy(recv, d + bytes, bytes, hipMemcpyDeviceToHost));
  return LSPIV_OK;
}

int lspiv_comm_barrier(lspiv_comm* c) {
  if (!c) return comm_fail(LSPIV_EINVAL, "comm is NULL");
  if (c->transport == LSPIV_COMM_SHM) return shm_barrier(c);
  float one = 1.0f, sum = 0.0f;
  int rc = lspiv_comm_allreduce(c, &one, &sum, 1, LSPIV_F32, LSPIV_COMM_SUM);
  if (rc) return rc;
  if ((int)sum != c->world) return comm_fail(LSPIV_EHIP, "barrier all-reduce returned %g for %d ranks", (double)sum, c->world);
  return LSPIV_OK;
}

int lspiv_comm_destroy(lspiv_comm* c) {
  if (!c) return LSPIV_OK;
  if (c->nccl) g_rccl.CommDestroy(c->nccl);
  if (c->d_stage) hipFree(c->d_stage);
  if (c->ctl) {
    const int left = c->ctl->attached.fetch_sub(1) - 1;
    munmap(c->ctl, sizeof(ShmCtl));
    if (left <= 0) shm_unlink(shm_name(c->id, "ctl", 0).c_str());
  }
  if (c->my_fd >= 0) {
    close(c->my_fd);
    shm_unlink(shm_name(c->id, "slot", c->rank).c_str());
  }
  delete c;
  return LSPIV_OK;
}

}  // extern "C"
