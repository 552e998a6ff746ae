// C ABI of liblspiv_hip.so (see include/lspiv.h for the contract and the reference call sites
// each entry point replaces).  Host logic only: argument checks, window grid, HBM workspaces,
// H2D/D2H staging, kernel dispatch.  No PyTorch, no CPU compute fallback: without a gfx950
// device every compute entry point fails with LSPIV_ENODEV.
#include "../../include/lspiv.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "common.h"
#include "project_tile.h"
#include "project_fused.h"
#include "host_stage.h"

namespace lspiv {   // project.hip
hipError_t launch_project_u8(const uint8_t* frames, int64_t src_elems, int n_frames, const int* qlo1, const int* qlo2,
                             const uint32_t* qdesc, const int* nn_src, uint8_t* out, int n_out, hipStream_t s);
}

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      int code_ = (e_ == hipErrorOutOfMemory) ? LSPIV_ENOMEM                                   \
                  : (e_ == hipErrorNoDevice || e_ == hipErrorNoBinaryForGpu) ? LSPIV_ENODEV    \
                                                                             : LSPIV_EHIP;     \
      return fail(code_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    }                                                                                          \
  } while (0)

size_t elem_size(int dtype) { return dtype == LSPIV_U8 ? 1 : dtype == LSPIV_F32 ? 4 : 8; }

struct Grid {
  int64_t n_rows = 0, n_cols = 0;
};

int check_window(int wy, int wx, int oy, int ox) {
  if (wy < 2 || wx < 2 || wy > LSPIV_MAX_WINDOW || wx > LSPIV_MAX_WINDOW)
    return fail(LSPIV_EUNSUPPORTED, "window %dx%d outside supported range 2..%d", wy, wx, LSPIV_MAX_WINDOW);
  if (oy < 0 || ox < 0 || oy >= wy || ox >= wx)
    return fail(LSPIV_EINVAL, "overlap (%d,%d) must satisfy 0 <= overlap < window (%d,%d)", oy, ox, wy, wx);
  return LSPIV_OK;
}

// ffpiv.window.get_axis_shape restated: (dim - win)//(win - overlap) + 1
int64_t axis_shape(int64_t dim, int win, int ov) { return dim < win ? 0 : (dim - win) / (win - ov) + 1; }

int make_grid(int64_t H, int64_t W, int wy, int wx, int oy, int ox, Grid* g) {
  int rc = check_window(wy, wx, oy, ox);
  if (rc) return rc;
  if (H <= 0 || W <= 0) return fail(LSPIV_ESHAPE, "frame shape (%lld,%lld) invalid", (long long)H, (long long)W);
  g->n_rows = axis_shape(H, wy, oy);
  g->n_cols = axis_shape(W, wx, ox);
  if (g->n_rows <= 0 || g->n_cols <= 0)
    return fail(LSPIV_ESHAPE, "frame (%lld,%lld) smaller than window (%d,%d)", (long long)H, (long long)W, wy, wx);
  return LSPIV_OK;
}

// ---- per-device context: launch stream + grow-only HBM / pinned workspaces -------------------
struct DeviceCtx {
  hipStream_t stream = nullptr;
  hipStream_t copy_stream = nullptr;
  void* d_frames = nullptr;  size_t frames_cap = 0;
  float* d_out = nullptr;    size_t out_cap = 0;
  float* d_planes = nullptr; size_t planes_cap = 0;
  void* d_scratch = nullptr; size_t scratch_cap = 0;   // temporaries of *_dev entry points (normalize)
  void* d_dft = nullptr; size_t dft_cap = 0;           // plane slots of the windows above 128 px (grow-only)
  uint8_t* d_keep = nullptr; size_t keep_cap = 0;      // per-window flags of the "stack" signal mode
  // float64 rescue pass: one set of lists per launch stream (a stream orders its own PIV kernel -> rescue kernel pairs;
  // two streams must not share counters), grow-only
  struct RescueWs { hipStream_t stream; void* base; size_t cap_bytes; uint32_t cap_fit, cap_amb; };
  std::vector<RescueWs> rescue;
  void* pinned[2] = {nullptr, nullptr}; size_t pinned_cap = 0;  // H2D staging ring
  hipEvent_t staged[2] = {nullptr, nullptr};
  // host-pointer projection entry points (round 6): kProjSlots independent sets of {stream, input buffer, output buffer}, each under
  // its own lock (DeviceLocks::project), none of them shared with the PIV host entry points -- a project_hip block that dask runs
  // on a worker thread neither waits for the `host` lock a PIV call holds for its whole upload + kernels + download, nor for the
  // other block in flight: block k + 1 crosses PCIe while block k's kernel runs and its result goes back
  // pin[2]: pinned bounce buffers of kProjPinBytes each -- the frames of a block go up and its result comes down in slices through
  // them (staging threads on one slice, DMA on the other): a pageable hipMemcpyAsync moved a block's 93 MB of float32 result at a
  // few GB/s, which WAS the time of the generic project_hip -> get_piv path (bench.py: dropin_generic_path)
  struct ProjWs { hipStream_t stream = nullptr; void* d_in = nullptr; size_t in_cap = 0; void* d_out = nullptr; size_t out_cap = 0;
                  void* pin[2] = {nullptr, nullptr}; hipEvent_t ev[2] = {nullptr, nullptr}; };
  static constexpr size_t kProjPinBytes = (size_t)16 << 20;
  static constexpr int kProjSlots = 2;
  ProjWs proj[kProjSlots];
  bool arch_ok = false;
};
std::mutex g_mu;        // the table of contexts itself (created lazily); never held while another lock is taken
std::vector<DeviceCtx*> g_ctx;
// Locks are PER DEVICE ("one host thread (or process) per GPU"): a process that drives several GPUs from several
// threads serialises only the calls that share a device's workspaces, not all of them (round 4 had three process-wide mutexes:
// every launch of every device queued behind one lock).  A fixed table, so that an entry point can take its lock before a context
// exists (and on a machine without a device: slot 0).
//   host:     host-pointer entry points share one set of workspaces (upload buffer, result buffer, pinned ring) per device
//   dispatch: the PIV kernel and the rescue kernels of ONE launch share their stream's lists and counters (see dispatch())
//   lists:    the per-stream rescue lists of a context (DeviceCtx::rescue)
//   project:  one per projection slot (DeviceCtx::proj): the host-pointer projection entry points; never nested with the others
// Order when nested: host -> dispatch -> lists.
constexpr int kMaxDevices = 64;
struct DeviceLocks { std::mutex host, dispatch, lists, project[DeviceCtx::kProjSlots]; std::atomic<unsigned> next_project{0}; };
DeviceLocks g_locks[kMaxDevices];
static int current_device_slot() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
  return dev >= 0 && dev < kMaxDevices ? dev : 0;
}
static DeviceLocks& locks_here() { return g_locks[current_device_slot()]; }

int get_ctx(DeviceCtx** out) {
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0) return fail(LSPIV_ENODEV, "no HIP device visible (%s)", hipGetErrorString(e));
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_mu);
  if ((int)g_ctx.size() < ndev) g_ctx.resize(ndev, nullptr);
  if (!g_ctx[dev]) {
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
      return fail(LSPIV_ENODEV, "device %d is %s; this library carries gfx950 (MI355X) code only", dev, prop.gcnArchName);
    DeviceCtx* c = new DeviceCtx();
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    c->arch_ok = true;
    g_ctx[dev] = c;
  }
  *out = g_ctx[dev];
  return LSPIV_OK;
}

// A projection slot of the current device, locked: the first free one, else the next in turn (callers queue fairly on two locks).
struct ProjSlot {
  std::unique_lock<std::mutex> lk;
  DeviceCtx::ProjWs* ws = nullptr;
};
static int take_proj_slot(DeviceCtx* c, ProjSlot* out) {
  DeviceLocks& l = locks_here();
  int k = -1;
  for (int i = 0; i < DeviceCtx::kProjSlots && k < 0; ++i) {
    std::unique_lock<std::mutex> t(l.project[i], std::try_to_lock);
    if (t.owns_lock()) { out->lk = std::move(t); k = i; }
  }
  if (k < 0) {
    k = (int)(l.next_project.fetch_add(1) % DeviceCtx::kProjSlots);
    out->lk = std::unique_lock<std::mutex>(l.project[k]);
  }
  out->ws = &c->proj[k];
  if (!out->ws->stream) HIP_TRY(hipStreamCreateWithFlags(&out->ws->stream, hipStreamNonBlocking));
  return LSPIV_OK;
}

// "trace" (tests / measurement only, off by default): HIP events the LIBRARY records around the spans below, on the streams the work
// runs on; lspiv_trace_read returns them in milliseconds since lspiv_trace(1).  What a wall clock cannot show -- that a projection
// block's kernel ran INSIDE a concurrent PIV host call on the same device -- two event pairs can.
struct TraceRec { int kind; hipEvent_t e0, e1; };
struct Trace { std::mutex mu; std::atomic<bool> on{false}; hipEvent_t base = nullptr; std::vector<TraceRec> recs; };
Trace g_trace[kMaxDevices];
struct TraceSpan { Trace* t = nullptr; TraceRec r{}; };
static void trace_begin(TraceSpan* sp, int kind, hipStream_t s) {
  Trace& t = g_trace[current_device_slot()];
  if (!t.on.load()) return;
  sp->r.kind = kind;
  if (hipEventCreate(&sp->r.e0) != hipSuccess || hipEventCreate(&sp->r.e1) != hipSuccess) { (void)hipGetLastError(); return; }
  if (hipEventRecord(sp->r.e0, s) != hipSuccess) { (void)hipGetLastError(); return; }
  sp->t = &t;
}
static void trace_end(TraceSpan* sp, hipStream_t s) {
  if (!sp->t) return;
  if (hipEventRecord(sp->r.e1, s) != hipSuccess) { (void)hipGetLastError(); return; }
  std::lock_guard<std::mutex> lk(sp->t->mu);
  sp->t->recs.push_back(sp->r);
  sp->t = nullptr;
}

// host memory registered with HIP (hipHostMalloc / lspiv_host_alloc / hipHostRegister) can be DMA'd in place
static bool is_pinned(const void* p) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
  return a.type == hipMemoryTypeHost;
}

// the staging copies run on persistent host threads (host_stage.cpp: LSPIV_STAGE_THREADS, AVX2 non-temporal stores); float64
// frames (what pyorc's project_numpy hands over) are narrowed to float32 while they are staged: the kernels
// convert every sample to float32 first thing anyway (same IEEE round-to-nearest conversion on both sides, so the results are
// bit-identical), and the PCIe transfer -- the bound of the host entry points -- halves
using lspiv_host::staged_copy;
static void staged_narrow(float* dst, const double* src, size_t n) { lspiv_host::staged_narrow(dst, src, n, 1, nullptr); }

// two pinned staging slots of >= one frame each (LSPIV_STAGE_BYTES per slot, default 32 MiB)
int stage_ring(DeviceCtx* c, size_t frame_bytes) {
  const size_t want = getenv("LSPIV_STAGE_BYTES") ? (size_t)atoll(getenv("LSPIV_STAGE_BYTES")) : ((size_t)32 << 20);
  const size_t need = std::max(want, frame_bytes);
  if (c->pinned_cap >= need && c->pinned_cap < 2 * need + frame_bytes) return LSPIV_OK;
  for (int i = 0; i < 2; ++i) {
    if (c->pinned[i]) HIP_TRY(hipHostFree(c->pinned[i]));
    c->pinned[i] = nullptr;
  }
  c->pinned_cap = 0;
  for (int i = 0; i < 2; ++i) {
    HIP_TRY(hipHostMalloc(&c->pinned[i], need, hipHostMallocDefault));
    if (!c->staged[i]) HIP_TRY(hipEventCreateWithFlags(&c->staged[i], hipEventDisableTiming));
  }
  c->pinned_cap = need;
  return LSPIV_OK;
}

template <typename P>
int ensure(P** ptr, size_t* cap, size_t bytes) {
  if (bytes <= *cap) return LSPIV_OK;
  if (*ptr) HIP_TRY(hipFree(*ptr));
  *ptr = nullptr; *cap = 0;
  void* p = nullptr;
  HIP_TRY(hipMalloc(&p, bytes));
  *ptr = static_cast<P*>(p); *cap = bytes;
  return LSPIV_OK;
}

// Whole-stack upload through the pinned ring (staging threads + DMA overlapped, float64 narrowed to float32): the
// consumer stream waits for every slot's DMA.  *dev_dtype is the sample type the device copy ends up with.
int upload_stack(DeviceCtx* c, const void* frames, int dtype, int64_t T, int64_t frame_elems, int* dev_dtype) {
  *dev_dtype = dtype == LSPIV_F64 ? LSPIV_F32 : dtype;
  const size_t frame_bytes = (size_t)frame_elems * elem_size(*dev_dtype), src_frame_bytes = (size_t)frame_elems * elem_size(dtype);
  int rc = ensure(&c->d_frames, &c->frames_cap, (size_t)T * frame_bytes);
  if (rc) return rc;
  rc = stage_ring(c, frame_bytes);
  if (rc) return rc;
  const int64_t fpb = std::max<int64_t>(1, (int64_t)(c->pinned_cap / frame_bytes));
  int batch = 0;
  for (int64_t f0 = 0; f0 < T; ++batch) {
    const int64_t f1 = std::min<int64_t>(T, f0 + fpb);
    const int slot = batch & 1;
    if (batch >= 2) HIP_TRY(hipEventSynchronize(c->staged[slot]));  // the slot's previous DMA has drained
    const size_t nb = (size_t)(f1 - f0) * frame_bytes;
    if (dtype == LSPIV_F64)
      staged_narrow((float*)c->pinned[slot], (const double*)((const char*)frames + (size_t)f0 * src_frame_bytes), (size_t)(f1 - f0) * frame_elems);
    else
      staged_copy(c->pinned[slot], (const char*)frames + (size_t)f0 * src_frame_bytes, nb);
    HIP_TRY(hipMemcpyAsync((char*)c->d_frames + (size_t)f0 * frame_bytes, c->pinned[slot], nb, hipMemcpyHostToDevice, c->copy_stream));
    HIP_TRY(hipEventRecord(c->staged[slot], c->copy_stream));
    HIP_TRY(hipStreamWaitEvent(c->stream, c->staged[slot], 0));
    f0 = f1;
  }
  return LSPIV_OK;
}

// the engine semantics that could not be pinned on a real ffpiv run: run-time options,
// defaults = the oracle's reading; LSPIV_BORDER_PEAK / LSPIV_SIGNAL_MODE / LSPIV_SIGNAL_POSITIVE preset them
int env_opt(const char* name, int lo, int hi) {
  const char* e = getenv(name);
  const int v = e ? atoi(e) : 0;
  return v < lo || v > hi ? 0 : v;
}
std::atomic<int> g_opt_border{env_opt("LSPIV_BORDER_PEAK", 0, 2)};        // 0 NaN, 1 plane centre, 2 integer peak
std::atomic<int> g_opt_signal_mode{env_opt("LSPIV_SIGNAL_MODE", 0, 1)};   // 0 per window pair, 1 per window position over the chunk
std::atomic<int> g_opt_signal_pos{env_opt("LSPIV_SIGNAL_POSITIVE", 0, 1)}; // 0 samples != 0, 1 samples > 0

// the readings of ffpiv added in round 3 (same names and values as oracle/'s SEMANTICS)
std::atomic<int> g_opt_v_sign{env_opt("LSPIV_V_SIGN", 0, 1)};      // 0 v as it comes out of the plane, 1 negated
std::atomic<int> g_opt_norm_clip{getenv("LSPIV_NORM_CLIP") && atoi(getenv("LSPIV_NORM_CLIP")) == 0 ? 0 : 1};   // 1 negative lobes of the normalised window removed (A3), 0 kept
std::atomic<int> g_opt_std_ddof{env_opt("LSPIV_STD_DDOF", 0, 1)};    // 0 population, 1 sample standard deviation
std::atomic<int> g_opt_round_odd{env_opt("LSPIV_ROUND_ODD", 0, 2)};   // round_to_even of odd sizes: 0 half-even of x / 2, 1 up, 2 down (host side; kept here so that ONE place holds every switch)

// float64 rescue pass (piv_rescue.hip): on by default; LSPIV_RESCUE=0 / lspiv_set_option("rescue", 0) keeps the float32 results.
// rescue_kappa: the plane noise the flags assume, in 1e-9 of the plane maximum (measured worst case 2.7e-7 in the units of the
// flag's error model -- tools/calib_rescue.py; default 500 = 5e-7); rescue_tau: relative arg-max gap, in 1e-9, below which the
// whole plane is re-evaluated (float32 noise between two samples is <= 8.4e-7; default 4000 = 4e-6)
static int env_opt_def(const char* name, int lo, int hi, int def) {
  const char* e = getenv(name);
  if (!e) return def;
  const int v = atoi(e);
  return v < lo || v > hi ? def : v;
}
std::atomic<int> g_opt_rescue{env_opt_def("LSPIV_RESCUE", 0, 1, 1)};
// float64 host stacks: a frame whose DC offset (host_stage.h, frame_offset) reaches this magnitude has it taken off while it is
// narrowed to float32 -- PIV entry points only (the per-window normalisation does not see it), and only without a signal threshold
// (which counts samples != 0).  -1: never.
std::atomic<int> g_opt_narrow_offset{env_opt_def("LSPIV_NARROW_OFFSET", -1, 1 << 30, 1024)};
static std::vector<double> narrow_offsets(const double* frames, size_t frame_elems, int64_t n_frames, float signal_threshold) {
  std::vector<double> off((size_t)n_frames, 0.0);
  const int min_abs = g_opt_narrow_offset.load();
  if (min_abs < 0 || signal_threshold >= 0.0f) return off;
  for (int64_t f = 0; f < n_frames; ++f) off[(size_t)f] = lspiv_host::frame_offset(frames + (size_t)f * frame_elems, frame_elems, (double)min_abs);
  return off;
}
std::atomic<int> g_opt_rescue_kappa{env_opt_def("LSPIV_RESCUE_KAPPA", 0, 1000000, 500)};
std::atomic<int> g_opt_rescue_tau{env_opt_def("LSPIV_RESCUE_TAU", 0, 1000000, 4000)};

// the rescue lists of stream `s`, large enough for a launch of n_tiles windows (a quarter of them "fit", a sixteenth "amb":
// beyond that the excess keeps its float32 result -- imagery THAT sparse has no usable peaks)
int rescue_ws(DeviceCtx* c, hipStream_t s, uint32_t n_tiles, lspiv::PivParams* p) {
  std::lock_guard<std::mutex> lk(locks_here().lists);
  DeviceCtx::RescueWs* ws = nullptr;
  for (auto& w : c->rescue) if (w.stream == s) ws = &w;
  if (!ws) { c->rescue.push_back({s, nullptr, 0, 0, 0}); ws = &c->rescue.back(); }
  const uint32_t cap_fit = std::max<uint32_t>(4096u, n_tiles / 4u), cap_amb = std::max<uint32_t>(1024u, n_tiles / 16u);
  const size_t hdr_bytes = 256;   // sizeof(RescueHdr) rounded up, keeps the lists 256-byte aligned
  static_assert(sizeof(lspiv::RescueHdr) <= 256, "header slot too small");
  if (cap_fit > ws->cap_fit || cap_amb > ws->cap_amb || !ws->base) {
    const uint32_t nf = std::max(cap_fit, ws->cap_fit), na = std::max(cap_amb, ws->cap_amb);
    const size_t bytes = hdr_bytes + (size_t)nf * sizeof(uint4) + (size_t)na * sizeof(uint32_t);
    void* grown = nullptr;
    HIP_TRY(hipMalloc(&grown, bytes));
    if (ws->base) {
      // the header moves along: its per-pass counters are zero between passes (the rescue kernel's last block resets them) and
      // the totals of lspiv_rescue_stats ("summed over all launches") survive the regrow
      hipError_t e = hipStreamSynchronize(s);
      if (e == hipSuccess) e = hipMemcpy(grown, ws->base, hdr_bytes, hipMemcpyDeviceToDevice);
      if (e != hipSuccess) { (void)hipFree(grown); return fail(LSPIV_EHIP, "rescue lists: %s", hipGetErrorString(e)); }
      HIP_TRY(hipFree(ws->base));
    } else {
      HIP_TRY(hipMemsetAsync(grown, 0, hdr_bytes, s));   // ordered before the kernels of this stream
    }
    ws->base = grown;
    ws->cap_bytes = bytes; ws->cap_fit = nf; ws->cap_amb = na;
  }
  p->rescue_hdr = static_cast<lspiv::RescueHdr*>(ws->base);
  p->rescue_fit = reinterpret_cast<uint4*>((char*)ws->base + hdr_bytes);
  p->rescue_amb = reinterpret_cast<uint32_t*>((char*)ws->base + hdr_bytes + (size_t)ws->cap_fit * sizeof(uint4));
  p->rescue_cap_fit = ws->cap_fit;
  p->rescue_cap_amb = ws->cap_amb;
  return LSPIV_OK;
}

int fill_params(lspiv::PivParams* p, const void* d_frames, int dtype, int64_t T, int64_t H, int64_t W, int wy, int wx,
                int oy, int ox, float signal_threshold, const Grid& g) {
  if (dtype < 0 || dtype > 2) return fail(LSPIV_EINVAL, "dtype %d not in {0:u8, 1:f32, 2:f64}", dtype);
  if (T < 2) return fail(LSPIV_ESHAPE, "need at least 2 frames, got %lld", (long long)T);
  const int64_t n_win = g.n_rows * g.n_cols;
  const int64_t n_tiles = (T - 1) * n_win;
  if (n_tiles >= (int64_t)1 << 31 || H * W >= (int64_t)1 << 31)
    return fail(LSPIV_EINVAL, "chunk too large for one launch: %lld windows (limit 2^31); use smaller chunks",
                (long long)n_tiles);
  memset(p, 0, sizeof(*p));
  p->frames = d_frames;
  p->frame_elems = H * W;
  p->H = (int)H; p->W = (int)W;
  p->wy = wy; p->wx = wx;
  p->sy = wy - oy; p->sx = wx - ox;
  p->n_rows = (int)g.n_rows; p->n_cols = (int)g.n_cols;
  p->n_win = (uint32_t)n_win;
  p->n_tiles = (uint32_t)n_tiles;
  p->n_pairs = (uint32_t)(T - 1);
  p->signal_threshold = signal_threshold;
  p->border_mode = g_opt_border.load();
  p->nz_positive = g_opt_signal_pos.load();
  {
    const double n = (double)wy * (double)wx;
    const double g2 = g_opt_std_ddof.load() && n > 1.0 ? (n - 1.0) / n : 1.0;   // sample std: every normalised window shrinks by sqrt((n-1)/n)
    p->std_gain2 = (float)g2;
    p->std_gain = (float)std::sqrt(g2);
    p->norm_clip = g_opt_norm_clip.load();
  }
  // flag model of the rescue pass (common.h, peak_cond): k = 2 kappa / (ln 2 * 1e-4)
  p->rescue_k = (float)(2.0 * g_opt_rescue_kappa.load() * 1e-9 / (0.6931471805599453 * 1e-4));
  p->rescue_tau = (float)(g_opt_rescue_tau.load() * 1e-9);
  p->div_ncols = lspiv::FastDiv::make((uint32_t)g.n_cols);
  p->div_jobs = lspiv::FastDiv::make((uint32_t)((n_win + 1) / 2));
  p->div_nwin = lspiv::FastDiv::make((uint32_t)n_win);
  return LSPIV_OK;
}

int dispatch_kernels(const lspiv::PivParams& p, int dtype, bool ensemble, hipStream_t s);

// PIV kernel of the window's family, then -- per-timestep mode, unless switched off -- the float64 rescue pass over the
// windows that kernel flagged (piv_rescue.hip)
// The PIV kernel and the two rescue kernels of ONE launch share the stream's lists and counters: they are issued under one
// lock, so that two host threads launching on the same stream (NULL -> the library's stream) cannot interleave as PIV 1,
// PIV 2, rescue 1 -- rescue 1 would then consume launch 2's records with launch 1's parameters.  Launches are asynchronous:
// the lock is held for microseconds.  (The rescue kernels skip a record whose index is outside their launch all the same.)
// "time_kernel" option (measurement only, off by default): HIP events on the launch's own stream right before and right after the
// PIV kernel(s) of a launch -- NOT around the rescue kernels that follow --, i.e. the duration rocprofv3 --kernel-trace reports for
// the dominant kernel, inside the launch exactly as a caller issues it (rescue pass on: the kernel appends its records).  A ring of
// the last 16 launches per device; lspiv_kernel_times reads and empties it.  bench.py's roofline.achieved comes from here.
std::atomic<int> g_opt_time_kernel{0};
constexpr int kTimerRing = 16;
struct KernelTimer { hipEvent_t e0[kTimerRing] = {}, e1[kTimerRing] = {}; int n = 0; };
KernelTimer g_timers[kMaxDevices];   // guarded by the device's dispatch lock
static int timed_dispatch_kernels(const lspiv::PivParams& p, int dtype, bool ensemble, hipStream_t s) {
  if (!g_opt_time_kernel.load()) return dispatch_kernels(p, dtype, ensemble, s);
  KernelTimer& t = g_timers[current_device_slot()];
  const int k = t.n % kTimerRing;
  if (!t.e0[k]) { HIP_TRY(hipEventCreate(&t.e0[k])); HIP_TRY(hipEventCreate(&t.e1[k])); }
  HIP_TRY(hipEventRecord(t.e0[k], s));
  const int rc = dispatch_kernels(p, dtype, ensemble, s);
  if (rc) return rc;
  HIP_TRY(hipEventRecord(t.e1[k], s));
  ++t.n;
  return LSPIV_OK;
}
int dispatch(const lspiv::PivParams& p0, int dtype, bool ensemble, hipStream_t s) {
  if (ensemble || !g_opt_rescue.load()) {
    if (!g_opt_time_kernel.load()) return dispatch_kernels(p0, dtype, ensemble, s);
    std::lock_guard<std::mutex> launch_lock(locks_here().dispatch);
    return timed_dispatch_kernels(p0, dtype, ensemble, s);
  }
  DeviceCtx* c;
  int rc = get_ctx(&c);
  if (rc) return rc;
  lspiv::PivParams p = p0;
  std::lock_guard<std::mutex> launch_lock(locks_here().dispatch);
  rc = rescue_ws(c, s, p.n_tiles, &p);
  if (rc) return rc;
  rc = timed_dispatch_kernels(p, dtype, false, s);
  if (rc) return rc;
  const hipError_t e = lspiv::launch_piv_rescue(p, dtype, s);
  if (e != hipSuccess) return fail(LSPIV_EHIP, "rescue kernel launch failed: %s", hipGetErrorString(e));
  return LSPIV_OK;
}

// "v_sign" option: the engine's v negated (after the kernels and the rescue pass; the default costs nothing)
int apply_v_sign(float* d_v, int64_t n, hipStream_t s) {
  if (!g_opt_v_sign.load() || !d_v) return LSPIV_OK;
  const hipError_t e = lspiv::launch_negate(d_v, n, s);
  if (e != hipSuccess) return fail(LSPIV_EHIP, "kernel launch failed: %s", hipGetErrorString(e));
  return LSPIV_OK;
}

int dispatch_kernels(const lspiv::PivParams& p, int dtype, bool ensemble, hipStream_t s) {
  const int kind = lspiv_kernel_kind(p.wy, p.wx);
  hipError_t e;
  switch (kind) {
    case 7: e = lspiv::launch_piv_embed16(p, dtype, ensemble, s); break;
    case 4: e = lspiv::launch_piv_embed32(p, dtype, ensemble, s); break;
    case 5: e = lspiv::launch_piv_embed64(p, dtype, ensemble, s); break;
    case 6: e = p.wy == 8 ? lspiv::launch_piv_fft8(p, dtype, ensemble, s) : lspiv::launch_piv_fft16(p, dtype, ensemble, s); break;
    case 8:
      switch (p.wy) {
#define LSPIV_PFA_CASE(n) case n: e = lspiv::launch_piv_fft##n(p, dtype, ensemble, s); break;
        LSPIV_PFA_SIZES(LSPIV_PFA_CASE)
#undef LSPIV_PFA_CASE
        default: return fail(LSPIV_EUNSUPPORTED, "no prime-factor kernel for window %d", p.wy);
      }
      break;
    case 1: e = lspiv::launch_piv_fft32(p, dtype, ensemble, s); break;
    case 2: e = lspiv::launch_piv_fft64(p, dtype, ensemble, s); break;
    case 3: e = lspiv::launch_piv_direct(p, dtype, ensemble, s); break;
    case 9: e = lspiv::launch_piv_dft(p, dtype, ensemble, s); break;
    case 10: {   // above 128 px: the same passes on slots of HBM scratch owned by the context (one stream at a time, like the
                 // other context workspaces)
      DeviceCtx* c;
      int rc = get_ctx(&c);
      if (rc) return rc;
      const size_t slot = lspiv::piv_dft_global_slot_floats(p.wy, p.wx);
      const int blocks = lspiv::piv_dft_global_blocks(ensemble ? p.n_win : p.n_tiles);
      rc = ensure(&c->d_dft, &c->dft_cap, slot * (size_t)blocks * sizeof(float));
      if (rc) return rc;
      lspiv::PivParams q = p;
      q.dft_scratch = (float*)c->d_dft;
      q.dft_slot = slot;
      e = lspiv::launch_piv_dft_global(q, dtype, ensemble, s);
      break;
    }
    default: return fail(LSPIV_EUNSUPPORTED, "no kernel for window %dx%d", p.wy, p.wx);
  }
  if (e != hipSuccess) return fail(LSPIV_EHIP, "kernel launch failed: %s", hipGetErrorString(e));
  return LSPIV_OK;
}

// Host-pointer projection (what a project_hip dask block calls): a projection slot's own stream and buffers under the slot's own lock
// (round 6) -- not the PIV host entry points' workspaces and `host` lock, which a concurrent lspiv_piv_pairs holds from its first
// upload to its last download.
template <typename Launch>
static int project_host(size_t ib, size_t ob, const void* frames, void* out, Launch&& launch) {
  DeviceCtx* c;
  int rc = get_ctx(&c);
  if (rc) return rc;
  ProjSlot slot;
  rc = take_proj_slot(c, &slot);
  if (rc) return rc;
  DeviceCtx::ProjWs* w = slot.ws;
  rc = ensure(&w->d_in, &w->in_cap, ib);
  if (rc) return rc;
  rc = ensure(&w->d_out, &w->out_cap, ob);
  if (rc) return rc;
  for (int k = 0; k < 2; ++k) {
    if (!w->pin[k]) HIP_TRY(hipHostMalloc(&w->pin[k], DeviceCtx::kProjPinBytes, hipHostMallocDefault));
    if (!w->ev[k]) HIP_TRY(hipEventCreateWithFlags(&w->ev[k], hipEventDisableTiming));
  }
  const size_t slice = DeviceCtx::kProjPinBytes;
  const bool in_pinned = is_pinned(frames), out_pinned = is_pinned(out);
  // up: staging threads copy slice k into one pinned buffer while the DMA of slice k - 1 drains the other
  if (in_pinned) {
    HIP_TRY(hipMemcpyAsync(w->d_in, frames, ib, hipMemcpyHostToDevice, w->stream));
  } else {
    int k = 0;
    for (size_t off = 0; off < ib; off += slice, ++k) {
      const size_t nb = std::min(slice, ib - off);
      if (k >= 2) HIP_TRY(hipEventSynchronize(w->ev[k & 1]));
      staged_copy(w->pin[k & 1], (const char*)frames + off, nb);
      HIP_TRY(hipMemcpyAsync((char*)w->d_in + off, w->pin[k & 1], nb, hipMemcpyHostToDevice, w->stream));
      HIP_TRY(hipEventRecord(w->ev[k & 1], w->stream));
    }
  }
  TraceSpan span;
  trace_begin(&span, LSPIV_TRACE_PROJECT_HOST, w->stream);
  rc = launch(w->d_in, w->d_out, w->stream);
  if (rc) return rc;
  trace_end(&span, w->stream);
  // down: the DMA of slice k fills one pinned buffer while the staging threads copy slice k - 1 out of the other
  if (out_pinned) {
    HIP_TRY(hipMemcpyAsync(out, w->d_out, ob, hipMemcpyDeviceToHost, w->stream));
    HIP_TRY(hipStreamSynchronize(w->stream));
    return LSPIV_OK;
  }
  const size_t n_slices = (ob + slice - 1) / slice;
  for (size_t k = 0; k <= n_slices; ++k) {
    if (k < n_slices) {
      const size_t off = k * slice, nb = std::min(slice, ob - off);
      HIP_TRY(hipMemcpyAsync(w->pin[k & 1], (const char*)w->d_out + off, nb, hipMemcpyDeviceToHost, w->stream));
      HIP_TRY(hipEventRecord(w->ev[k & 1], w->stream));
    }
    if (k >= 1) {
      const size_t off = (k - 1) * slice, nb = std::min(slice, ob - off);
      HIP_TRY(hipEventSynchronize(w->ev[(k - 1) & 1]));
      staged_copy((char*)out + off, w->pin[(k - 1) & 1], nb);
    }
  }
  return LSPIV_OK;
}

}  // namespace

namespace lspiv_comm_detail {   // lspiv_comm.hip reports through the same thread-local message
int comm_fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
}  // namespace lspiv_comm_detail

struct lspiv_projection {
  int64_t src_h, src_w, dst_h, dst_w;
  int device;
  int *d_nn, *d_grp_of, *d_grp_off, *d_grp_src;
  int *d_qlo1 = nullptr, *d_qlo2 = nullptr;   // quad-window plan for uint8 frames (project.hip), nullptr: not built
  uint32_t* d_qdesc = nullptr;
  int* d_slow_q = nullptr; int n_slow = 0;    // the quads that plan leaves to the per-cell kernel
  // mixed plan for uint8 frames when the plan has group means (round 6, project.hip: project_mix_kernel): per quad mix_nw 8-byte
  // windows, per cell one byte mask per window and the sample count; nullptr: not built
  int* d_mwin = nullptr; uint32_t* d_mcell = nullptr; int mix_nw = 0;
  int* d_mslow = nullptr; int n_mslow = 0;
  // tiled form of the mixed plan (project_tile_kernel): per wave -- a block of 2^tile_lg x 64 / 2^tile_lg quads -- the sorted list of
  // the 8-byte chunks of the camera frame its windows touch (64 * tile_rmax entries, lane l loads entry l), parked in LDS; the windows
  // of its quads as byte offsets into that tile; nullptr: not built
  int* d_wchunk = nullptr; int* d_twin = nullptr; int tile_rmax = 0, tile_lg = 6, tile_wq = 0, tile_rows = 0;
  int* d_tslow = nullptr; int n_tslow = 0;
  // tiles for float32 frames (project_tile_f32_kernel): chunk lists of four pixels, per cell f_dw descriptor words (tile positions of
  // its samples in the reference's order, count, group flag); nullptr: not built
  int* d_fchunk = nullptr; uint32_t* d_fdesc = nullptr; int f_dw = 0, f_rmax = 0, f_lg = 6, f_wq = 0, f_rows = 0;
  int* d_fslow = nullptr; int n_fslow = 0;
  int64_t n_groups = 0;                       // 0: nearest neighbour only -- uint8 frames may stay uint8 (lspiv_project_frames_u8)
};

struct lspiv_ensemble {
  int64_t H, W;
  int wy, wx, oy, ox;
  Grid g;
  int device;
  float* d_sum;    // n_win * wy * wx
  float* d_count;  // n_win
  float* d_part;   // walking kernels: per-segment partial sums + counts (grow-only workspace)
  size_t part_cap;
  int64_t pairs_done;   // pairs accumulated so far = absolute index of the next chunk's first pair (segment anchoring)
  // float64 rescue of the final fit (piv_rescue.hip, ens_*): the chunks' frames and masked corr_max stay reachable until
  // lspiv_ensemble_finish -- owned copies (host entry point: the upload buffer itself; "_dev": a device copy, LSPIV_RETAIN_COPY)
  // or the caller's pointer (LSPIV_RETAIN_BORROW)
  struct Kept { void* d_frames; bool owned; int dtype; int64_t T; float* d_cmax; };
  std::vector<Kept> kept;
  int retain_mode;          // "_dev" entry point: LSPIV_RETAIN_*; the host entry point always keeps its upload buffers
  size_t kept_bytes;        // HBM held BECAUSE of this handle: owned frame copies, the corr_max records, and the borrowed chunks too
                            // (they are the caller's allocations, but it keeps them alive for the handle): all against the budget
  // the chunks' masked corr_max records live in a few large blocks (geometric growth) instead of one hipMalloc per accumulate
  struct CmaxBlock { char* base; size_t cap, used; };
  std::vector<CmaxBlock> cmax_blocks;
  // accumulate_dev may run on caller streams while flag / partials / finish run on the context's stream: one event per stream
  // that accumulated, recorded after each accumulate, waited for by every reader of the sums and of the kept records
  struct AccEvent { hipStream_t stream; hipEvent_t ev; };
  std::vector<AccEvent> acc_events;
  bool retain_complete;     // false: some chunk could not be kept (budget, mode NONE, imported state) -> float32 fits stay
  void* d_rescue; size_t rescue_cap;     // EnsRescueHdr (256 B) + records
  double* d_partial; size_t partial_cap;
  double* d_totals; size_t totals_cap;   // (n_rec, kEnsMaxCand * 5): the partial sums merged over this handle's pair-blocks
  int64_t last_flagged, last_rescued, last_skipped;
  bool foreign;             // the sums were replaced by lspiv_ensemble_import: they hold other handles' pairs as well
  uint32_t n_rec;           // records of the last lspiv_ensemble_flag (sorted by window), 0 if none
  uint64_t rec_digest;      // FNV-1a over (w, ncand, pos[0 .. ncand-1]) of those records: what ranks compare before they sum partials
  float flag_min_count;     // count_min * n_frames of that call
};

// HBM the retained chunks of one ensemble may occupy: LSPIV_ENSEMBLE_RETAIN_BYTES, default a quarter of the device
static size_t ensemble_retain_budget() {
  if (const char* e = getenv("LSPIV_ENSEMBLE_RETAIN_BYTES")) return (size_t)atoll(e);
  size_t f = 0, t = 0;
  if (hipMemGetInfo(&f, &t) != hipSuccess) { (void)hipGetLastError(); return (size_t)16 << 30; }
  return t / 4;
}
static void ensemble_drop_kept(lspiv_ensemble* h) {
  for (auto& k : h->kept)
    if (k.owned && k.d_frames) (void)hipFree(k.d_frames);
  for (auto& b : h->cmax_blocks) (void)hipFree(b.base);
  h->kept.clear();
  h->cmax_blocks.clear();
  h->kept_bytes = 0;
}
// n bytes (256-byte granules) from the handle's record blocks; a new block is twice the last one (>= 1 MiB, >= n, <= 1 GiB unless n
// is larger): a handle that takes hundreds of chunks calls hipMalloc a dozen times, not hundreds
static void* ensemble_cmax_alloc(lspiv_ensemble* h, size_t n) {
  n = (n + 255) & ~(size_t)255;
  if (!h->cmax_blocks.empty()) {
    auto& b = h->cmax_blocks.back();
    if (b.used + n <= b.cap) { void* p = b.base + b.used; b.used += n; return p; }
  }
  const size_t last = h->cmax_blocks.empty() ? 0 : h->cmax_blocks.back().cap;
  const size_t cap = std::max(n, std::min<size_t>(std::max<size_t>(2 * last, (size_t)1 << 20), (size_t)1 << 30));
  void* base = nullptr;
  if (hipMalloc(&base, cap) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  h->cmax_blocks.push_back({(char*)base, cap, n});
  h->kept_bytes += cap;
  return base;
}
// the handle stops being rescuable: what it kept is of no use any more (the float32 fits stay, lspiv_ensemble_stats says so)
static void ensemble_give_up_retention(lspiv_ensemble* h) {
  h->retain_complete = false;
  ensemble_drop_kept(h);
}
// keep the masked corr_max of a chunk (the kernels' keep decisions) next to its frames (`frame_bytes` of them: counted against
// the budget whether the handle owns them or borrows them); on any failure -- budget, allocation, copy -- the ensemble simply stops
// being rescuable, the accumulation itself is not affected.  Takes ownership of an `owned` buffer either way.
static void ensemble_keep(lspiv_ensemble* h, void* d_frames, bool owned, size_t frame_bytes, int dtype, int64_t T, const float* d_cmax,
                          hipStream_t s) {
  const size_t n_tiles = (size_t)(T - 1) * h->g.n_rows * h->g.n_cols;
  void* cm = nullptr;
  if (h->kept_bytes + frame_bytes + n_tiles * sizeof(float) > ensemble_retain_budget() ||
      !(cm = ensemble_cmax_alloc(h, n_tiles * sizeof(float))) ||
      hipMemcpyAsync(cm, d_cmax, n_tiles * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) {
    (void)hipGetLastError();
    if (owned) (void)hipFree(d_frames);
    ensemble_give_up_retention(h);
    return;
  }
  h->kept.push_back({d_frames, owned, dtype, T, (float*)cm});
  h->kept_bytes += frame_bytes;
}
// accumulate_dev ran on stream `s`: note where that stream stands; readers on another stream wait for it
static void ensemble_mark_accumulated(lspiv_ensemble* h, hipStream_t s) {
  lspiv_ensemble::AccEvent* a = nullptr;
  for (auto& e : h->acc_events) if (e.stream == s) a = &e;
  if (!a) {
    hipEvent_t ev = nullptr;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); (void)hipStreamSynchronize(s); return; }
    h->acc_events.push_back({s, ev});
    a = &h->acc_events.back();
  }
  if (hipEventRecord(a->ev, s) != hipSuccess) { (void)hipGetLastError(); (void)hipStreamSynchronize(s); }
}
static void ensemble_wait_accumulated(lspiv_ensemble* h, hipStream_t reader) {
  for (auto& e : h->acc_events)
    if (e.stream != reader && hipStreamWaitEvent(reader, e.ev, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipStreamSynchronize(e.stream); }
}

static std::atomic<int> g_opt_walk{-1};   // lspiv_set_option("walk", v); -1: not set, fall back to the environment
// "stack" signal mode: one pass over the chunk's frames leaves a keep flag per window position; the PIV kernels then run
// their threshold path with a pair threshold that always passes and consult the flags instead
static int apply_signal_mode(DeviceCtx* c, lspiv::PivParams* p, int dtype, hipStream_t s) {
  if (g_opt_signal_mode.load() != 1 || p->signal_threshold < 0.0f) return LSPIV_OK;
  int rc = ensure(&c->d_keep, &c->keep_cap, (size_t)p->n_win);
  if (rc) return rc;
  hipError_t e = lspiv::launch_window_signal(p->frames, dtype, (int64_t)p->n_pairs + 1, *p, p->signal_threshold, c->d_keep, s);
  if (e != hipSuccess) return fail(LSPIV_EHIP, "kernel launch failed: %s", hipGetErrorString(e));
  p->win_keep = c->d_keep;
  p->signal_threshold = 0.0f;
  return LSPIV_OK;
}
uint32_t lspiv::job_slots(int waves_per_simd, int groups) {
  static const int cus = [] {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess || prop.multiProcessorCount < 1) return 256;
    return prop.multiProcessorCount;
  }();
  return (uint32_t)(cus * 4 * waves_per_simd * groups);
}

// window kinds served by the time-walking kernels (every even square window 6 .. 64)
static bool kind_walks(int kind) { return kind == 1 || kind == 2 || kind == 6 || kind == 8; }

int lspiv::walk_setting() {
  const int v = g_opt_walk.load();
  if (v >= 0) return v;
  const char* e = getenv("LSPIV_WALK");
  return e ? atoi(e) : 1;
}

extern "C" {

int lspiv_abi_version(void) { return LSPIV_ABI_VERSION; }
#ifndef LSPIV_KERNEL_HASH
#define LSPIV_KERNEL_HASH "unknown"   // built without csrc/Makefile
#endif
#ifndef LSPIV_SOURCE_HASH
#define LSPIV_SOURCE_HASH "unknown"
#endif
const char* lspiv_version(void) { return "lspiv-hip 0.2.0 (gfx950) src " LSPIV_SOURCE_HASH; }
const char* lspiv_build_info(int what) {
  switch (what) {
    case LSPIV_BUILD_KERNEL_HASH: return LSPIV_KERNEL_HASH;
    case LSPIV_BUILD_SOURCE_HASH: return LSPIV_SOURCE_HASH;
    default: return "";
  }
}
const char* lspiv_last_error(void) { return g_err.c_str(); }

int lspiv_device_count(int* n) {
  if (!n) return fail(LSPIV_EINVAL, "n is NULL");
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  *n = (e == hipSuccess) ? c : 0;
  return LSPIV_OK;
}
int lspiv_set_device(int device) { HIP_TRY(hipSetDevice(device)); return LSPIV_OK; }
int lspiv_get_device(int* device) {
  if (!device) return fail(LSPIV_EINVAL, "device is NULL");
  HIP_TRY(hipGetDevice(device));
  return LSPIV_OK;
}
int lspiv_device_name(int device, char* buf, size_t len) {
  if (!buf || len == 0) return fail(LSPIV_EINVAL, "buf is NULL");
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  snprintf(buf, len, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
  return LSPIV_OK;
}
int lspiv_synchronize(void) { HIP_TRY(hipDeviceSynchronize()); return LSPIV_OK; }

int lspiv_set_option(const char* name, int value) {
  if (!name) return fail(LSPIV_EINVAL, "option name is NULL");
  if (strcmp(name, "walk") == 0) {
    if (value < -1) return fail(LSPIV_EINVAL, "walk must be -1 (environment), 0, 1 or a segment length");
    g_opt_walk.store(value);
    return LSPIV_OK;
  }
  if (strcmp(name, "border_peak") == 0) {
    if (value < 0 || value > 2) return fail(LSPIV_EINVAL, "border_peak must be 0 (NaN), 1 (plane centre) or 2 (integer peak)");
    g_opt_border.store(value);
    return LSPIV_OK;
  }
  if (strcmp(name, "signal_mode") == 0) {
    if (value < 0 || value > 1) return fail(LSPIV_EINVAL, "signal_mode must be 0 (per window pair) or 1 (per window position over the chunk)");
    g_opt_signal_mode.store(value);
    return LSPIV_OK;
  }
  if (strcmp(name, "signal_positive") == 0) {
    if (value < 0 || value > 1) return fail(LSPIV_EINVAL, "signal_positive must be 0 (samples != 0) or 1 (samples > 0)");
    g_opt_signal_pos.store(value);
    return LSPIV_OK;
  }
  if (strcmp(name, "v_sign") == 0) {
    if (value < 0 || value > 1) return fail(LSPIV_EINVAL, "v_sign must be 0 (v = row shift of the peak) or 1 (negated)");
    g_opt_v_sign.store(value);
    return LSPIV_OK;
  }
  if (strcmp(name, "norm_clip") == 0) {
    if (value < 0 || value > 1) return fail(LSPIV_EINVAL, "norm_clip must be 1 (negative lobes of the normalised window removed) or 0");
    g_opt_norm_clip.store(value);
    return LSPIV_OK;
  }
  if (strcmp(name, "std_ddof") == 0) {
    if (value < 0 || value > 1) return fail(LSPIV_EINVAL, "std_ddof must be 0 (population standard deviation) or 1 (sample)");
    g_opt_std_ddof.store(value);
    return LSPIV_OK;
  }
  if (strcmp(name, "round_odd") == 0) {
    if (value < 0 || value > 2) return fail(LSPIV_EINVAL, "round_odd must be 0 (half-even of x / 2), 1 (up) or 2 (down)");
    g_opt_round_odd.store(value);
    return LSPIV_OK;
  }
  if (strcmp(name, "rescue") == 0) {
    if (value < 0 || value > 1) return fail(LSPIV_EINVAL, "rescue must be 0 (float32 results as they are) or 1 (float64 rescue pass)");
    g_opt_rescue.store(value);
    return LSPIV_OK;
  }
  if (strcmp(name, "narrow_offset") == 0) {
    if (value < -1) return fail(LSPIV_EINVAL, "narrow_offset must be -1 (never) or the smallest |DC offset| of a float64 frame that is removed while narrowing");
    g_opt_narrow_offset.store(value);
    return LSPIV_OK;
  }
  if (strcmp(name, "rescue_kappa") == 0) {
    if (value < 0 || value > 1000000) return fail(LSPIV_EINVAL, "rescue_kappa must be 0 .. 1000000 (units of 1e-9)");
    g_opt_rescue_kappa.store(value);
    return LSPIV_OK;
  }
  if (strcmp(name, "rescue_tau") == 0) {
    if (value < 0 || value > 1000000) return fail(LSPIV_EINVAL, "rescue_tau must be 0 .. 1000000 (units of 1e-9)");
    g_opt_rescue_tau.store(value);
    return LSPIV_OK;
  }
  if (strcmp(name, "time_kernel") == 0) {
    if (value < 0 || value > 1) return fail(LSPIV_EINVAL, "time_kernel must be 0 or 1 (HIP events around the PIV kernel of every launch, lspiv_kernel_times)");
    g_opt_time_kernel.store(value);
    return LSPIV_OK;
  }
  return fail(LSPIV_EINVAL, "unknown option '%s'", name);
}
int lspiv_get_option(const char* name, int* value) {
  if (!name || !value) return fail(LSPIV_EINVAL, "NULL argument");
  if (strcmp(name, "walk") == 0) { *value = lspiv::walk_setting(); return LSPIV_OK; }
  if (strcmp(name, "border_peak") == 0) { *value = g_opt_border.load(); return LSPIV_OK; }
  if (strcmp(name, "signal_mode") == 0) { *value = g_opt_signal_mode.load(); return LSPIV_OK; }
  if (strcmp(name, "signal_positive") == 0) { *value = g_opt_signal_pos.load(); return LSPIV_OK; }
  if (strcmp(name, "v_sign") == 0) { *value = g_opt_v_sign.load(); return LSPIV_OK; }
  if (strcmp(name, "norm_clip") == 0) { *value = g_opt_norm_clip.load(); return LSPIV_OK; }
  if (strcmp(name, "std_ddof") == 0) { *value = g_opt_std_ddof.load(); return LSPIV_OK; }
  if (strcmp(name, "round_odd") == 0) { *value = g_opt_round_odd.load(); return LSPIV_OK; }
  if (strcmp(name, "narrow_offset") == 0) { *value = g_opt_narrow_offset.load(); return LSPIV_OK; }
  if (strcmp(name, "rescue") == 0) { *value = g_opt_rescue.load(); return LSPIV_OK; }
  if (strcmp(name, "rescue_kappa") == 0) { *value = g_opt_rescue_kappa.load(); return LSPIV_OK; }
  if (strcmp(name, "rescue_tau") == 0) { *value = g_opt_rescue_tau.load(); return LSPIV_OK; }
  if (strcmp(name, "time_kernel") == 0) { *value = g_opt_time_kernel.load(); return LSPIV_OK; }
  return fail(LSPIV_EINVAL, "unknown option '%s'", name);
}

int lspiv_kernel_times(float* ms, int cap, int* n) {
  if (!ms || !n || cap < 0) return fail(LSPIV_EINVAL, "bad argument");
  std::lock_guard<std::mutex> launch_lock(locks_here().dispatch);
  KernelTimer& t = g_timers[current_device_slot()];
  const int have = std::min(std::min(t.n, kTimerRing), cap);
  for (int i = 0; i < have; ++i) {
    const int k = (t.n - have + i) % kTimerRing;
    HIP_TRY(hipEventSynchronize(t.e1[k]));
    HIP_TRY(hipEventElapsedTime(&ms[i], t.e0[k], t.e1[k]));
  }
  *n = have;
  t.n = 0;
  return LSPIV_OK;
}

int lspiv_rescue_stats(void* stream, int64_t* stats) {
  if (!stats) return fail(LSPIV_EINVAL, "stats is NULL");
  DeviceCtx* c;
  int rc = get_ctx(&c);
  if (rc) return rc;
  hipStream_t s = stream ? (hipStream_t)stream : c->stream;
  void* base = nullptr;
  {
    std::lock_guard<std::mutex> lk(locks_here().lists);
    for (auto& w : c->rescue) if (w.stream == s) base = w.base;
  }
  for (int k = 0; k < 5; ++k) stats[k] = 0;
  if (!base) return LSPIV_OK;   // no pass has run on this stream
  lspiv::RescueHdr h;
  HIP_TRY(hipMemcpyAsync(&h, base, sizeof(h), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  stats[0] = h.last_fit; stats[1] = h.last_amb;
  stats[2] = (int64_t)h.total_fit; stats[3] = (int64_t)h.total_amb; stats[4] = (int64_t)h.total_windows;
  return LSPIV_OK;
}

int lspiv_kernel_kind(int wy, int wx) {
  if (wy < 2 || wx < 2 || wy > LSPIV_MAX_WINDOW || wx > LSPIV_MAX_WINDOW) return LSPIV_EUNSUPPORTED;
  if (wy > 64 || wx > 64) return lspiv::piv_dft_fits(wy, wx) ? 9 : 10;   // 2-D DFT of any shape: LDS-resident up to 128 x 128, HBM slots above
  if (!g_opt_norm_clip.load()) {   // the un-clipped reading of A3 lives in the block-per-window kernels only (a switch, not a tuned path)
    static const int min_area = getenv("LSPIV_DFT_MIN_AREA") ? atoi(getenv("LSPIV_DFT_MIN_AREA")) : 1500;
    return wy * wx >= min_area && lspiv::piv_dft_fits(wy, wx) ? 9 : 3;
  }
  if (wy == 32 && wx == 32) return 1;
  if (wy == 64 && wx == 64) return 2;
  if (wy == 16 && wx == 16) return 6;
  if (wy == 8 && wx == 8 && getenv("LSPIV_NO_PFA") == nullptr) return 6;
  const bool no_pfa = getenv("LSPIV_NO_PFA") != nullptr;             // A/B switch: the P * 2^m sizes without their own FFT kernels
  if (!no_pfa && wy == wx) {
    switch (wy) {
#define LSPIV_PFA_CASE(n) case n:
      LSPIV_PFA_SIZES(LSPIV_PFA_CASE)
#undef LSPIV_PFA_CASE
        return 8;
      default: break;
    }
  }
  const bool no_embed = getenv("LSPIV_NO_EMBED") != nullptr;   // A/B switch: direct kernel for every other size
  if (!no_embed && wy == wx && wy >= 4 && wy <= 8) return 7;
  if (!no_embed && wy == wx && wy >= 9 && wy <= 15) return 4;
  // 17..20: the direct kernel's N^4 multiply-adds are still cheaper than two 64-point transforms per window (measured
  // crossover between 20 and 22, tools/direct_bench.py)
  if (!no_embed && wy == wx && wy > 20 && wy < 32) return 5;
  // what is left -- non-square windows, odd 17 / 19 / 33 .. 63: the direct spatial kernel costs wy wx multiply-adds per plane
  // sample, the LDS-resident DFT passes (wy + wx) complex ones, a composite length in two shorter passes.  Measured on
  // 785 x 875 frames (tools/gpu_ab_direct.sh; M window pairs/s, direct | DFT): 40 x 24 23.2 | 13.1, 35 x 35 14.0 | 12.2,
  // 48 x 32 10.4 | 11.9, 41 x 41 8.1 | 9.2, 64 x 32 6.1 | 11.3, 49 x 49 3.6 | 10.7, 63 x 63 1.6 | 7.2 -- the DFT kernel takes over
  // from 1500 samples per window (LSPIV_DFT_MIN_AREA)
  static const int dft_min_area = getenv("LSPIV_DFT_MIN_AREA") ? atoi(getenv("LSPIV_DFT_MIN_AREA")) : 1500;
  if (wy * wx >= dft_min_area && lspiv::piv_dft_fits(wy, wx)) return 9;
  return 3;
}

int lspiv_grid_shape(int64_t H, int64_t W, int wy, int wx, int oy, int ox, int64_t* n_rows, int64_t* n_cols) {
  if (!n_rows || !n_cols) return fail(LSPIV_EINVAL, "output pointer is NULL");
  int rc = check_window(wy, wx, oy, ox);
  if (rc) return rc;
  *n_rows = axis_shape(H, wy, oy);
  *n_cols = axis_shape(W, wx, ox);
  return LSPIV_OK;
}

int lspiv_grid_coords(int64_t H, int64_t W, int wy, int wx, int oy, int ox, int64_t* rows, int64_t* cols) {
  int64_t nr, nc;
  int rc = lspiv_grid_shape(H, W, wy, wx, oy, ox, &nr, &nc);
  if (rc) return rc;
  if ((nr > 0 && !rows) || (nc > 0 && !cols)) return fail(LSPIV_EINVAL, "output pointer is NULL");
  // ffpiv.window.get_axis_coords restated: int64(arange(n) * (win - overlap) + win / 2.0)
  for (int64_t k = 0; k < nr; ++k) rows[k] = (int64_t)std::floor((double)k * (wy - oy) + wy / 2.0);
  for (int64_t k = 0; k < nc; ++k) cols[k] = (int64_t)std::floor((double)k * (wx - ox) + wx / 2.0);
  return LSPIV_OK;
}

int64_t lspiv_required_bytes(int64_t T, int64_t H, int64_t W, int dtype, int wy, int wx, int oy, int ox,
                             int with_planes) {
  Grid g;
  int rc = make_grid(H, W, wy, wx, oy, ox, &g);
  if (rc) return rc;
  if (dtype < 0 || dtype > 2 || T < 2) return fail(LSPIV_EINVAL, "bad dtype/T");
  const int64_t n_tiles = (T - 1) * g.n_rows * g.n_cols;
  int64_t b = T * H * W * (int64_t)elem_size(dtype) + 4 * n_tiles * 4;
  if (with_planes) b += n_tiles * wy * wx * 4;
  if (g_opt_rescue.load())   // the rescue lists of the launch stream: n_tiles / 4 records of 16 bytes + n_tiles / 16 of 4 (rescue_ws)
    b += 256 + (int64_t)std::max<int64_t>(4096, n_tiles / 4) * 16 + (int64_t)std::max<int64_t>(1024, n_tiles / 16) * 4;
  return b;
}

int lspiv_available_bytes(int64_t* free_bytes, int64_t* total_bytes) {
  DeviceCtx* c;
  int rc = get_ctx(&c);
  if (rc) return rc;
  size_t f = 0, t = 0;
  HIP_TRY(hipMemGetInfo(&f, &t));
  // workspaces this library already holds are reusable, count them as free
  f += c->frames_cap + c->out_cap + c->planes_cap;
  if (free_bytes) *free_bytes = (int64_t)f;
  if (total_bytes) *total_bytes = (int64_t)t;
  return LSPIV_OK;
}

// run length of the window family on small grids (kWalkAnchor), 1 for per-pair kernels, a forced length as it is
static int base_alignment(int wy, int wx) {
  const int kind = lspiv_kernel_kind(wy, wx);
  if (kind < 0) return kind;
  const int walk = lspiv::walk_setting();
  if (!kind_walks(kind) || walk == 0) return 1;
  return walk > 1 ? walk : (int)lspiv::kWalkAnchor;
}
// Without a grid: the alignment that is right for EVERY frame shape -- the longest run length the family uses, a multiple of the
// shorter one (ABI 5; it returned the short one before, which cut chunks off the anchors of large grids: ADVICE r05)
int lspiv_chunk_alignment(int wy, int wx) {
  const int base = base_alignment(wy, wx);
  if (base <= 1 || lspiv::walk_setting() > 1) return base;
  static_assert(lspiv::kWalkAnchorLong % lspiv::kWalkAnchor == 0, "the long anchor must be a multiple of the short one");
  return (int)lspiv::kWalkAnchorLong;
}
// the anchor length the walking kernels use on a grid of n_win windows (common.h, walk_anchor)
static int chunk_alignment_for(int wy, int wx, int64_t n_win) {
  const int base = base_alignment(wy, wx);
  if (base <= 1 || lspiv::walk_setting() > 1) return base;     // per-pair kernels, or a forced anchor length
  return (int)lspiv::walk_anchor(wy, (uint32_t)std::min<int64_t>(n_win, 0x7fffffff));
}
int lspiv_chunk_alignment_grid(int64_t H, int64_t W, int wy, int wx, int oy, int ox) {
  Grid g;
  const int rc = make_grid(H, W, wy, wx, oy, ox, &g);
  if (rc) return rc;
  return chunk_alignment_for(wy, wx, g.n_rows * g.n_cols);
}

int lspiv_piv_pairs_dev_at(const void* d_frames, int dtype, int64_t T, int64_t H, int64_t W, int wy, int wx, int oy,
                           int ox, float signal_threshold, int64_t pair_offset, float* d_out, float* d_corr_planes,
                           void* stream) {
  if (!d_frames || !d_out) return fail(LSPIV_EINVAL, "d_frames / d_out is NULL");
  if (pair_offset < 0) return fail(LSPIV_EINVAL, "pair_offset %lld is negative", (long long)pair_offset);
  Grid g;
  int rc = make_grid(H, W, wy, wx, oy, ox, &g);
  if (rc) return rc;
  DeviceCtx* c;
  rc = get_ctx(&c);
  if (rc) return rc;
  lspiv::PivParams p;
  rc = fill_params(&p, d_frames, dtype, T, H, W, wy, wx, oy, ox, signal_threshold, g);
  if (rc) return rc;
  p.pair_offset = pair_offset;
  p.u = d_out;
  p.v = d_out + (size_t)p.n_tiles;
  p.cmax = d_out + 2 * (size_t)p.n_tiles;
  p.s2n = d_out + 3 * (size_t)p.n_tiles;
  p.planes = d_corr_planes;
  hipStream_t s = stream ? (hipStream_t)stream : c->stream;
  rc = apply_signal_mode(c, &p, dtype, s);
  if (rc) return rc;
  rc = dispatch(p, dtype, false, s);
  if (rc) return rc;
  return apply_v_sign(p.v, (int64_t)p.n_tiles, s);
}

int lspiv_piv_pairs_dev(const void* d_frames, int dtype, int64_t T, int64_t H, int64_t W, int wy, int wx, int oy,
                        int ox, float signal_threshold, float* d_out, float* d_corr_planes, void* stream) {
  return lspiv_piv_pairs_dev_at(d_frames, dtype, T, H, W, wy, wx, oy, ox, signal_threshold, 0, d_out, d_corr_planes, stream);
}

// the host entry point, optionally with the px -> m/s scaling of pyorc/velocimetry/ffpiv.py:418-419 applied on the device before the
// results come back (dt != NULL: seconds per pair, T - 1 entries)
static int piv_pairs_host(const void* frames, int dtype, int64_t T, int64_t H, int64_t W, int wy, int wx, int oy, int ox,
                          float signal_threshold, int64_t pair_offset, float* u, float* v, float* corr_max, float* s2n,
                          float* corr_planes, const double* dt, double res_x, double res_y);

int lspiv_piv_pairs_at(const void* frames, int dtype, int64_t T, int64_t H, int64_t W, int wy, int wx, int oy, int ox,
                       float signal_threshold, int64_t pair_offset, float* u, float* v, float* corr_max, float* s2n,
                       float* corr_planes) {
  return piv_pairs_host(frames, dtype, T, H, W, wy, wx, oy, ox, signal_threshold, pair_offset, u, v, corr_max, s2n, corr_planes, nullptr, 1.0, 1.0);
}

int lspiv_piv_velocity_at(const void* frames, int dtype, int64_t T, int64_t H, int64_t W, int wy, int wx, int oy, int ox,
                          float signal_threshold, int64_t pair_offset, double res_x, double res_y, const double* dt, float* v_x, float* v_y,
                          float* corr_max, float* s2n) {
  if (!dt) return fail(LSPIV_EINVAL, "dt is NULL");
  return piv_pairs_host(frames, dtype, T, H, W, wy, wx, oy, ox, signal_threshold, pair_offset, v_x, v_y, corr_max, s2n, nullptr, dt, res_x, res_y);
}

static int piv_pairs_host(const void* frames, int dtype, int64_t T, int64_t H, int64_t W, int wy, int wx, int oy, int ox,
                          float signal_threshold, int64_t pair_offset, float* u, float* v, float* corr_max, float* s2n,
                          float* corr_planes, const double* dt, double res_x, double res_y) {
  std::lock_guard<std::mutex> host_lock(locks_here().host);
  if (!frames || !u || !v || !corr_max || !s2n) return fail(LSPIV_EINVAL, "NULL buffer");
  if (pair_offset < 0) return fail(LSPIV_EINVAL, "pair_offset %lld is negative", (long long)pair_offset);
  Grid g;
  int rc = make_grid(H, W, wy, wx, oy, ox, &g);
  if (rc) return rc;
  if (dtype < 0 || dtype > 2) return fail(LSPIV_EINVAL, "dtype %d not in {0:u8, 1:f32, 2:f64}", dtype);
  if (T < 2) return fail(LSPIV_ESHAPE, "need at least 2 frames, got %lld", (long long)T);
  DeviceCtx* c;
  rc = get_ctx(&c);
  if (rc) return rc;
  const int dev_dtype = dtype == LSPIV_F64 ? LSPIV_F32 : dtype;   // float64 is narrowed while it is staged
  const size_t fbytes = (size_t)T * H * W * elem_size(dev_dtype);
  const size_t n_tiles = (size_t)(T - 1) * g.n_rows * g.n_cols;
  rc = ensure(&c->d_frames, &c->frames_cap, fbytes);
  if (rc) return rc;
  rc = ensure(&c->d_out, &c->out_cap, 4 * n_tiles * sizeof(float));
  if (rc) return rc;
  if (corr_planes) {
    rc = ensure(&c->d_planes, &c->planes_cap, n_tiles * wy * wx * sizeof(float));
    if (rc) return rc;
  }
  // Pipelined upload: the stack is copied through a two-slot pinned ring in sub-batches of whole frames; the kernel
  // for the pairs that have become resident runs while the next sub-batch is staged and DMA'd.  Launches are cut at
  // the anchors of the walking kernels' segments (multiples of lspiv_chunk_alignment of the absolute pair index), so
  // the pipelined run issues exactly the jobs of one launch over the whole stack: same bits as lspiv_piv_pairs_dev_at.
  lspiv::PivParams base;
  rc = fill_params(&base, c->d_frames, dev_dtype, T, H, W, wy, wx, oy, ox, signal_threshold, g);
  if (rc) return rc;
  const size_t n_win = (size_t)g.n_rows * g.n_cols;
  const size_t frame_bytes = (size_t)H * W * elem_size(dev_dtype);
  const size_t src_frame_bytes = (size_t)H * W * elem_size(dtype);
  rc = stage_ring(c, frame_bytes);
  if (rc) return rc;
  const int64_t fpb = std::max<int64_t>(1, (int64_t)(c->pinned_cap / frame_bytes));
  const bool src_pinned = dtype != LSPIV_F64 && is_pinned(frames);
  const int64_t align = std::max(1, chunk_alignment_for(wy, wx, (int64_t)n_win));
  // "stack" signal mode scores a window position over ALL frames of the chunk: one launch once everything is resident
  const bool whole_chunk_only = g_opt_signal_mode.load() == 1 && signal_threshold >= 0.0f;
  int64_t launched = 0;   // pairs [0, launched) have been issued
  int batch = 0;
  TraceSpan span;
  trace_begin(&span, LSPIV_TRACE_PIV_HOST, c->stream);
  for (int64_t f0 = 0; f0 < T; ++batch) {
    const int64_t f1 = std::min<int64_t>(T, f0 + fpb);
    const int slot = batch & 1;
    if (batch >= 2) HIP_TRY(hipEventSynchronize(c->staged[slot]));  // the slot's previous DMA has drained
    const size_t nb = (size_t)(f1 - f0) * frame_bytes;
    const void* dma_src = c->pinned[slot];
    if (dtype == LSPIV_F64) {
      const double* src64 = (const double*)((const char*)frames + (size_t)f0 * src_frame_bytes);
      const std::vector<double> off = narrow_offsets(src64, (size_t)H * W, f1 - f0, signal_threshold);
      lspiv_host::staged_narrow((float*)c->pinned[slot], src64, (size_t)H * W, (size_t)(f1 - f0), off.data());
    } else if (src_pinned)
      dma_src = (const char*)frames + (size_t)f0 * src_frame_bytes;   // caller's stack is pinned: no staging copy
    else
      staged_copy(c->pinned[slot], (const char*)frames + (size_t)f0 * src_frame_bytes, nb);
    HIP_TRY(hipMemcpyAsync((char*)c->d_frames + (size_t)f0 * frame_bytes, dma_src, nb, hipMemcpyHostToDevice,
                           c->copy_stream));
    HIP_TRY(hipEventRecord(c->staged[slot], c->copy_stream));
    HIP_TRY(hipStreamWaitEvent(c->stream, c->staged[slot], 0));
    // pairs whose two frames are resident: [0, f1 - 1); issue up to the last anchor below that (everything at the end)
    int64_t p1 = f1 - 1;
    if (f1 < T) p1 = whole_chunk_only ? 0 : ((pair_offset + p1) / align) * align - pair_offset;
    const int64_t p0 = launched;
    if (p1 > p0) {
      lspiv::PivParams p = base;
      p.frames = (const char*)c->d_frames + (size_t)p0 * frame_bytes;
      p.pair_offset = pair_offset + p0;
      p.n_pairs = (uint32_t)(p1 - p0);
      p.n_tiles = (uint32_t)((p1 - p0) * n_win);
      p.u = c->d_out + p0 * n_win;
      p.v = c->d_out + n_tiles + p0 * n_win;
      p.cmax = c->d_out + 2 * n_tiles + p0 * n_win;
      p.s2n = c->d_out + 3 * n_tiles + p0 * n_win;
      p.planes = corr_planes ? c->d_planes + (size_t)p0 * n_win * wy * wx : nullptr;
      rc = apply_signal_mode(c, &p, dev_dtype, c->stream);
      if (rc) return rc;
      rc = dispatch(p, dev_dtype, false, c->stream);
      if (rc) return rc;
      rc = apply_v_sign(p.v, (int64_t)p.n_tiles, c->stream);
      if (rc) return rc;
      launched = p1;
    }
    f0 = f1;
  }
  if (dt) {
    // u, v to metres per second where they are: (u * res / dt).astype(float32) -- float32 product, float64 division, one rounding
    // (masks.hip, scale_velocity_kernel: bit for bit what numpy computes for a python-float resolution, tests/test_masks.py)
    rc = ensure(&c->d_scratch, &c->scratch_cap, (size_t)(T - 1) * sizeof(double));
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(c->d_scratch, dt, (size_t)(T - 1) * sizeof(double), hipMemcpyHostToDevice, c->stream));
    const hipError_t e = lspiv::launch_scale_velocity(c->d_out, T - 1, (int64_t)n_win, (float)res_x, (float)res_y, (const double*)c->d_scratch, c->stream);
    if (e != hipSuccess) return fail(LSPIV_EHIP, "kernel launch failed: %s", hipGetErrorString(e));
  }
  const size_t ob = n_tiles * sizeof(float);
  HIP_TRY(hipMemcpyAsync(u, c->d_out, ob, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(v, c->d_out + n_tiles, ob, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(corr_max, c->d_out + 2 * n_tiles, ob, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(s2n, c->d_out + 3 * n_tiles, ob, hipMemcpyDeviceToHost, c->stream));
  if (corr_planes)
    HIP_TRY(hipMemcpyAsync(corr_planes, c->d_planes, n_tiles * wy * wx * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  trace_end(&span, c->stream);
  HIP_TRY(hipStreamSynchronize(c->stream));
  return LSPIV_OK;
}

int lspiv_piv_pairs(const void* frames, int dtype, int64_t T, int64_t H, int64_t W, int wy, int wx, int oy, int ox,
                    float signal_threshold, float* u, float* v, float* corr_max, float* s2n, float* corr_planes) {
  return lspiv_piv_pairs_at(frames, dtype, T, H, W, wy, wx, oy, ox, signal_threshold, 0, u, v, corr_max, s2n, corr_planes);
}

int lspiv_u_v_displacement(const float* corr_planes, int64_t P, int64_t n_win, int wy, int wx, float* u, float* v) {
  std::lock_guard<std::mutex> host_lock(locks_here().host);
  if (!corr_planes || !u || !v) return fail(LSPIV_EINVAL, "NULL buffer");
  if (P < 0 || n_win < 0 || wy < 1 || wx < 1 || wy > 4096 || wx > 4096) return fail(LSPIV_EINVAL, "bad shape");
  const int64_t n = P * n_win;
  if (n == 0) return LSPIV_OK;
  if (n >= (int64_t)1 << 31) return fail(LSPIV_EINVAL, "too many planes for one launch");
  DeviceCtx* c;
  int rc = get_ctx(&c);
  if (rc) return rc;
  const size_t pb = (size_t)n * wy * wx * sizeof(float);
  rc = ensure(&c->d_planes, &c->planes_cap, pb);
  if (rc) return rc;
  rc = ensure(&c->d_out, &c->out_cap, 2 * (size_t)n * sizeof(float));
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(c->d_planes, corr_planes, pb, hipMemcpyHostToDevice, c->stream));
  hipError_t e = lspiv::launch_peaks_from_planes(c->d_planes, (uint32_t)n, wy, wx, g_opt_border.load(), c->d_out, c->d_out + n, c->stream);
  if (e != hipSuccess) return fail(LSPIV_EHIP, "kernel launch failed: %s", hipGetErrorString(e));
  rc = apply_v_sign(c->d_out + n, n, c->stream);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(u, c->d_out, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(v, c->d_out + n, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return LSPIV_OK;
}

// ---- ensemble -------------------------------------------------------------------------------
int lspiv_ensemble_begin(int64_t H, int64_t W, int wy, int wx, int oy, int ox, lspiv_ensemble** handle) {
  if (!handle) return fail(LSPIV_EINVAL, "handle is NULL");
  Grid g;
  int rc = make_grid(H, W, wy, wx, oy, ox, &g);
  if (rc) return rc;
  DeviceCtx* c;
  rc = get_ctx(&c);
  if (rc) return rc;
  lspiv_ensemble* h = new lspiv_ensemble();
  h->H = H; h->W = W; h->wy = wy; h->wx = wx; h->oy = oy; h->ox = ox; h->g = g;
  h->d_sum = nullptr; h->d_count = nullptr; h->d_part = nullptr; h->part_cap = 0; h->pairs_done = 0;
  h->retain_mode = LSPIV_RETAIN_NONE; h->kept_bytes = 0; h->retain_complete = true;
  h->d_rescue = nullptr; h->rescue_cap = 0; h->d_partial = nullptr; h->partial_cap = 0;
  h->last_flagged = h->last_rescued = h->last_skipped = 0;
  h->d_totals = nullptr; h->totals_cap = 0; h->foreign = false; h->n_rec = 0; h->flag_min_count = 0.0f;
  HIP_TRY(hipGetDevice(&h->device));
  const size_t n_win = (size_t)g.n_rows * g.n_cols;
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, n_win * wy * wx * sizeof(float));
  if (e != hipSuccess) { delete h; return fail(LSPIV_ENOMEM, "hipMalloc corr_sum: %s", hipGetErrorString(e)); }
  h->d_sum = (float*)p;
  e = hipMalloc(&p, n_win * sizeof(float));
  if (e != hipSuccess) { hipFree(h->d_sum); delete h; return fail(LSPIV_ENOMEM, "hipMalloc corr_count: %s", hipGetErrorString(e)); }
  h->d_count = (float*)p;
  HIP_TRY(hipMemsetAsync(h->d_sum, 0, n_win * wy * wx * sizeof(float), c->stream));
  HIP_TRY(hipMemsetAsync(h->d_count, 0, n_win * sizeof(float), c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));  // lspiv_ensemble_accumulate_dev may be given another stream
  *handle = h;
  return LSPIV_OK;
}

static int ensemble_launch(lspiv_ensemble* h, DeviceCtx* c, const void* d_frames, int dtype, int64_t T, float corr_min,
                           float s2n_min, float signal_threshold, float* d_cmax, float* d_s2n, hipStream_t s) {
  lspiv::PivParams p;
  int rc = fill_params(&p, d_frames, dtype, T, h->H, h->W, h->wy, h->wx, h->oy, h->ox, signal_threshold, h->g);
  if (rc) return rc;
  p.cmax = d_cmax;
  p.s2n = d_s2n;
  p.corr_min = corr_min;
  p.s2n_min = s2n_min;
  p.corr_sum = h->d_sum;
  p.corr_count = h->d_count;
  const int kind = lspiv_kernel_kind(h->wy, h->wx);
  const int walk = lspiv::walk_setting();
  p.pair_offset = h->pairs_done;   // advanced only once the launch has been issued (a failed accumulate changes nothing)
  if (kind_walks(kind) && walk != 0) {
    // segments anchored at multiples of the anchor length of the absolute pair index (common.h): the partial sums, and
    // the order they are merged in, are the same for every chunking whose boundaries are multiples of that length
    const lspiv::WalkSegments w = lspiv::walk_segments(p.n_pairs, p.pair_offset, walk > 1 ? (uint32_t)walk : lspiv::walk_anchor(h->wy, p.n_win));
    p.seg_len = w.seg_len; p.seg_first = w.seg_first; p.n_seg = w.n_seg;
    const size_t plane = (size_t)h->wy * h->wx;
    const size_t need = (size_t)p.n_seg * p.n_win * (plane + 1) * sizeof(float);
    rc = ensure(&h->d_part, &h->part_cap, need);
    if (rc) return rc;
    // not zeroed: every (segment, window) job writes its whole slot and its count (first iteration stores, later ones add)
    p.part_sum = h->d_part;
    p.part_cnt = h->d_part + (size_t)p.n_seg * p.n_win * plane;
  }
  rc = apply_signal_mode(c, &p, dtype, s);
  if (rc) return rc;
  rc = dispatch(p, dtype, true, s);
  if (rc) return rc;
  h->pairs_done += p.n_pairs;
  return LSPIV_OK;
}

int lspiv_ensemble_accumulate_dev(lspiv_ensemble* h, const void* d_frames, int dtype, int64_t T, float corr_min,
                                  float s2n_min, float signal_threshold, float* d_corr_s2n, void* stream) {
  if (!h || !d_frames || !d_corr_s2n) return fail(LSPIV_EINVAL, "NULL argument");
  if (T < 2) return fail(LSPIV_ESHAPE, "need at least 2 frames, got %lld", (long long)T);
  DeviceCtx* c;
  int rc = get_ctx(&c);
  if (rc) return rc;
  const size_t n_tiles = (size_t)(T - 1) * h->g.n_rows * h->g.n_cols;
  hipStream_t s = stream ? (hipStream_t)stream : c->stream;
  rc = ensemble_launch(h, c, d_frames, dtype, T, corr_min, s2n_min, signal_threshold, d_corr_s2n, d_corr_s2n + n_tiles, s);
  if (rc) return rc;
  // retention for the float64 rescue of the final fit (lspiv_ensemble_set_retain)
  const size_t fbytes = (size_t)T * h->H * h->W * elem_size(dtype);
  if (h->retain_mode == LSPIV_RETAIN_NONE || !h->retain_complete || !g_opt_rescue.load()) {
    if (h->retain_complete) ensemble_give_up_retention(h);
  } else if (h->retain_mode == LSPIV_RETAIN_BORROW) {
    ensemble_keep(h, const_cast<void*>(d_frames), false, fbytes, dtype, T, d_corr_s2n, s);
  } else {
    void* copy = nullptr;
    if (h->kept_bytes + fbytes > ensemble_retain_budget() || hipMalloc(&copy, fbytes) != hipSuccess) {
      (void)hipGetLastError();
      ensemble_give_up_retention(h);
    } else if (hipMemcpyAsync(copy, d_frames, fbytes, hipMemcpyDeviceToDevice, s) != hipSuccess) {
      (void)hipGetLastError(); (void)hipFree(copy);
      ensemble_give_up_retention(h);
    } else {
      ensemble_keep(h, copy, true, fbytes, dtype, T, d_corr_s2n, s);
    }
  }
  ensemble_mark_accumulated(h, s);   // flag / partials / finish (context stream) wait for the sums, the records and the copies
  return LSPIV_OK;
}

int lspiv_ensemble_accumulate(lspiv_ensemble* h, const void* frames, int dtype, int64_t T, float corr_min,
                              float s2n_min, float signal_threshold, float* corr_max, float* s2n) {
  std::lock_guard<std::mutex> host_lock(locks_here().host);
  if (!h || !frames || !corr_max || !s2n) return fail(LSPIV_EINVAL, "NULL argument");
  DeviceCtx* c;
  int rc = get_ctx(&c);
  if (rc) return rc;
  if (dtype < 0 || dtype > 2) return fail(LSPIV_EINVAL, "dtype %d not in {0:u8, 1:f32, 2:f64}", dtype);
  if (T < 2) return fail(LSPIV_ESHAPE, "need at least 2 frames, got %lld", (long long)T);
  const size_t n_win = (size_t)h->g.n_rows * h->g.n_cols, n_tiles = (size_t)(T - 1) * n_win;
  rc = ensure(&c->d_out, &c->out_cap, 2 * n_tiles * sizeof(float));
  if (rc) return rc;
  // Pipelined like lspiv_piv_pairs: the ensemble sums are additive over pairs, so the pairs of sub-batch k are
  // accumulated (in order, on one stream) while sub-batch k+1 is staged and DMA'd.  float64 is narrowed while staged.
  const int dev_dtype = dtype == LSPIV_F64 ? LSPIV_F32 : dtype;
  const size_t frame_elems = (size_t)h->H * h->W;
  const size_t frame_bytes = frame_elems * elem_size(dev_dtype), src_frame_bytes = frame_elems * elem_size(dtype);
  // the chunk is uploaded into a buffer of its own that stays with the handle until finish (float64 rescue of the final fit),
  // as long as the retained chunks fit their budget; beyond it, into the shared workspace as before (float32 fits stay)
  void* own = nullptr;
  if (g_opt_rescue.load() && h->retain_complete && h->kept_bytes + (size_t)T * frame_bytes <= ensemble_retain_budget()) {
    if (hipMalloc(&own, (size_t)T * frame_bytes) != hipSuccess) { (void)hipGetLastError(); own = nullptr; }
  }
  if (!own) {
    if (h->retain_complete) ensemble_give_up_retention(h);
    rc = ensure(&c->d_frames, &c->frames_cap, (size_t)T * frame_bytes);
    if (rc) return rc;
  }
  char* const d_chunk = own ? (char*)own : (char*)c->d_frames;
  struct OwnGuard { void* p; ~OwnGuard() { if (p) (void)hipFree(p); } } own_guard{own};   // released on every error path below
  rc = stage_ring(c, frame_bytes);
  if (rc) return rc;
  const int64_t fpb = std::max<int64_t>(1, (int64_t)(c->pinned_cap / frame_bytes));
  const int64_t align = std::max(1, chunk_alignment_for(h->wy, h->wx, (int64_t)n_win)), base_offset = h->pairs_done;
  int64_t launched = 0;
  {
    int batch = 0;
    for (int64_t f0 = 0; f0 < T; ++batch) {
      const int64_t f1 = std::min<int64_t>(T, f0 + fpb);
      const int slot = batch & 1;
      if (batch >= 2) HIP_TRY(hipEventSynchronize(c->staged[slot]));
      const size_t nb = (size_t)(f1 - f0) * frame_bytes;
      if (dtype == LSPIV_F64) {
        const double* src64 = (const double*)((const char*)frames + (size_t)f0 * src_frame_bytes);
        const std::vector<double> off = narrow_offsets(src64, frame_elems, f1 - f0, signal_threshold);
        lspiv_host::staged_narrow((float*)c->pinned[slot], src64, frame_elems, (size_t)(f1 - f0), off.data());
      } else {
        staged_copy(c->pinned[slot], (const char*)frames + (size_t)f0 * src_frame_bytes, nb);
      }
      HIP_TRY(hipMemcpyAsync(d_chunk + (size_t)f0 * frame_bytes, c->pinned[slot], nb, hipMemcpyHostToDevice, c->copy_stream));
      HIP_TRY(hipEventRecord(c->staged[slot], c->copy_stream));
      HIP_TRY(hipStreamWaitEvent(c->stream, c->staged[slot], 0));
      // pairs [0, f1 - 1) are resident; accumulate up to the last segment anchor below that (everything at the end)
      int64_t p1 = f1 - 1;
      if (f1 < T) p1 = (g_opt_signal_mode.load() == 1 && signal_threshold >= 0.0f) ? 0 : ((base_offset + p1) / align) * align - base_offset;
      const int64_t p0 = launched;
      if (p1 > p0) {
        rc = ensemble_launch(h, c, d_chunk + (size_t)p0 * frame_bytes, dev_dtype, p1 - p0 + 1, corr_min, s2n_min,
                             signal_threshold, c->d_out + p0 * n_win, c->d_out + n_tiles + p0 * n_win, c->stream);
        if (rc) return rc;
        launched = p1;
      }
      f0 = f1;
    }
  }
  if (own) {
    own_guard.p = nullptr;          // the handle owns it from here
    ensemble_keep(h, own, true, (size_t)T * frame_bytes, dev_dtype, T, c->d_out, c->stream);
  }
  HIP_TRY(hipMemcpyAsync(corr_max, c->d_out, n_tiles * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(s2n, c->d_out + n_tiles, n_tiles * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return LSPIV_OK;
}

// ---- float64 rescue of the final fit, in stages (piv_rescue.hip, ens_*) ---------------------------------------------------
// mean planes (count filter) -> c->d_planes, their float32 fits -> c->d_out [u | v]
static int ensemble_mean_fit(lspiv_ensemble* h, DeviceCtx* c, float min_count) {
  const size_t n_win = (size_t)h->g.n_rows * h->g.n_cols;
  ensemble_wait_accumulated(h, c->stream);   // accumulate_dev may have run on caller streams
  int rc = ensure(&c->d_planes, &c->planes_cap, n_win * h->wy * h->wx * sizeof(float));
  if (rc) return rc;
  rc = ensure(&c->d_out, &c->out_cap, 2 * n_win * sizeof(float));
  if (rc) return rc;
  hipError_t e = lspiv::launch_ensemble_mean(h->d_sum, h->d_count, min_count, (uint32_t)n_win, h->wy * h->wx, c->d_planes, c->stream);
  if (e != hipSuccess) return fail(LSPIV_EHIP, "kernel launch failed: %s", hipGetErrorString(e));
  e = lspiv::launch_peaks_from_planes(c->d_planes, (uint32_t)n_win, h->wy, h->wx, g_opt_border.load(), c->d_out, c->d_out + n_win, c->stream);
  if (e != hipSuccess) return fail(LSPIV_EHIP, "kernel launch failed: %s", hipGetErrorString(e));
  return LSPIV_OK;
}
static lspiv::EnsRescueRec* ensemble_recs(lspiv_ensemble* h) { return reinterpret_cast<lspiv::EnsRescueRec*>((char*)h->d_rescue + 256); }

// flag the windows whose float32 fit (c->d_out, of the mean planes in c->d_planes) cannot be trusted to 1e-4; the records end
// up sorted by window index -- the same list on every handle that holds the same state (multi-GPU: after the all-reduce)
static int ensemble_flag(lspiv_ensemble* h, DeviceCtx* c) {
  h->last_flagged = h->last_rescued = h->last_skipped = 0;
  h->n_rec = 0;
  const uint32_t n_win = (uint32_t)(h->g.n_rows * h->g.n_cols);
  const size_t hdr_bytes = 256;
  int rc = ensure(&h->d_rescue, &h->rescue_cap, hdr_bytes + (size_t)n_win * sizeof(lspiv::EnsRescueRec));
  if (rc) return rc;
  lspiv::EnsRescueHdr* d_hdr = static_cast<lspiv::EnsRescueHdr*>(h->d_rescue);
  HIP_TRY(hipMemsetAsync(d_hdr, 0, hdr_bytes, c->stream));
  // flag model of the per-pair epilogues (fill_params); the block-per-window kernels (kinds 3 / 9 / 10) assume twice the plane noise
  // of the fused FFT kernels there, and so does the mean of their planes here
  const int kind = lspiv_kernel_kind(h->wy, h->wx);
  const double noise_mult = (kind == 3 || kind == 9 || kind == 10) ? 2.0 : 1.0;
  const float k = (float)(noise_mult * 2.0 * g_opt_rescue_kappa.load() * 1e-9 / (0.6931471805599453 * 1e-4));
  hipError_t e = lspiv::launch_ens_flag(c->d_planes, n_win, h->wy, h->wx, c->d_out, c->d_out + n_win, k, (float)(g_opt_rescue_tau.load() * 1e-9),
                                        d_hdr, ensemble_recs(h), n_win, c->stream);
  if (e != hipSuccess) return fail(LSPIV_EHIP, "kernel launch failed: %s", hipGetErrorString(e));
  lspiv::EnsRescueHdr hdr;
  HIP_TRY(hipMemcpyAsync(&hdr, d_hdr, sizeof(hdr), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  h->last_flagged = hdr.n_rec;
  h->last_skipped = hdr.n_skipped;
  const uint32_t n_rec = std::min<uint32_t>(hdr.n_rec, n_win);
  uint64_t digest = 0xcbf29ce484222325ull;   // FNV-1a
  if (n_rec > 0) {   // the kernel appends in whatever order its waves finish: sort (a few records, once per video)
    std::vector<lspiv::EnsRescueRec> recs(n_rec);
    HIP_TRY(hipMemcpy(recs.data(), ensemble_recs(h), n_rec * sizeof(lspiv::EnsRescueRec), hipMemcpyDeviceToHost));
    std::sort(recs.begin(), recs.end(), [](const lspiv::EnsRescueRec& a, const lspiv::EnsRescueRec& b) { return a.w < b.w; });
    if (n_rec > 1) HIP_TRY(hipMemcpy(ensemble_recs(h), recs.data(), n_rec * sizeof(lspiv::EnsRescueRec), hipMemcpyHostToDevice));
    auto mix = [&digest](uint32_t v) { for (int b = 0; b < 4; ++b) { digest ^= (v >> (8 * b)) & 0xffu; digest *= 0x100000001b3ull; } };
    for (const auto& r : recs) {
      mix(r.w); mix(r.ncand);
      for (uint32_t k = 0; k < std::min<uint32_t>(r.ncand, lspiv::kEnsMaxCand); ++k) mix(r.pos[k]);
    }
  }
  h->rec_digest = digest;
  h->n_rec = n_rec;
  return LSPIV_OK;
}

// this handle's share of the float64 sums: over the pairs of its retained chunks, merged in pair-block order -> h->d_totals
// (n_rec, kEnsMaxCand * 5).  *complete = false (and zeros) when some chunk of this handle could not be kept.
static int ensemble_partials(lspiv_ensemble* h, DeviceCtx* c, bool* complete) {
  const size_t row = (size_t)lspiv::kEnsMaxCand * 5 * sizeof(double);
  int rc = ensure(&h->d_totals, &h->totals_cap, std::max<size_t>(1, h->n_rec) * row);
  if (rc) return rc;
  ensemble_wait_accumulated(h, c->stream);   // the kept records and frame copies were written on the accumulating streams
  HIP_TRY(hipMemsetAsync(h->d_totals, 0, std::max<size_t>(1, h->n_rec) * row, c->stream));
  *complete = h->retain_complete;
  if (h->n_rec == 0 || !h->retain_complete || h->kept.empty()) return LSPIV_OK;
  uint32_t n_blk = 0;
  for (const auto& kp : h->kept) n_blk += (uint32_t)((kp.T - 1 + lspiv::kEnsPairBlock - 1) / lspiv::kEnsPairBlock);
  const size_t per_rec = (size_t)n_blk * row;
  if ((size_t)h->n_rec * per_rec > ((size_t)4 << 30)) { *complete = false; return LSPIV_OK; }   // (thousands of flagged windows x thousands of pair-blocks)
  rc = ensure(&h->d_partial, &h->partial_cap, (size_t)h->n_rec * per_rec);
  if (rc) return rc;
  lspiv::EnsRescueArgs a;
  memset(&a, 0, sizeof(a));
  a.recs = ensemble_recs(h); a.n_rec = h->n_rec; a.n_blk = n_blk; a.partial = h->d_partial; a.count = h->d_count;
  lspiv::PivParams p;
  uint32_t blk0 = 0;
  for (const auto& kp : h->kept) {
    rc = fill_params(&p, kp.d_frames, kp.dtype, kp.T, h->H, h->W, h->wy, h->wx, h->oy, h->ox, -1.0f, h->g);
    if (rc) return rc;
    a.cmax = kp.d_cmax; a.n_pairs = (uint32_t)(kp.T - 1); a.blk0 = blk0;
    hipError_t e = lspiv::launch_ens_partial(p, kp.dtype, a, c->stream);
    if (e != hipSuccess) return fail(LSPIV_EHIP, "kernel launch failed: %s", hipGetErrorString(e));
    blk0 += (a.n_pairs + lspiv::kEnsPairBlock - 1) / lspiv::kEnsPairBlock;
  }
  hipError_t e = lspiv::launch_ens_merge(a, h->d_totals, c->stream);
  if (e != hipSuccess) return fail(LSPIV_EHIP, "kernel launch failed: %s", hipGetErrorString(e));
  return LSPIV_OK;
}

// the fit of the flagged windows from the float64 totals (all pairs of the sum), overwriting c->d_out [u | v]
static int ensemble_final(lspiv_ensemble* h, DeviceCtx* c, const double* d_totals) {
  if (h->n_rec == 0) return LSPIV_OK;
  const size_t n_win = (size_t)h->g.n_rows * h->g.n_cols;
  lspiv::PivParams p;
  memset(&p, 0, sizeof(p));
  p.wy = h->wy; p.wx = h->wx; p.border_mode = g_opt_border.load();
  lspiv::EnsRescueArgs a;
  memset(&a, 0, sizeof(a));
  a.recs = ensemble_recs(h); a.n_rec = h->n_rec; a.count = h->d_count;
  hipError_t e = lspiv::launch_ens_final(p, a, d_totals, c->d_out, c->d_out + n_win, c->stream);
  if (e != hipSuccess) return fail(LSPIV_EHIP, "kernel launch failed: %s", hipGetErrorString(e));
  h->last_rescued = (int64_t)h->n_rec - h->last_skipped;
  return LSPIV_OK;
}

// results of c->d_out / c->d_planes / the count to the caller ("v_sign" applied first)
static int ensemble_deliver(lspiv_ensemble* h, DeviceCtx* c, float* u, float* v, float* corr_count, float* corr_mean) {
  const size_t n_win = (size_t)h->g.n_rows * h->g.n_cols;
  int rc = apply_v_sign(c->d_out + n_win, (int64_t)n_win, c->stream);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(u, c->d_out, n_win * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(v, c->d_out + n_win, n_win * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  if (corr_count) HIP_TRY(hipMemcpyAsync(corr_count, h->d_count, n_win * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  if (corr_mean) HIP_TRY(hipMemcpyAsync(corr_mean, c->d_planes, n_win * h->wy * h->wx * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return LSPIV_OK;
}

int lspiv_ensemble_flag(lspiv_ensemble* h, float count_min, float n_frames, int64_t* n_records) {
  std::lock_guard<std::mutex> host_lock(locks_here().host);
  if (!h || !n_records) return fail(LSPIV_EINVAL, "NULL argument");
  DeviceCtx* c;
  int rc = get_ctx(&c);
  if (rc) return rc;
  h->flag_min_count = count_min * n_frames;
  h->n_rec = 0;
  *n_records = 0;
  if (!g_opt_rescue.load()) return LSPIV_OK;
  rc = ensemble_mean_fit(h, c, h->flag_min_count);
  if (rc) return rc;
  rc = ensemble_flag(h, c);
  if (rc) return rc;
  *n_records = h->n_rec;
  return LSPIV_OK;
}

int lspiv_ensemble_flag_digest(lspiv_ensemble* h, uint64_t* digest) {
  if (!h || !digest) return fail(LSPIV_EINVAL, "NULL argument");
  *digest = h->n_rec ? h->rec_digest : 0;
  return LSPIV_OK;
}

int lspiv_ensemble_partials(lspiv_ensemble* h, double* partials, int* complete) {
  std::lock_guard<std::mutex> host_lock(locks_here().host);
  if (!h || !complete || (h->n_rec && !partials)) return fail(LSPIV_EINVAL, "NULL argument");
  DeviceCtx* c;
  int rc = get_ctx(&c);
  if (rc) return rc;
  bool ok = false;
  rc = ensemble_partials(h, c, &ok);
  if (rc) return rc;
  *complete = ok ? 1 : 0;
  if (h->n_rec)
    HIP_TRY(hipMemcpyAsync(partials, h->d_totals, (size_t)h->n_rec * LSPIV_ENS_PARTIAL_DOUBLES * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return LSPIV_OK;
}

int lspiv_ensemble_finish_partials(lspiv_ensemble* h, const double* partials, float* u, float* v, float* corr_count, float* corr_mean) {
  std::lock_guard<std::mutex> host_lock(locks_here().host);
  if (!h || !u || !v || (h->n_rec && !partials)) return fail(LSPIV_EINVAL, "NULL argument");
  DeviceCtx* c;
  int rc = get_ctx(&c);
  if (rc) return rc;
  rc = ensemble_mean_fit(h, c, h->flag_min_count);   // the shared workspaces may have been used since lspiv_ensemble_flag
  if (rc) return rc;
  if (h->n_rec) {
    const size_t bytes = (size_t)h->n_rec * LSPIV_ENS_PARTIAL_DOUBLES * sizeof(double);
    rc = ensure(&h->d_totals, &h->totals_cap, bytes);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(h->d_totals, partials, bytes, hipMemcpyHostToDevice, c->stream));
    rc = ensemble_final(h, c, h->d_totals);
    if (rc) return rc;
  }
  return ensemble_deliver(h, c, u, v, corr_count, corr_mean);
}

int lspiv_ensemble_set_retain(lspiv_ensemble* h, int mode) {
  if (!h) return fail(LSPIV_EINVAL, "NULL argument");
  if (mode < LSPIV_RETAIN_NONE || mode > LSPIV_RETAIN_BORROW) return fail(LSPIV_EINVAL, "retain mode %d not in {0, 1, 2}", mode);
  h->retain_mode = mode;
  return LSPIV_OK;
}

int lspiv_ensemble_stats(lspiv_ensemble* h, int64_t* stats) {
  if (!h || !stats) return fail(LSPIV_EINVAL, "NULL argument");
  stats[0] = h->last_flagged; stats[1] = h->last_rescued; stats[2] = h->last_skipped;
  stats[3] = (int64_t)h->kept.size(); stats[4] = (int64_t)h->kept_bytes; stats[5] = h->retain_complete ? 1 : 0;
  return LSPIV_OK;
}

int lspiv_ensemble_finish(lspiv_ensemble* h, float count_min, float n_frames, float* u, float* v, float* corr_count,
                          float* corr_mean) {
  std::lock_guard<std::mutex> host_lock(locks_here().host);
  if (!h || !u || !v) return fail(LSPIV_EINVAL, "NULL argument");
  DeviceCtx* c;
  int rc = get_ctx(&c);
  if (rc) return rc;
  h->flag_min_count = count_min * n_frames;
  h->n_rec = 0;
  rc = ensemble_mean_fit(h, c, h->flag_min_count);
  if (rc) return rc;
  if (g_opt_rescue.load()) {
    // float64 rescue of the ill-conditioned fits (include/lspiv.h): needs the frames of EVERY pair in the sum -- a state that
    // was imported holds other handles' pairs (the multi-GPU path runs the three stages itself and all-reduces the partials)
    rc = ensemble_flag(h, c);
    if (rc) return rc;
    bool complete = false;
    if (h->n_rec && !h->foreign) {
      rc = ensemble_partials(h, c, &complete);
      if (rc) return rc;
    }
    if (h->n_rec && complete) rc = ensemble_final(h, c, h->d_totals);
    else h->last_skipped = h->last_flagged;
    if (rc) return rc;
  }
  return ensemble_deliver(h, c, u, v, corr_count, corr_mean);
}

int lspiv_ensemble_export(lspiv_ensemble* h, float* corr_sum, float* corr_count) {
  if (!h || !corr_sum || !corr_count) return fail(LSPIV_EINVAL, "NULL argument");
  DeviceCtx* c;
  int rc = get_ctx(&c);
  if (rc) return rc;
  const size_t n_win = (size_t)h->g.n_rows * h->g.n_cols;
  ensemble_wait_accumulated(h, c->stream);
  HIP_TRY(hipMemcpyAsync(corr_sum, h->d_sum, n_win * h->wy * h->wx * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(corr_count, h->d_count, n_win * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return LSPIV_OK;
}

int lspiv_ensemble_import(lspiv_ensemble* h, const float* corr_sum, const float* corr_count, int add) {
  if (!h || !corr_sum || !corr_count) return fail(LSPIV_EINVAL, "NULL argument");
  DeviceCtx* c;
  int rc = get_ctx(&c);
  if (rc) return rc;
  const size_t n_win = (size_t)h->g.n_rows * h->g.n_cols, np = n_win * h->wy * h->wx;
  // the sums now hold pairs whose frames this handle never saw.  Replaced by a total over several handles (multi-GPU: the
  // all-reduced state): the staged finish (lspiv_ensemble_flag / _partials / _finish_partials) still reaches every pair, each handle
  // through its own retained chunks.  Added to: this handle's chunks no longer tell which pairs are in the sum -- float32 fits.
  if (add) ensemble_give_up_retention(h);
  else h->foreign = true;
  ensemble_wait_accumulated(h, c->stream);
  if (!add) {
    HIP_TRY(hipMemcpyAsync(h->d_sum, corr_sum, np * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(h->d_count, corr_count, n_win * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return LSPIV_OK;
  }
  // add on the host side of the boundary: export, sum, import (a few tens of MB, once per video)
  std::vector<float> s(np), k(n_win);
  rc = lspiv_ensemble_export(h, s.data(), k.data());
  if (rc) return rc;
  for (size_t i = 0; i < np; ++i) s[i] += corr_sum[i];
  for (size_t i = 0; i < n_win; ++i) k[i] += corr_count[i];
  return lspiv_ensemble_import(h, s.data(), k.data(), 0);
}

int lspiv_ensemble_destroy(lspiv_ensemble* h) {
  if (!h) return LSPIV_OK;
  if (h->d_sum) hipFree(h->d_sum);
  if (h->d_count) hipFree(h->d_count);
  if (h->d_part) hipFree(h->d_part);
  ensemble_drop_kept(h);
  for (auto& e : h->acc_events) (void)hipEventDestroy(e.ev);
  if (h->d_rescue) hipFree(h->d_rescue);
  if (h->d_partial) hipFree(h->d_partial);
  if (h->d_totals) hipFree(h->d_totals);
  delete h;
  return LSPIV_OK;
}

// ---- tiles of the orthoprojection plans (project.hip: project_tile_kernel, project_tile_f32_kernel) ------------------------------
// A wave owns a block of 64 quads of the ortho grid, 2^lg quads wide and 64 / 2^lg rows high (grids whose rows are not whole quads: 64
// consecutive quads of the flat index), and loads the sorted list of the aligned CHUNKS of the camera frame its cells read (8 bytes of a
// uint8 frame, four pixels of a float32 frame), one chunk per lane and list row.
extern "C++" {
namespace {
struct TileShape {
  int lg = 6, rmax = 0;                                     // block width 2^lg quads; list rows of 64 chunks (1, 2 or 4)
  int64_t wq = 0, rows = 0;                                 // quads per grid row, grid rows (flat: all quads in one row)
  int64_t bqx() const { return (int64_t)1 << lg; }
  int64_t bqy() const { return 64 >> lg; }
  int64_t tiles_x() const { return (wq + bqx() - 1) / bqx(); }
  int64_t n_waves() const { return tiles_x() * ((rows + bqy() - 1) / bqy()); }
};

// the sorted chunk list of wave wv into lst; QC: bool(size_t quad, std::vector<int>& lst) appends the chunks a quad reads, false: the
// quad is not served by the tiles.  Returns whether the wave has a quad of its own.
template <class QC>
bool tile_wave_list(const TileShape& sh, int64_t wv, QC& quad_chunks, std::vector<int>& lst) {
  lst.clear();
  const int64_t ty = wv / sh.tiles_x(), tx = wv % sh.tiles_x();
  bool any = false;
  for (int64_t r = ty * sh.bqy(); r < std::min(sh.rows, (ty + 1) * sh.bqy()); ++r)
    for (int64_t c = tx * sh.bqx(); c < std::min(sh.wq, (tx + 1) * sh.bqx()); ++c) any = quad_chunks((size_t)(r * sh.wq + c), lst) || any;
  std::sort(lst.begin(), lst.end());
  lst.erase(std::unique(lst.begin(), lst.end()), lst.end());
  return any;
}

// One list row (64 chunks) when all but 1 in 100 waves fit, else two, else four (2 : 1 oversampling and beyond); among the shapes with
// the shortest lists the one with the fewest chunks wins; waves beyond the list hand their quads to a slow kernel; more than 1 in 100
// beyond four rows: no tiles.  *cap_limit: the list length beyond which a wave goes to the slow kernel (the LSPIV_PROJECT_TILE_CAP hook).
template <class QC>
bool tile_pick_shape(int64_t dst_h, int64_t dst_w, size_t nq, QC& quad_chunks, const char* what, TileShape* out, int* cap_limit) {
  std::vector<TileShape> shapes;
  auto shape = [&](int lg, int64_t wq, int64_t rows) { TileShape t; t.lg = lg; t.wq = wq; t.rows = rows; return t; };
  if (dst_w % 4 == 0) for (int lg : {5, 4, 6, 3}) shapes.push_back(shape(lg, dst_w / 4, dst_h));
  else shapes.push_back(shape(6, (int64_t)nq, 1));
  if (const char* f = getenv("LSPIV_PROJECT_TILE_LG")) {       // A/B: force a block width
    const int lg = atoi(f);
    if (dst_w % 4 == 0 && lg >= 0 && lg <= 6) { shapes.clear(); shapes.push_back(shape(lg, dst_w / 4, dst_h)); }
  }
  const bool say = getenv("LSPIV_PROJECT_DEBUG") != nullptr;
  std::vector<int> lst;
  int best = -1;
  int64_t best_total = 0;
  for (size_t i = 0; i < shapes.size(); ++i) {
    TileShape& sh = shapes[i];
    size_t over1 = 0, over2 = 0, over4 = 0;
    int64_t total = 0, seen = 0;
    // the shapes are compared on a sample of their waves (every k-th, about a thousand: a 1080p grid has 4 600 per shape and four shapes
    // per plan; the builder of the chosen shape then visits every wave and sends whatever does not fit to the slow kernel)
    const int64_t step = std::max<int64_t>(1, sh.n_waves() / 1024);
    for (int64_t wv = 0; wv < sh.n_waves(); wv += step, ++seen) {
      tile_wave_list(sh, wv, quad_chunks, lst);
      over1 += lst.size() > 64; over2 += lst.size() > 128; over4 += lst.size() > 256;
      total += (int64_t)lst.size();
    }
    const size_t few = (size_t)seen / 100;                    // waves a shape may leave to the slow kernel
    sh.rmax = over1 <= few ? 1 : over2 <= few ? 2 : over4 <= few ? 4 : 0;
    total = total * sh.n_waves() / std::max<int64_t>(seen, 1);   // (shapes differ in their number of waves: compare chunks per grid)
    if (say)
      fprintf(stderr, "lspiv projection (%s): blocks of %lld x %lld quads: %.1f chunks per wave, of %lld sampled waves (%lld in all) %zu need more than 64, %zu more than 128, %zu more than 256\n",
              what, (long long)sh.bqx(), (long long)sh.bqy(), (double)total / (double)sh.n_waves(), (long long)seen, (long long)sh.n_waves(), over1, over2, over4);
    if (!sh.rmax) continue;
    if (best < 0 || sh.rmax < shapes[(size_t)best].rmax || (sh.rmax == shapes[(size_t)best].rmax && total < best_total)) { best = (int)i; best_total = total; }
  }
  if (const char* f = getenv("LSPIV_PROJECT_TILE_RMAX"))       // A/B: more list rows than the plan needs
    if (best >= 0 && (atoi(f) == 2 || atoi(f) == 4)) shapes[(size_t)best].rmax = std::max(shapes[(size_t)best].rmax, atoi(f));
  *cap_limit = 256;
  if (const char* f = getenv("LSPIV_PROJECT_TILE_CAP")) {      // test hook: waves with longer lists go to the slow kernel, whatever their share
    *cap_limit = std::max(2, atoi(f));
    if (best < 0) { best = 0; shapes[0].rmax = 1; }
  }
  if (best < 0) return false;
  *out = shapes[(size_t)best];
  return true;
}
}  // namespace
}  // extern "C++"

// ---- orthoprojection (N1) and int16 packing (N4) -------------------------------------------------
int lspiv_projection_create(int64_t src_h, int64_t src_w, int64_t dst_h, int64_t dst_w, const int64_t* idx_img,
                            const int64_t* idx_ortho, int64_t K, const int64_t* src_idx, const int64_t* norm_idx,
                            int64_t M, const int64_t* uidx, int64_t G, lspiv_projection** handle) {
  if (!handle) return fail(LSPIV_EINVAL, "handle is NULL");
  if (src_h <= 0 || src_w <= 0 || dst_h <= 0 || dst_w <= 0 || K < 0 || M < 0 || G < 0)
    return fail(LSPIV_ESHAPE, "bad projection shape");
  const int64_t n_src = src_h * src_w, n_out = dst_h * dst_w;
  if (n_src >= (int64_t)1 << 31 || n_out >= (int64_t)1 << 31 || M >= (int64_t)1 << 31)
    return fail(LSPIV_EINVAL, "projection too large for 32-bit indices");
  if ((K > 0 && (!idx_img || !idx_ortho)) || (M > 0 && (!src_idx || !norm_idx)) || (G > 0 && !uidx))
    return fail(LSPIV_EINVAL, "NULL index array");
  if (M > 0 && G == 0) return fail(LSPIV_EINVAL, "group samples without groups");
  DeviceCtx* c;
  int rc = get_ctx(&c);
  if (rc) return rc;
  // host plan: nearest source per cell; CSR of group members in their ORIGINAL order (stable counting sort)
  std::vector<int> nn((size_t)n_out, -1), grp_of((size_t)n_out, -1), off((size_t)G + 1, 0), members((size_t)M);
  for (int64_t k = 0; k < K; ++k) {
    if (idx_ortho[k] < 0 || idx_ortho[k] >= n_out || idx_img[k] < 0 || idx_img[k] >= n_src)
      return fail(LSPIV_EINVAL, "nearest-neighbour index %lld out of range", (long long)k);
    nn[(size_t)idx_ortho[k]] = (int)idx_img[k];
  }
  for (int64_t g = 0; g < G; ++g) {
    if (uidx[g] < 0 || uidx[g] >= n_out) return fail(LSPIV_EINVAL, "uidx[%lld] out of range", (long long)g);
    grp_of[(size_t)uidx[g]] = (int)g;
  }
  for (int64_t i = 0; i < M; ++i) {
    if (norm_idx[i] < 0 || norm_idx[i] >= G || src_idx[i] < 0 || src_idx[i] >= n_src)
      return fail(LSPIV_EINVAL, "group sample %lld out of range", (long long)i);
    off[(size_t)norm_idx[i] + 1]++;
  }
  for (int64_t g = 0; g < G; ++g) off[(size_t)g + 1] += off[(size_t)g];
  {
    std::vector<int> cur(off.begin(), off.end() - 1);
    for (int64_t i = 0; i < M; ++i) members[(size_t)cur[(size_t)norm_idx[i]]++] = (int)src_idx[i];
  }
  lspiv_projection* h = new lspiv_projection();
  h->src_h = src_h; h->src_w = src_w; h->dst_h = dst_h; h->dst_w = dst_w;
  h->n_groups = G;
  h->d_nn = h->d_grp_of = h->d_grp_off = h->d_grp_src = nullptr;
  HIP_TRY(hipGetDevice(&h->device));
  auto up = [&](int** d, const std::vector<int>& v) -> hipError_t {
    const size_t b = std::max<size_t>(v.size(), 1) * sizeof(int);
    hipError_t e = hipMalloc((void**)d, b);
    if (e != hipSuccess) return e;
    // on the library's stream, then waited for: a null-stream hipMemcpy from pageable memory may return before the
    // DMA has landed, and the (non-blocking) stream the kernels run on does not synchronise with the null stream
    if (v.empty()) return hipSuccess;
    e = hipMemcpyAsync(*d, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice, c->stream);
    return e != hipSuccess ? e : hipStreamSynchronize(c->stream);
  };
  hipError_t e = up(&h->d_nn, nn);
  if (e == hipSuccess) e = up(&h->d_grp_of, grp_of);
  if (e == hipSuccess) e = up(&h->d_grp_off, off);
  if (e == hipSuccess) e = up(&h->d_grp_src, members);
  // quad-window plan (uint8 frames): per four consecutive output cells two 8-byte source windows and, per cell, window +
  // byte.  Built when the grid has whole quads and most of them fit (a smooth homography: a few source bytes per quad,
  // one camera row or two); LSPIV_PROJECT_ONE_CELL=1 keeps the one-cell kernel for A/B.
  if (e == hipSuccess && n_out % 4 == 0 && n_src >= 8 && !getenv("LSPIV_PROJECT_ONE_CELL")) {
    const size_t nq = (size_t)n_out / 4;
    std::vector<int> qlo1(nq, 0), qlo2(nq, 0);
    std::vector<int> qdesc(nq, 0), slow;
    size_t fit = 0;
    for (size_t q = 0; q < nq; ++q) {
      const int* v = &nn[4 * q];
      bool ok = true;
      for (int k = 0; k < 4; ++k) ok = ok && grp_of[4 * q + k] < 0;
      int lo1 = -1, lo2 = -1;
      for (int k = 0; k < 4 && ok; ++k)
        if (v[k] >= 0 && (lo1 < 0 || v[k] < lo1)) lo1 = v[k];
      for (int k = 0; k < 4 && ok; ++k)
        if (v[k] >= 0 && v[k] - lo1 > 7 && (lo2 < 0 || v[k] < lo2)) lo2 = v[k];
      for (int k = 0; k < 4 && ok; ++k)
        if (v[k] >= 0 && v[k] - lo1 > 7 && v[k] - lo2 > 7) ok = false;
      if (!ok) { qdesc[q] = (int)0x80000000u; slow.push_back((int)q); continue; }
      if (lo1 < 0) lo1 = 0;
      if (lo2 < 0) lo2 = lo1;
      const int w1 = (int)std::min<int64_t>(lo1, n_src - 8), w2 = (int)std::min<int64_t>(lo2, n_src - 8);   // windows end inside the frame
      uint32_t d = 0;
      for (int k = 0; k < 4; ++k) {
        if (v[k] < 0) continue;
        const bool second = v[k] - lo1 > 7;
        d |= ((uint32_t)(v[k] - (second ? w2 : w1)) | (second ? 8u : 0u) | 16u) << (5 * k);
      }
      qlo1[q] = w1; qlo2[q] = w2; qdesc[q] = (int)d;
      ++fit;
    }
    if (fit * 10 >= nq * 9) {
      e = up(&h->d_qlo1, qlo1);
      if (e == hipSuccess) e = up(&h->d_qlo2, qlo2);
      int* dd = nullptr;
      if (e == hipSuccess) e = up(&dd, qdesc);
      h->d_qdesc = reinterpret_cast<uint32_t*>(dd);
      if (e == hipSuccess) e = up(&h->d_slow_q, slow);
      h->n_slow = (int)slow.size();
    }
  }
  // mixed plan (round 6): a plan with group means, uint8 frames.  Every cell is a set of samples (its group in the reference's
  // order, else its nearest-neighbour byte, else nothing); the samples of a quad are covered greedily with 8-byte windows over
  // the FLAT source index; a quad fits with at most NW windows and at most 255 samples per cell.
  // (nearest-neighbour-only plans too: a group of one sample -- their float32 output goes through the same tiled kernel)
  if (e == hipSuccess && n_out % 4 == 0 && n_src >= 16 && !getenv("LSPIV_PROJECT_ONE_CELL") && !getenv("LSPIV_PROJECT_NO_MIX")) {
    const size_t nq = (size_t)n_out / 4;
    std::vector<int> need(nq, 0);
    std::vector<int> px;
    auto samples_of = [&](size_t o, const int** first, int* n) {
      const int g = grp_of[o];
      if (g >= 0) { *first = &members[(size_t)off[(size_t)g]]; *n = off[(size_t)g + 1] - off[(size_t)g]; }
      else if (nn[o] >= 0) { *first = &nn[o]; *n = 1; }
      else { *first = nullptr; *n = 0; }
    };
    auto windows_of = [&](size_t q, int* starts, int cap) -> int {     // greedy cover of the quad's sample set; returns the number of windows
      px.clear();
      for (int k = 0; k < 4; ++k) {
        const int* f; int n;
        samples_of(4 * q + k, &f, &n);
        px.insert(px.end(), f, f + n);
      }
      std::sort(px.begin(), px.end());
      int nw = 0;
      int64_t end = -1;
      for (int v : px) {
        if (v < end) continue;
        int64_t st = std::min<int64_t>(v, n_src - 8);
        // the kernel reads the three aligned dwords around a window in one 12-byte load: they must lie inside the frame
        if ((st & ~(int64_t)3) + 12 > n_src) st = std::min<int64_t>(st, (n_src - 12) & ~(int64_t)3);
        if (v >= st + 8) return cap + 1;                     // the frame's last bytes cannot be reached that way: the slow kernel's quad
        if (nw < cap) starts[nw] = (int)st;
        ++nw;
        end = st + 8;
      }
      return nw;
    };
    size_t over2 = 0, over4 = 0;
    int tmp[4];
    for (size_t q = 0; q < nq; ++q) {
      need[q] = windows_of(q, tmp, 4);
      over2 += need[q] > 2;
      over4 += need[q] > 4;
    }
    const int NW = over2 * 50 <= nq ? 2 : 4;                 // at most 2 % of the quads left to the slow kernel: two windows will do
    const size_t left = NW == 2 ? over2 : over4;
    if (getenv("LSPIV_PROJECT_DEBUG"))
      fprintf(stderr, "lspiv projection: %zu quads, %zu need more than two 8-byte windows, %zu more than four: %s\n", nq, over2, over4,
              left * 10 <= nq ? (NW == 2 ? "mixed plan, two windows" : "mixed plan, four windows") : "no mixed plan");
    if (left * 10 <= nq) {                                   // otherwise the geometry is too scattered for windows: the one-cell kernel
      const int CW = NW / 2;
      std::vector<int> mwin(nq * NW, 0), mslow;
      std::vector<int> mcell(nq * 4 * CW, 0);
      for (size_t q = 0; q < nq; ++q) {
        int st[4] = {0, 0, 0, 0};
        bool ok = need[q] <= NW;
        const int nw = ok ? windows_of(q, st, NW) : 0;
        uint32_t words[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int k = 0; k < 4 && ok; ++k) {
          const int* f; int n;
          samples_of(4 * q + k, &f, &n);
          if (n > 255) { ok = false; break; }            // the kernel's division-free quotient is checked for counts up to 255
          uint32_t masks = 0;
          for (int i = 0; i < n; ++i) {
            int wsel = -1;
            for (int j = 0; j < nw; ++j)
              if (f[i] >= st[j] && f[i] < st[j] + 8) { wsel = j; break; }
            if (wsel < 0) { ok = false; break; }
            masks |= 1u << (8 * wsel + (f[i] - st[wsel]));
          }
          const uint32_t cnt = (uint32_t)std::max(n, 1);     // a cell without samples: 0 / 1
          if (NW == 2) words[k] = masks | (cnt << 16);
          else { words[2 * k] = masks; words[2 * k + 1] = cnt; }
        }
        if (!ok) {
          mwin[q * NW] = -1;
          mslow.push_back((int)q);
          continue;
        }
        for (int j = 0; j < NW; ++j) mwin[q * NW + j] = st[j < nw ? j : 0];    // unused windows repeat the first (masks 0)
        if (nw == 0) for (int j = 0; j < NW; ++j) mwin[q * NW + j] = 0;
        for (int j = 0; j < 4 * CW; ++j) mcell[q * 4 * CW + j] = (int)words[j];
      }
      e = up(&h->d_mwin, mwin);
      int* dc = nullptr;
      if (e == hipSuccess) e = up(&dc, mcell);
      h->d_mcell = reinterpret_cast<uint32_t*>(dc);
      if (e == hipSuccess) e = up(&h->d_mslow, mslow);
      h->n_mslow = (int)mslow.size();
      h->mix_nw = NW;
      // tiled form (project_tile_kernel): chunks of 8 bytes; the three dwords around a window lie in chunk c0 = (start & ~3) >> 3 and
      // c0 + 1, neighbours in the wave's sorted list; a window becomes its byte offset into the wave's tile.
      if (e == hipSuccess && n_src % 8 == 0 && !getenv("LSPIV_PROJECT_NO_TILE")) {
        std::vector<int> qch(nq * NW, -1);                   // per quad and window: c0, or -1 (window unused / quad not served)
        for (size_t q = 0; q < nq; ++q) {
          if (mwin[q * NW] < 0) continue;
          for (int j = 0; j < NW; ++j) {
            uint32_t m = 0;                                  // window j's masks of the four cells: none set = the window is not read
            for (int k = 0; k < 4; ++k) m |= ((uint32_t)mcell[q * 4 * CW + (size_t)k * CW] >> (8 * j)) & 0xffu;
            if (m) qch[q * NW + j] = (mwin[q * NW + j] & ~3) >> 3;
          }
        }
        auto quad_chunks = [&](size_t q, std::vector<int>& l) {
          if (mwin[q * NW] < 0) return false;
          for (int j = 0; j < NW; ++j)
            if (qch[q * NW + j] >= 0) { l.push_back(qch[q * NW + j]); l.push_back(qch[q * NW + j] + 1); }
          return true;
        };
        TileShape sh;
        int cap_limit = 0;
        if (tile_pick_shape(dst_h, dst_w, nq, quad_chunks, "uint8", &sh, &cap_limit)) {
          const int cap = 64 * sh.rmax;
          size_t failed = 0;
          std::vector<int> lst, wchunk((size_t)sh.n_waves() * cap, -1), twin(mwin.size(), 0), tslow(mslow);
          for (size_t q = 0; q < nq; ++q) if (mwin[q * NW] < 0) twin[q * NW] = -1;
          for (int64_t wv = 0; wv < sh.n_waves(); ++wv) {
            if (!tile_wave_list(sh, wv, quad_chunks, lst)) continue;      // (all -1: the wave returns at once)
            const int64_t ty = wv / sh.tiles_x(), tx = wv % sh.tiles_x();
            const bool fits = (int)lst.size() <= std::min(cap, cap_limit);
            failed += !fits;
            if (fits) {
              if (lst.empty()) lst.push_back(0);             // cells without a source only: the wave still runs and writes their zeros
              for (int i = 0; i < cap; ++i) wchunk[(size_t)wv * cap + i] = lst[std::min<size_t>((size_t)i, lst.size() - 1)];
            }
            for (int64_t r = ty * sh.bqy(); r < std::min(sh.rows, (ty + 1) * sh.bqy()); ++r)
              for (int64_t c = tx * sh.bqx(); c < std::min(sh.wq, (tx + 1) * sh.bqx()); ++c) {
                const size_t q = (size_t)(r * sh.wq + c);
                if (mwin[q * NW] < 0) continue;
                if (!fits) { twin[q * NW] = -1; tslow.push_back((int)q); continue; }
                for (int j = 0; j < NW; ++j) {
                  const int c0 = qch[q * NW + j];
                  if (c0 < 0) continue;                       // (offset 0: any resident bytes do under a zero mask)
                  const int pos = (int)(std::lower_bound(lst.begin(), lst.end(), c0) - lst.begin());
                  twin[q * NW + j] = 8 * pos + (mwin[q * NW + j] - 8 * c0);
                }
              }
          }
          if (getenv("LSPIV_PROJECT_DEBUG"))
            fprintf(stderr, "lspiv projection (uint8): tiles of %lld x %lld quads, %d list row(s), %zu waves to the slow kernel (%zu slow quads in all)\n",
                    (long long)sh.bqx(), (long long)sh.bqy(), sh.rmax, failed, tslow.size());
          e = up(&h->d_wchunk, wchunk);
          if (e == hipSuccess) e = up(&h->d_twin, twin);
          if (e == hipSuccess) e = up(&h->d_tslow, tslow);
          h->n_tslow = (int)tslow.size();
          h->tile_rmax = sh.rmax; h->tile_lg = sh.lg; h->tile_wq = (int)sh.wq; h->tile_rows = (int)sh.rows;
        }
      }
    }
  }
  // float32 frames in tiles (project_tile_f32_kernel): chunks of four pixels; a cell = the tile positions of its samples in the
  // reference's order (the group's members, or the one nearest neighbour), at most 6 (two descriptor words per cell) or 9 (three).
  if (e == hipSuccess && n_out % 4 == 0 && n_src % 4 == 0 && !getenv("LSPIV_PROJECT_ONE_CELL") && !getenv("LSPIV_PROJECT_NO_TILE")) {
    const size_t nq = (size_t)n_out / 4;
    auto count_of = [&](size_t o) { const int g = grp_of[o]; return g >= 0 ? off[(size_t)g + 1] - off[(size_t)g] : nn[o] >= 0 ? 1 : 0; };
    size_t over6 = 0, over9 = 0;
    for (size_t q = 0; q < nq; ++q) {
      int m = 0;
      for (int k = 0; k < 4; ++k) m = std::max(m, count_of(4 * q + k));
      over6 += m > 6; over9 += m > 9;
    }
    const int DW = over6 * 100 <= nq ? 2 : 3, maxs = 3 * DW;
    if (getenv("LSPIV_PROJECT_DEBUG"))
      fprintf(stderr, "lspiv projection (float32): %zu quads, %zu with a cell of more than 6 samples, %zu of more than 9: %s\n", nq, over6, over9,
              (DW == 2 ? over6 : over9) * 10 <= nq ? (DW == 2 ? "two descriptor words per cell" : "three descriptor words per cell") : "no tiles");
    if ((DW == 2 ? over6 : over9) * 10 <= nq) {
      auto served = [&](size_t q) {
        for (int k = 0; k < 4; ++k) if (count_of(4 * q + k) > maxs) return false;
        return true;
      };
      auto samples_of = [&](size_t o, const int** first, int* n) {
        const int g = grp_of[o];
        if (g >= 0) { *first = &members[(size_t)off[(size_t)g]]; *n = off[(size_t)g + 1] - off[(size_t)g]; }
        else if (nn[o] >= 0) { *first = &nn[o]; *n = 1; }
        else { *first = nullptr; *n = 0; }
      };
      auto quad_chunks = [&](size_t q, std::vector<int>& l) {
        if (!served(q)) return false;
        for (int k = 0; k < 4; ++k) {
          const int* f; int n;
          samples_of(4 * q + k, &f, &n);
          for (int i = 0; i < n; ++i) l.push_back(f[i] >> 2);
        }
        return true;
      };
      TileShape sh;
      int cap_limit = 0;
      if (tile_pick_shape(dst_h, dst_w, nq, quad_chunks, "float32", &sh, &cap_limit)) {
        const int cap = 64 * sh.rmax;
        size_t failed = 0;
        std::vector<int> lst, wchunk((size_t)sh.n_waves() * cap, -1), fdesc(nq * 4 * DW, 0), fslow;
        for (size_t q = 0; q < nq; ++q) if (!served(q)) { fdesc[q * 4 * DW] = -1; fslow.push_back((int)q); }
        for (int64_t wv = 0; wv < sh.n_waves(); ++wv) {
          if (!tile_wave_list(sh, wv, quad_chunks, lst)) continue;
          const int64_t ty = wv / sh.tiles_x(), tx = wv % sh.tiles_x();
          const bool fits = (int)lst.size() <= std::min(cap, cap_limit);
          failed += !fits;
          if (fits) {
            if (lst.empty()) lst.push_back(0);
            for (int i = 0; i < cap; ++i) wchunk[(size_t)wv * cap + i] = lst[std::min<size_t>((size_t)i, lst.size() - 1)];
          }
          for (int64_t r = ty * sh.bqy(); r < std::min(sh.rows, (ty + 1) * sh.bqy()); ++r)
            for (int64_t c = tx * sh.bqx(); c < std::min(sh.wq, (tx + 1) * sh.bqx()); ++c) {
              const size_t q = (size_t)(r * sh.wq + c);
              if (!served(q)) continue;
              if (!fits) { fdesc[q * 4 * DW] = -1; fslow.push_back((int)q); continue; }
              for (int k = 0; k < 4; ++k) {
                const int* f; int n;
                samples_of(4 * q + k, &f, &n);
                uint32_t w[3] = {0, 0, 0};
                for (int i = 0; i < n; ++i) {
                  const int pos = 4 * (int)(std::lower_bound(lst.begin(), lst.end(), f[i] >> 2) - lst.begin()) + (f[i] & 3);
                  w[i / 3] |= (uint32_t)pos << (10 * (i % 3));
                }
                const uint32_t grp = grp_of[4 * q + k] >= 0;
                w[0] |= ((uint32_t)n & 3u) << 30;
                if (DW == 2) w[1] |= (((uint32_t)n >> 2) & 1u) << 30 | grp << 31;
                else { w[1] |= (((uint32_t)n >> 2) & 3u) << 30; w[2] |= grp << 30; }
                for (int j = 0; j < DW; ++j) fdesc[(q * 4 + (size_t)k) * DW + j] = (int)w[j];
              }
            }
        }
        if (getenv("LSPIV_PROJECT_DEBUG"))
          fprintf(stderr, "lspiv projection (float32): tiles of %lld x %lld quads, %d list row(s), %zu waves to the slow kernel (%zu slow quads in all)\n",
                  (long long)sh.bqx(), (long long)sh.bqy(), sh.rmax, failed, fslow.size());
        e = up(&h->d_fchunk, wchunk);
        int* dd = nullptr;
        if (e == hipSuccess) e = up(&dd, fdesc);
        h->d_fdesc = reinterpret_cast<uint32_t*>(dd);
        if (e == hipSuccess) e = up(&h->d_fslow, fslow);
        h->n_fslow = (int)fslow.size();
        h->f_dw = DW; h->f_rmax = sh.rmax; h->f_lg = sh.lg; h->f_wq = (int)sh.wq; h->f_rows = (int)sh.rows;
      }
    }
  }
  if (e != hipSuccess) {
    lspiv_projection_destroy(h);
    return fail(e == hipErrorOutOfMemory ? LSPIV_ENOMEM : LSPIV_EHIP, "projection plan upload: %s", hipGetErrorString(e));
  }
  *handle = h;
  return LSPIV_OK;
}

int lspiv_project_frames_dev(lspiv_projection* h, const void* d_frames, int dtype, int64_t T, float* d_out, void* stream) {
  if (!h || !d_frames || !d_out) return fail(LSPIV_EINVAL, "NULL argument");
  if (dtype < 0 || dtype > 2) return fail(LSPIV_EINVAL, "dtype %d not in {0:u8, 1:f32, 2:f64}", dtype);
  if (T < 0 || T >= (int64_t)1 << 28) return fail(LSPIV_ESHAPE, "bad frame count");
  DeviceCtx* c;
  int rc = get_ctx(&c);
  if (rc) return rc;
  hipStream_t s = stream ? (hipStream_t)stream : c->stream;
  const bool tile = dtype == 0 && h->d_mcell && h->d_twin && (reinterpret_cast<uintptr_t>(d_out) & 15) == 0 && (reinterpret_cast<uintptr_t>(d_frames) & 7) == 0;
  const bool win = dtype == 0 && !tile && h->d_qdesc && (reinterpret_cast<uintptr_t>(d_out) & 15) == 0;
  const bool mix = dtype == 0 && !tile && !win && h->d_mcell && (reinterpret_cast<uintptr_t>(d_out) & 15) == 0;
  const bool tile_f = dtype == 1 && h->d_fdesc && (reinterpret_cast<uintptr_t>(d_out) & 15) == 0 && (reinterpret_cast<uintptr_t>(d_frames) & 15) == 0;
  hipError_t e = tile_f ? lspiv::launch_project_tile_f32((const float*)d_frames, h->src_h * h->src_w, (int)T, h->f_dw, h->f_rmax, h->d_fchunk, h->d_fdesc,
                                                          h->f_wq, h->f_rows, h->f_lg, h->d_fslow, h->n_fslow, h->d_nn, h->d_grp_of, h->d_grp_off, h->d_grp_src,
                                                          d_out, (int)(h->dst_h * h->dst_w), s)
               : tile ? lspiv::launch_project_tile((const uint8_t*)d_frames, h->src_h * h->src_w, (int)T, h->mix_nw, h->tile_rmax, h->d_wchunk, h->d_twin,
                                                    h->d_mcell, h->tile_wq, h->tile_rows, h->tile_lg, h->d_tslow, h->n_tslow, h->d_nn, h->d_grp_of, h->d_grp_off, h->d_grp_src, d_out,
                                                    (int)(h->dst_h * h->dst_w), s)
               : mix ? lspiv::launch_project_mix((const uint8_t*)d_frames, h->src_h * h->src_w, (int)T, h->mix_nw, h->d_mwin, h->d_mcell, h->d_mslow,
                                                  h->n_mslow, h->d_nn, h->d_grp_of, h->d_grp_off, h->d_grp_src, d_out, (int)(h->dst_h * h->dst_w), s)
               : win ? lspiv::launch_project_win((const uint8_t*)d_frames, h->src_h * h->src_w, (int)T, h->d_qlo1, h->d_qlo2, h->d_qdesc,
                                                  h->d_slow_q, h->n_slow, h->d_nn, h->d_grp_of, h->d_grp_off, h->d_grp_src, d_out, (int)(h->dst_h * h->dst_w), s)
                     : lspiv::launch_project(d_frames, dtype, h->src_h * h->src_w, (int)T, h->d_nn, h->d_grp_of, h->d_grp_off,
                                             h->d_grp_src, d_out, (int)(h->dst_h * h->dst_w), s);
  if (e != hipSuccess) return fail(LSPIV_EHIP, "kernel launch failed: %s", hipGetErrorString(e));
  return LSPIV_OK;
}

int lspiv_project_frames(lspiv_projection* h, const void* frames, int dtype, int64_t T, float* out) {
  if (!h || !frames || !out) return fail(LSPIV_EINVAL, "NULL argument");
  if (dtype < 0 || dtype > 2) return fail(LSPIV_EINVAL, "dtype %d not in {0:u8, 1:f32, 2:f64}", dtype);
  if (T <= 0) return LSPIV_OK;
  const size_t ib = (size_t)T * h->src_h * h->src_w * elem_size(dtype);
  const size_t ob = (size_t)T * h->dst_h * h->dst_w * sizeof(float);
  return project_host(ib, ob, frames, out, [&](void* d_in, void* d_out, hipStream_t s) {
    return lspiv_project_frames_dev(h, d_in, dtype, T, (float*)d_out, s);
  });
}

int lspiv_project_frames_u8_dev(lspiv_projection* h, const uint8_t* d_frames, int64_t T, uint8_t* d_out, void* stream) {
  if (!h || !d_frames || !d_out) return fail(LSPIV_EINVAL, "NULL argument");
  if (h->n_groups != 0)
    return fail(LSPIV_EINVAL, "the plan averages %lld cells (reducer \"mean\"): their values are no bytes, use lspiv_project_frames",
                (long long)h->n_groups);
  if (T < 0 || T >= (int64_t)1 << 28) return fail(LSPIV_ESHAPE, "bad frame count");
  DeviceCtx* c;
  int rc = get_ctx(&c);
  if (rc) return rc;
  hipStream_t s = stream ? (hipStream_t)stream : c->stream;
  const bool tile = h->d_mcell && h->d_twin && (reinterpret_cast<uintptr_t>(d_out) & 3) == 0 && (reinterpret_cast<uintptr_t>(d_frames) & 7) == 0;
  hipError_t e = tile ? lspiv::launch_project_tile_u8(d_frames, h->src_h * h->src_w, (int)T, h->mix_nw, h->tile_rmax, h->d_wchunk, h->d_twin, h->d_mcell,
                                                       h->tile_wq, h->tile_rows, h->tile_lg, h->d_tslow, h->n_tslow, h->d_nn, d_out,
                                                       (int)(h->dst_h * h->dst_w), s)
                      : lspiv::launch_project_u8(d_frames, h->src_h * h->src_w, (int)T, h->d_qlo1, h->d_qlo2, h->d_qdesc, h->d_nn, d_out,
                                                 (int)(h->dst_h * h->dst_w), s);
  if (e != hipSuccess) return fail(LSPIV_EHIP, "kernel launch failed: %s", hipGetErrorString(e));
  return LSPIV_OK;
}

int lspiv_project_frames_u8(lspiv_projection* h, const uint8_t* frames, int64_t T, uint8_t* out) {
  if (!h || !frames || !out) return fail(LSPIV_EINVAL, "NULL argument");
  if (T <= 0) return LSPIV_OK;
  const size_t ib = (size_t)T * h->src_h * h->src_w, ob = (size_t)T * h->dst_h * h->dst_w;
  return project_host(ib, ob, frames, out, [&](void* d_in, void* d_out, hipStream_t s) {
    return lspiv_project_frames_u8_dev(h, (const uint8_t*)d_in, T, (uint8_t*)d_out, s);
  });
}

int lspiv_projection_destroy(lspiv_projection* h) {
  if (!h) return LSPIV_OK;
  if (h->d_nn) hipFree(h->d_nn);
  if (h->d_grp_of) hipFree(h->d_grp_of);
  if (h->d_grp_off) hipFree(h->d_grp_off);
  if (h->d_grp_src) hipFree(h->d_grp_src);
  if (h->d_qlo1) hipFree(h->d_qlo1);
  if (h->d_qlo2) hipFree(h->d_qlo2);
  if (h->d_qdesc) hipFree(h->d_qdesc);
  if (h->d_slow_q) hipFree(h->d_slow_q);
  if (h->d_mwin) hipFree(h->d_mwin);
  if (h->d_mcell) hipFree(h->d_mcell);
  if (h->d_mslow) hipFree(h->d_mslow);
  if (h->d_wchunk) hipFree(h->d_wchunk);
  if (h->d_twin) hipFree(h->d_twin);
  if (h->d_tslow) hipFree(h->d_tslow);
  if (h->d_fchunk) hipFree(h->d_fchunk);
  if (h->d_fdesc) hipFree(h->d_fdesc);
  if (h->d_fslow) hipFree(h->d_fslow);
  delete h;
  return LSPIV_OK;
}

// ---- project_cv (N1, method="cv"): cv2.undistort + cv2.warpPerspective restated as two fixed-point bilinear remaps ---
struct lspiv_remap {
  int64_t src_h, src_w, dst_h, dst_w;
  bool undistort;
  int *d_mx1, *d_my1, *d_mx2, *d_my2;      // integer source coordinates of the undistortion map / of the warp map
  uint16_t *d_mf1, *d_mf2;                 // 1/32-pixel fraction index fy * 32 + fx
  void* d_tmp; size_t tmp_cap;             // undistorted frames of one call (grow-only)
  // quad plans for uint8 frames (project.hip, remap_win_kernel), one per remap; nullptr: not built
  int *d_qb1, *d_qb2, *d_slow1, *d_slow2;
  uint64_t *d_qd1, *d_qd2;
  int n_slow1, n_slow2;
  // both remaps in one kernel for uint8 frames (project.hip, remap_fused_kernel): tiles' boxes, destination pixel descriptors, the
  // undistortion map's quads with the row step; nullptr: not built (the two passes run)
  void* d_ft; uint32_t* d_fpx; int* d_fqb; uint64_t* d_fqd;
  int f_tiles, f_tiles_x, f_cap;
  std::mutex host_mu;                      // host-pointer calls on ONE handle queue: they share d_tmp (two handles run side by side)
};

namespace {
// saturate_cast<int>(double): round half to even, saturating
int64_t cv_round(double v) {
  if (!(v == v)) return 0;
  if (v >= 2147483647.0) return 2147483647;
  if (v <= -2147483648.0) return -2147483648LL;
  return (int64_t)std::nearbyint(v);
}
bool invert3(const double* m, double* o) {
  const double a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
  const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
  if (det == 0.0 || !(det == det)) return false;
  const double r = 1.0 / det;
  o[0] = (e * i - f * h) * r; o[1] = (c * h - b * i) * r; o[2] = (b * f - c * e) * r;
  o[3] = (f * g - d * i) * r; o[4] = (a * i - c * g) * r; o[5] = (c * d - a * f) * r;
  o[6] = (d * h - e * g) * r; o[7] = (b * g - a * h) * r; o[8] = (a * e - b * d) * r;
  return true;
}
int upload_map(const std::vector<int>& mx, const std::vector<int>& my, const std::vector<uint16_t>& mf, int** d_mx, int** d_my,
               uint16_t** d_mf) {
  const size_t n = mx.size();
  void* p = nullptr;
  HIP_TRY(hipMalloc(&p, n * sizeof(int))); *d_mx = (int*)p;
  HIP_TRY(hipMalloc(&p, n * sizeof(int))); *d_my = (int*)p;
  HIP_TRY(hipMalloc(&p, n * sizeof(uint16_t))); *d_mf = (uint16_t*)p;
  HIP_TRY(hipMemcpy(*d_mx, mx.data(), n * sizeof(int), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(*d_my, my.data(), n * sizeof(int), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(*d_mf, mf.data(), n * sizeof(uint16_t), hipMemcpyHostToDevice));
  return LSPIV_OK;
}
int clamp_short(int64_t v) { return (int)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); }   // OpenCV keeps the integer part as short

// Quad plan of a remap for uint8 frames: four consecutive destination pixels whose 2 x 2 source neighbourhoods are all
// interior, share the source row pair and fit 8 bytes (max ix - min ix <= 6) read two 8-byte windows instead of eight 2-byte
// pairs.  Built when the destination has whole quads and >= 80 % of them qualify; the others are listed for the per-pixel
// kernel.  LSPIV_PROJECT_ONE_CELL=1 skips it (A/B).
int build_remap_quads(const std::vector<int>& mx, const std::vector<int>& my, const std::vector<uint16_t>& mf, int64_t Hs, int64_t Ws,
                      int** d_qb, uint64_t** d_qd, int** d_slow, int* n_slow) {
  const size_t n = mx.size();
  if (n % 4 != 0 || Hs * Ws < 16 || getenv("LSPIV_PROJECT_ONE_CELL")) return LSPIV_OK;
  const size_t nq = n / 4;
  std::vector<int> qb(nq, 0), slow;
  std::vector<uint64_t> qd(nq, 0);
  for (size_t q = 0; q < nq; ++q) {
    const size_t o = 4 * q;
    bool ok = true, outside = true;
    int lo = mx[o], hi = mx[o];
    for (int k = 0; k < 4; ++k) {
      const int ix = mx[o + k], iy = my[o + k];
      ok = ok && ix >= 0 && ix + 1 < Ws && iy >= 0 && iy + 1 < Hs && iy == my[o];
      outside = outside && !((ix >= -1 && ix < Ws) && (iy >= -1 && iy < Hs));   // no neighbour of the 2 x 2 patch inside
      lo = std::min(lo, ix); hi = std::max(hi, ix);
    }
    if (outside) { qd[q] = (uint64_t)1 << 62; continue; }
    const int64_t base = ok ? (int64_t)my[o] * Ws + lo : 0;
    ok = ok && hi - lo <= 6 && base + Ws + 8 <= Hs * Ws;     // both windows end inside the frame
    if (!ok) { qd[q] = (uint64_t)1 << 63; slow.push_back((int)q); continue; }
    uint64_t d = 0;
    for (int k = 0; k < 4; ++k) {
      const uint32_t fr = mf[o + k], fx = fr & 31u, fy = fr >> 5;
      d |= (uint64_t)((uint32_t)(mx[o + k] - lo) | (fx << 3) | (fy << 8)) << (16 * k);
    }
    qb[q] = (int)base; qd[q] = d;
  }
  if (slow.size() * 5 > nq) return LSPIV_OK;                  // fewer than 80 % fit: the per-pixel kernel does everything
  void* p = nullptr;
  HIP_TRY(hipMalloc(&p, nq * sizeof(int))); *d_qb = (int*)p;
  HIP_TRY(hipMalloc(&p, nq * sizeof(uint64_t))); *d_qd = (uint64_t*)p;
  HIP_TRY(hipMalloc(&p, std::max<size_t>(slow.size(), 1) * sizeof(int))); *d_slow = (int*)p;
  HIP_TRY(hipMemcpy(*d_qb, qb.data(), nq * sizeof(int), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(*d_qd, qd.data(), nq * sizeof(uint64_t), hipMemcpyHostToDevice));
  if (!slow.empty()) HIP_TRY(hipMemcpy(*d_slow, slow.data(), slow.size() * sizeof(int), hipMemcpyHostToDevice));
  *n_slow = (int)slow.size();
  return LSPIV_OK;
}

// Plan of remap_fused_kernel (project.hip): undistortion and warp of uint8 frames in one kernel.  Built when both widths are multiples of
// four and every 64 x 16 destination tile's box of undistorted pixels fits 16 000 bytes; LSPIV_PROJECT_CV_TWO_PASS=1 skips it (A/B, tests).
int build_remap_fused(lspiv_remap* h, const std::vector<int>& mx1, const std::vector<int>& my1, const std::vector<uint16_t>& mf1,
                      const std::vector<int>& mx2, const std::vector<int>& my2, const std::vector<uint16_t>& mf2) {
  const int64_t Hs = h->src_h, Ws = h->src_w, Hd = h->dst_h, Wd = h->dst_w;
  if (Ws % 4 != 0 || Wd % 4 != 0 || Hs * Ws < 64 || Hs > 32767 || Ws > 32767 || Hd > 32767 || Wd > 32767 ||   // (24-bit index arithmetic in the kernel)
      getenv("LSPIV_PROJECT_CV_TWO_PASS")) return LSPIV_OK;
  // the undistortion map by quads: two or three 8-byte windows of the camera frame per four undistorted pixels
  const size_t nq = (size_t)(Hs * Ws / 4);
  std::vector<int> qb(nq, 0);
  std::vector<uint64_t> qd(nq, 0);
  size_t n_slow = 0;
  for (size_t q = 0; q < nq; ++q) {
    const size_t o = 4 * q;
    bool ok = true, outside = true;
    int lo = mx1[o], hi = mx1[o], ylo = my1[o], yhi = my1[o];
    for (int k = 0; k < 4; ++k) {
      const int ix = mx1[o + k], iy = my1[o + k];
      ok = ok && ix >= 0 && ix + 1 < Ws && iy >= 0 && iy + 1 < Hs;
      outside = outside && !((ix >= -1 && ix < Ws) && (iy >= -1 && iy < Hs));
      lo = std::min(lo, ix); hi = std::max(hi, ix); ylo = std::min(ylo, iy); yhi = std::max(yhi, iy);
    }
    if (outside) { qd[q] = (uint64_t)1 << 62; continue; }
    const int64_t base = ok ? (int64_t)ylo * Ws + lo : 0;
    ok = ok && hi - lo <= 6 && yhi - ylo <= 1 && (base & ~(int64_t)3) + (int64_t)(yhi - ylo + 1) * Ws + 12 <= Hs * Ws;   // every 12-byte load (from the dword below the window) ends inside the frame
    if (!ok) { qd[q] = (uint64_t)1 << 63; ++n_slow; continue; }
    uint64_t d = yhi > ylo ? (uint64_t)1 << 15 : 0;
    for (int k = 0; k < 4; ++k) {
      const uint32_t fr = mf1[o + k], fx = fr & 31u, fy = fr >> 5;
      d |= (uint64_t)((uint32_t)(mx1[o + k] - lo) | (fx << 3) | (fy << 8) | ((uint32_t)(my1[o + k] - ylo) << 13)) << (16 * k);
    }
    qb[q] = (int)base; qd[q] = d;
  }
  if (n_slow * 5 > nq) return LSPIV_OK;                       // a map this folded: the per-pixel path would pace every wave
  // the warp by tiles
  constexpr int TW = 64, TH = 16;
  const int tiles_x = (int)((Wd + TW - 1) / TW), tiles_y = (int)((Hd + TH - 1) / TH);
  std::vector<int> tiles((size_t)tiles_x * tiles_y * 4, 0);
  std::vector<uint32_t> pxd((size_t)(Hd * Wd), 0x80000000u);
  int cap = 16;
  auto inside = [&](int ix, int iy) { return (ix >= -1 && ix < Ws) && (iy >= -1 && iy < Hs); };
  for (int ty = 0; ty < tiles_y; ++ty)
    for (int tx = 0; tx < tiles_x; ++tx) {
      int xlo = INT_MAX, xhi = INT_MIN, ylo = INT_MAX, yhi = INT_MIN;
      const int64_t y1 = std::min<int64_t>(Hd, (int64_t)(ty + 1) * TH), x1 = std::min<int64_t>(Wd, (int64_t)(tx + 1) * TW);
      for (int64_t y = (int64_t)ty * TH; y < y1; ++y)
        for (int64_t x = (int64_t)tx * TW; x < x1; ++x) {
          const int ix = mx2[(size_t)(y * Wd + x)], iy = my2[(size_t)(y * Wd + x)];
          if (!inside(ix, iy)) continue;
          xlo = std::min(xlo, ix); xhi = std::max(xhi, ix + 1); ylo = std::min(ylo, iy); yhi = std::max(yhi, iy + 1);
        }
      if (xlo > xhi) continue;                                // nothing of the tile inside the image: an empty box
      const int bx0 = xlo >= 0 ? xlo / 4 * 4 : -4, bw = (xhi - bx0 + 1 + 3) / 4 * 4, bh = yhi - ylo + 1;
      if ((int64_t)bw * bh > 16000) return LSPIV_OK;          // this tile reads too wide a piece of the image (four boxes + 16 bytes must fit 64 KB of LDS): two passes
      cap = std::max(cap, bw * bh);
      int* t = &tiles[((size_t)ty * tiles_x + tx) * 4];
      t[0] = bx0; t[1] = ylo; t[2] = bw; t[3] = bh;
      for (int64_t y = (int64_t)ty * TH; y < y1; ++y)
        for (int64_t x = (int64_t)tx * TW; x < x1; ++x) {
          const size_t o = (size_t)(y * Wd + x);
          const int ix = mx2[o], iy = my2[o];
          if (!inside(ix, iy)) continue;
          const uint32_t fr = mf2[o], fx = fr & 31u, fy = fr >> 5;
          pxd[o] = (uint32_t)((iy - ylo) * bw + (ix - bx0)) | (fx << 16) | (fy << 21);
        }
    }
  cap = (cap + 15) / 16 * 16;
  if (getenv("LSPIV_PROJECT_DEBUG"))
    fprintf(stderr, "lspiv project_cv fused plan: %d x %d tiles, largest box %d bytes, %zu of %zu undistortion quads per pixel\n", tiles_x,
            tiles_y, cap, n_slow, nq);
  void* p = nullptr;
  HIP_TRY(hipMalloc(&p, nq * sizeof(int))); h->d_fqb = (int*)p;
  HIP_TRY(hipMalloc(&p, nq * sizeof(uint64_t))); h->d_fqd = (uint64_t*)p;
  HIP_TRY(hipMalloc(&p, pxd.size() * sizeof(uint32_t))); h->d_fpx = (uint32_t*)p;
  HIP_TRY(hipMalloc(&p, tiles.size() * sizeof(int))); h->d_ft = p;
  HIP_TRY(hipMemcpy(h->d_fqb, qb.data(), nq * sizeof(int), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(h->d_fqd, qd.data(), nq * sizeof(uint64_t), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(h->d_fpx, pxd.data(), pxd.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(h->d_ft, tiles.data(), tiles.size() * sizeof(int), hipMemcpyHostToDevice));
  h->f_tiles = tiles_x * tiles_y; h->f_tiles_x = tiles_x; h->f_cap = cap;
  return LSPIV_OK;
}
}  // namespace

int lspiv_project_cv_create(int64_t src_h, int64_t src_w, int64_t dst_h, int64_t dst_w, const double* camera_matrix,
                            const double* dist_coeffs, int n_dist, const double* M, lspiv_remap** handle) {
  if (!handle || !M) return fail(LSPIV_EINVAL, "NULL argument");
  if (src_h <= 0 || src_w <= 0 || dst_h <= 0 || dst_w <= 0 || src_h * src_w >= (int64_t)1 << 31 || dst_h * dst_w >= (int64_t)1 << 31)
    return fail(LSPIV_ESHAPE, "bad projection shape");
  if (n_dist != 0 && n_dist != 4 && n_dist != 5 && n_dist != 8)
    return fail(LSPIV_EINVAL, "dist_coeffs must hold 0, 4, 5 or 8 values (k1 k2 p1 p2 [k3 [k4 k5 k6]]), got %d", n_dist);
  if (n_dist > 0 && (!dist_coeffs || !camera_matrix)) return fail(LSPIV_EINVAL, "distortion coefficients need a camera matrix");
  DeviceCtx* c;
  int rc = get_ctx(&c);
  if (rc) return rc;
  lspiv_remap* h = new lspiv_remap();
  memset(h, 0, sizeof(*h));
  h->src_h = src_h; h->src_w = src_w; h->dst_h = dst_h; h->dst_w = dst_w;
  h->undistort = camera_matrix != nullptr;
  std::vector<int> mx1, my1;                 // the undistortion map, kept for the fused plan
  std::vector<uint16_t> mf1;
  if (h->undistort) {
    // initUndistortRectifyMap(K, dist, R = I, newK = K, size, CV_16SC2): the row walk _x += ir[0] included
    double ir[9];
    if (!invert3(camera_matrix, ir)) { delete h; return fail(LSPIV_EINVAL, "camera matrix is singular"); }
    double k[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n_dist; ++i) k[i] = dist_coeffs[i];
    const double k1 = k[0], k2 = k[1], p1 = k[2], p2 = k[3], k3 = k[4], k4 = k[5], k5 = k[6], k6 = k[7];
    const double fx = camera_matrix[0], fy = camera_matrix[4], u0 = camera_matrix[2], v0 = camera_matrix[5];
    const size_t n = (size_t)(src_h * src_w);
    std::vector<int> mx(n), my(n);
    std::vector<uint16_t> mf(n);
    for (int64_t i = 0; i < src_h; ++i) {
      double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
      for (int64_t j = 0; j < src_w; ++j, _x += ir[0], _y += ir[3], _w += ir[6]) {
        const double w = 1.0 / _w, x = _x * w, y = _y * w;
        const double x2 = x * x, y2 = y * y, r2 = x2 + y2, _2xy = 2 * x * y;
        const double kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2);
        const double xd = x * kr + p1 * _2xy + p2 * (r2 + 2 * x2), yd = y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy;
        const int64_t iu = cv_round((fx * xd + u0) * 32.0), iv = cv_round((fy * yd + v0) * 32.0);
        const size_t o = (size_t)(i * src_w + j);
        mx[o] = clamp_short(iu >> 5); my[o] = clamp_short(iv >> 5);
        mf[o] = (uint16_t)((iv & 31) * 32 + (iu & 31));
      }
    }
    rc = upload_map(mx, my, mf, &h->d_mx1, &h->d_my1, &h->d_mf1);
    if (!rc) rc = build_remap_quads(mx, my, mf, src_h, src_w, &h->d_qb1, &h->d_qd1, &h->d_slow1, &h->n_slow1);
    if (rc) { lspiv_project_cv_destroy(h); return rc; }
    mx1.swap(mx); my1.swap(my); mf1.swap(mf);
  }
  {
    // cv2.warpPerspective(src, M, (dst_w, dst_h), INTER_AREA -> INTER_LINEAR): M is inverted, 64-pixel column blocks
    double Mi[9];
    if (!invert3(M, Mi)) { lspiv_project_cv_destroy(h); return fail(LSPIV_EINVAL, "homography is singular"); }
    const size_t n = (size_t)(dst_h * dst_w);
    std::vector<int> mx(n), my(n);
    std::vector<uint16_t> mf(n);
    for (int64_t y = 0; y < dst_h; ++y)
      for (int64_t xb = 0; xb < dst_w; xb += 64) {
        const double X0 = Mi[0] * xb + Mi[1] * y + Mi[2], Y0 = Mi[3] * xb + Mi[4] * y + Mi[5], W0 = Mi[6] * xb + Mi[7] * y + Mi[8];
        for (int64_t x1 = 0; x1 < 64 && xb + x1 < dst_w; ++x1) {
          double W = W0 + Mi[6] * x1;
          W = W != 0.0 ? 32.0 / W : 0.0;
          const double fX = std::max(-2147483648.0, std::min(2147483647.0, (X0 + Mi[0] * x1) * W));
          const double fY = std::max(-2147483648.0, std::min(2147483647.0, (Y0 + Mi[3] * x1) * W));
          const int64_t X = cv_round(fX), Y = cv_round(fY);
          const size_t o = (size_t)(y * dst_w + xb + x1);
          mx[o] = clamp_short(X >> 5); my[o] = clamp_short(Y >> 5);
          mf[o] = (uint16_t)((Y & 31) * 32 + (X & 31));
        }
      }
    rc = upload_map(mx, my, mf, &h->d_mx2, &h->d_my2, &h->d_mf2);
    if (!rc) rc = build_remap_quads(mx, my, mf, src_h, src_w, &h->d_qb2, &h->d_qd2, &h->d_slow2, &h->n_slow2);
    if (!rc && h->undistort) rc = build_remap_fused(h, mx1, my1, mf1, mx, my, mf);
    if (rc) { lspiv_project_cv_destroy(h); return rc; }
  }
  *handle = h;
  return LSPIV_OK;
}

int lspiv_project_cv_frames_dev(lspiv_remap* h, const void* d_frames, int dtype, int64_t T, void* d_out, void* stream) {
  if (!h || !d_frames || !d_out) return fail(LSPIV_EINVAL, "NULL argument");
  if (dtype != LSPIV_U8 && dtype != LSPIV_F32) return fail(LSPIV_EINVAL, "project_cv takes uint8 or float32 frames (cv2 keeps the frame dtype)");
  if (T < 0 || T >= (int64_t)1 << 28) return fail(LSPIV_ESHAPE, "bad frame count");
  DeviceCtx* c;
  int rc = get_ctx(&c);
  if (rc) return rc;
  hipStream_t s = stream ? (hipStream_t)stream : c->stream;
  const int64_t n_src = h->src_h * h->src_w, n_dst = h->dst_h * h->dst_w;
  const void* src = d_frames;
  hipError_t e;
  if (dtype == LSPIV_U8 && h->undistort && h->d_ft && ((reinterpret_cast<uintptr_t>(d_out) | reinterpret_cast<uintptr_t>(d_frames)) & 3) == 0) {   // both remaps in one kernel
    e = lspiv::launch_remap_fused((const uint8_t*)d_frames, n_src, (int)h->src_h, (int)h->src_w, (int)T, h->d_ft, h->f_tiles, h->f_tiles_x,
                                  h->f_cap, h->d_fpx, h->d_fqb, h->d_fqd, h->d_mx1, h->d_my1, h->d_mf1, (uint8_t*)d_out, (int)h->dst_h,
                                  (int)h->dst_w, s);
    if (e != hipSuccess) return fail(LSPIV_EHIP, "kernel launch failed: %s", hipGetErrorString(e));
    return LSPIV_OK;
  }
  if (dtype == LSPIV_F32 && h->undistort && h->d_ft && ((reinterpret_cast<uintptr_t>(d_out) & 15) | (reinterpret_cast<uintptr_t>(d_frames) & 3)) == 0) {
    e = lspiv::launch_remap_fused_f32((const float*)d_frames, n_src, (int)h->src_h, (int)h->src_w, (int)T, h->d_ft, h->f_tiles, h->f_tiles_x,
                                      h->f_cap, h->d_fpx, h->d_mx1, h->d_my1, h->d_mf1, (float*)d_out, (int)h->dst_h, (int)h->dst_w, s);
    if (e != hipSuccess) return fail(LSPIV_EHIP, "kernel launch failed: %s", hipGetErrorString(e));
    return LSPIV_OK;
  }
  if (h->undistort) {
    rc = ensure(&h->d_tmp, &h->tmp_cap, (size_t)T * n_src * elem_size(dtype));
    if (rc) return rc;
    if (dtype == LSPIV_U8 && h->d_qd1)
      e = lspiv::launch_remap_win((const uint8_t*)d_frames, n_src, (int)h->src_h, (int)h->src_w, (int)T, h->d_qb1, h->d_qd1, h->d_slow1,
                                  h->n_slow1, h->d_mx1, h->d_my1, h->d_mf1, (uint8_t*)h->d_tmp, (int)n_src, s);
    else
      e = lspiv::launch_remap(d_frames, dtype, n_src, (int)h->src_h, (int)h->src_w, (int)T, h->d_mx1, h->d_my1, h->d_mf1, h->d_tmp, (int)n_src, s);
    if (e != hipSuccess) return fail(LSPIV_EHIP, "kernel launch failed: %s", hipGetErrorString(e));
    src = h->d_tmp;
  }
  if (dtype == LSPIV_U8 && h->d_qd2 && (reinterpret_cast<uintptr_t>(d_out) & 3) == 0)
    e = lspiv::launch_remap_win((const uint8_t*)src, n_src, (int)h->src_h, (int)h->src_w, (int)T, h->d_qb2, h->d_qd2, h->d_slow2, h->n_slow2,
                                h->d_mx2, h->d_my2, h->d_mf2, (uint8_t*)d_out, (int)n_dst, s);
  else
    e = lspiv::launch_remap(src, dtype, n_src, (int)h->src_h, (int)h->src_w, (int)T, h->d_mx2, h->d_my2, h->d_mf2, d_out, (int)n_dst, s);
  if (e != hipSuccess) return fail(LSPIV_EHIP, "kernel launch failed: %s", hipGetErrorString(e));
  return LSPIV_OK;
}

int lspiv_project_cv_frames(lspiv_remap* h, const void* frames, int dtype, int64_t T, void* out) {
  if (!h || !frames || !out) return fail(LSPIV_EINVAL, "NULL argument");
  if (dtype != LSPIV_U8 && dtype != LSPIV_F32) return fail(LSPIV_EINVAL, "project_cv takes uint8 or float32 frames (cv2 keeps the frame dtype)");
  if (T <= 0) return LSPIV_OK;
  const size_t fb = (size_t)T * h->src_h * h->src_w * elem_size(dtype), ob = (size_t)T * h->dst_h * h->dst_w * elem_size(dtype);
  std::lock_guard<std::mutex> handle_lock(h->host_mu);
  return project_host(fb, ob, frames, out, [&](void* d_in, void* d_out, hipStream_t s) {
    return lspiv_project_cv_frames_dev(h, d_in, dtype, T, d_out, s);
  });
}

int lspiv_project_cv_destroy(lspiv_remap* h) {
  if (!h) return LSPIV_OK;
  for (void* p : {(void*)h->d_mx1, (void*)h->d_my1, (void*)h->d_mf1, (void*)h->d_mx2, (void*)h->d_my2, (void*)h->d_mf2, h->d_tmp,
                  (void*)h->d_qb1, (void*)h->d_qb2, (void*)h->d_qd1, (void*)h->d_qd2, (void*)h->d_slow1, (void*)h->d_slow2,
                  h->d_ft, (void*)h->d_fpx, (void*)h->d_fqb, (void*)h->d_fqd})
    if (p) hipFree(p);
  delete h;
  return LSPIV_OK;
}

int lspiv_pack_int16_dev(const float* d_values, int64_t n, float scale, int fill, int16_t* d_packed, void* stream) {
  if (!d_values || !d_packed) return fail(LSPIV_EINVAL, "NULL argument");
  if (n < 0 || !(scale > 0.0f) || fill < -32768 || fill > 32767) return fail(LSPIV_EINVAL, "bad argument");
  DeviceCtx* c;
  int rc = get_ctx(&c);
  if (rc) return rc;
  hipError_t e = lspiv::launch_pack_int16(d_values, n, scale, fill, d_packed, stream ? (hipStream_t)stream : c->stream);
  if (e != hipSuccess) return fail(LSPIV_EHIP, "kernel launch failed: %s", hipGetErrorString(e));
  return LSPIV_OK;
}

int lspiv_pack_int16(const float* values, int64_t n, float scale, int fill, int16_t* packed) {
  std::lock_guard<std::mutex> host_lock(locks_here().host);
  if (!values || !packed) return fail(LSPIV_EINVAL, "NULL argument");
  if (n <= 0) return n == 0 ? LSPIV_OK : fail(LSPIV_EINVAL, "bad n");
  DeviceCtx* c;
  int rc = get_ctx(&c);
  if (rc) return rc;
  rc = ensure(&c->d_planes, &c->planes_cap, (size_t)n * sizeof(float));
  if (rc) return rc;
  rc = ensure(&c->d_out, &c->out_cap, (size_t)n * sizeof(int16_t));
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(c->d_planes, values, (size_t)n * sizeof(float), hipMemcpyHostToDevice, c->stream));
  rc = lspiv_pack_int16_dev(c->d_planes, n, scale, fill, (int16_t*)c->d_out, c->stream);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(packed, c->d_out, (size_t)n * sizeof(int16_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return LSPIV_OK;
}

// ---- post-PIV masks (N3) ------------------------------------------------------------------------------
namespace {
const int kMaskParams[10] = {2, 2, 1, 1, 1, 2, 2, 2, 5, 6};
bool mask_is_2d(int kind) { return kind == LSPIV_MASK_COUNT || kind == LSPIV_MASK_VARIANCE; }

int check_fields(const void* f, int64_t T, int64_t R, int64_t C) {
  if (!f) return fail(LSPIV_EINVAL, "NULL argument");
  if (T < 1 || R < 1 || C < 1 || R >= (1 << 30) || C >= (1 << 30) || T * R * C >= ((int64_t)1 << 40))
    return fail(LSPIV_ESHAPE, "bad field shape (%lld, %lld, %lld)", (long long)T, (long long)R, (long long)C);
  return LSPIV_OK;
}
}  // namespace

int lspiv_mask_dev(const float* d_fields, int64_t T, int64_t R, int64_t C, int kind, const double* params, int n_params,
                   uint8_t* d_mask, void* stream) {
  int rc = check_fields(d_fields, T, R, C);
  if (rc) return rc;
  if (!d_mask || !params) return fail(LSPIV_EINVAL, "NULL argument");
  if (kind < 0 || kind > 9) return fail(LSPIV_EINVAL, "unknown mask kind %d", kind);
  if (n_params != kMaskParams[kind]) return fail(LSPIV_EINVAL, "mask kind %d takes %d parameters, got %d", kind, kMaskParams[kind], n_params);
  if (kind == LSPIV_MASK_ROLLING && (params[0] < 1 || params[0] > 1e6)) return fail(LSPIV_EINVAL, "rolling window must be >= 1");
  if (kind >= LSPIV_MASK_WINDOW_NAN) {
    const double* w = params + (kind == LSPIV_MASK_WINDOW_NAN ? 1 : 2);
    for (int i = 0; i < 4; ++i)
      if (!(std::fabs(w[i]) <= 1024)) return fail(LSPIV_EINVAL, "window stride %g out of range", w[i]);
  }
  DeviceCtx* c;
  rc = get_ctx(&c);
  if (rc) return rc;
  hipError_t e = lspiv::launch_mask(d_fields, T, (int)R, (int)C, kind, params, d_mask, stream ? (hipStream_t)stream : c->stream);
  if (e != hipSuccess) return fail(LSPIV_EHIP, "kernel launch failed: %s", hipGetErrorString(e));
  return LSPIV_OK;
}

int lspiv_mask(const float* fields, int64_t T, int64_t R, int64_t C, int kind, const double* params, int n_params,
               uint8_t* mask) {
  std::lock_guard<std::mutex> host_lock(locks_here().host);
  int rc = check_fields(fields, T, R, C);
  if (rc) return rc;
  if (!mask) return fail(LSPIV_EINVAL, "NULL argument");
  DeviceCtx* c;
  rc = get_ctx(&c);
  if (rc) return rc;
  const size_t fb = (size_t)4 * T * R * C * sizeof(float), mb = (size_t)(mask_is_2d(kind) ? 1 : T) * R * C;
  rc = ensure(&c->d_frames, &c->frames_cap, fb);
  if (rc) return rc;
  rc = ensure(&c->d_planes, &c->planes_cap, mb);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(c->d_frames, fields, fb, hipMemcpyHostToDevice, c->stream));
  rc = lspiv_mask_dev((const float*)c->d_frames, T, R, C, kind, params, n_params, (uint8_t*)c->d_planes, c->stream);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(mask, c->d_planes, mb, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return LSPIV_OK;
}

int lspiv_mask_apply_dev(float* d_fields, int64_t T, int64_t R, int64_t C, const uint8_t* d_mask, int mask_has_time,
                         void* stream) {
  int rc = check_fields(d_fields, T, R, C);
  if (rc) return rc;
  if (!d_mask) return fail(LSPIV_EINVAL, "NULL argument");
  DeviceCtx* c;
  rc = get_ctx(&c);
  if (rc) return rc;
  hipError_t e = lspiv::launch_mask_apply(d_fields, T, R * C, d_mask, mask_has_time != 0, stream ? (hipStream_t)stream : c->stream);
  if (e != hipSuccess) return fail(LSPIV_EHIP, "kernel launch failed: %s", hipGetErrorString(e));
  return LSPIV_OK;
}

int lspiv_mask_apply(float* fields, int64_t T, int64_t R, int64_t C, const uint8_t* mask, int mask_has_time) {
  std::lock_guard<std::mutex> host_lock(locks_here().host);
  int rc = check_fields(fields, T, R, C);
  if (rc) return rc;
  if (!mask) return fail(LSPIV_EINVAL, "NULL argument");
  DeviceCtx* c;
  rc = get_ctx(&c);
  if (rc) return rc;
  const size_t fb = (size_t)4 * T * R * C * sizeof(float), mb = (size_t)(mask_has_time ? T : 1) * R * C;
  rc = ensure(&c->d_frames, &c->frames_cap, fb);
  if (rc) return rc;
  rc = ensure(&c->d_planes, &c->planes_cap, mb);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(c->d_frames, fields, fb, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(c->d_planes, mask, mb, hipMemcpyHostToDevice, c->stream));
  rc = lspiv_mask_apply_dev((float*)c->d_frames, T, R, C, (const uint8_t*)c->d_planes, mask_has_time, c->stream);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(fields, c->d_frames, fb, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return LSPIV_OK;
}

int lspiv_time_mean_dev(const float* d_fields, int64_t T, int64_t R, int64_t C, float* d_out, void* stream) {
  int rc = check_fields(d_fields, T, R, C);
  if (rc) return rc;
  if (!d_out) return fail(LSPIV_EINVAL, "NULL argument");
  DeviceCtx* c;
  rc = get_ctx(&c);
  if (rc) return rc;
  hipError_t e = lspiv::launch_time_mean(d_fields, T, R * C, d_out, stream ? (hipStream_t)stream : c->stream);
  if (e != hipSuccess) return fail(LSPIV_EHIP, "kernel launch failed: %s", hipGetErrorString(e));
  return LSPIV_OK;
}

int lspiv_time_mean(const float* fields, int64_t T, int64_t R, int64_t C, float* out) {
  std::lock_guard<std::mutex> host_lock(locks_here().host);
  int rc = check_fields(fields, T, R, C);
  if (rc) return rc;
  if (!out) return fail(LSPIV_EINVAL, "NULL argument");
  DeviceCtx* c;
  rc = get_ctx(&c);
  if (rc) return rc;
  const size_t fb = (size_t)4 * T * R * C * sizeof(float), ob = (size_t)4 * R * C * sizeof(float);
  rc = ensure(&c->d_frames, &c->frames_cap, fb);
  if (rc) return rc;
  rc = ensure(&c->d_planes, &c->planes_cap, ob);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(c->d_frames, fields, fb, hipMemcpyHostToDevice, c->stream));
  rc = lspiv_time_mean_dev((const float*)c->d_frames, T, R, C, c->d_planes, c->stream);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(out, c->d_planes, ob, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return LSPIV_OK;
}

int lspiv_window_replace_dev(float* d_fields, int64_t T, int64_t R, int64_t C, int x_min, int x_max, int y_min,
                             int y_max, int iter, void* stream) {
  int rc = check_fields(d_fields, T, R, C);
  if (rc) return rc;
  if (iter < 0 || std::abs(x_min) > 1024 || std::abs(x_max) > 1024 || std::abs(y_min) > 1024 || std::abs(y_max) > 1024)
    return fail(LSPIV_EINVAL, "bad window / iteration count");
  DeviceCtx* c;
  rc = get_ctx(&c);
  if (rc) return rc;
  hipStream_t s = stream ? (hipStream_t)stream : c->stream;
  const size_t fb = (size_t)4 * T * R * C * sizeof(float);
  float* d_tmp = nullptr;
  HIP_TRY(hipMalloc((void**)&d_tmp, fb));
  hipError_t e = hipSuccess;
  float *src = d_fields, *dst = d_tmp;
  for (int i = 0; i < iter && e == hipSuccess; ++i) {
    e = lspiv::launch_window_replace(src, 4 * T, (int)R, (int)C, x_min, x_max, y_min, y_max, dst, s);
    std::swap(src, dst);
  }
  if (e == hipSuccess && src != d_fields) e = hipMemcpyAsync(d_fields, src, fb, hipMemcpyDeviceToDevice, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);  // the temporary is freed below
  hipFree(d_tmp);
  if (e != hipSuccess) return fail(LSPIV_EHIP, "window_replace failed: %s", hipGetErrorString(e));
  return LSPIV_OK;
}

int lspiv_window_replace(float* fields, int64_t T, int64_t R, int64_t C, int x_min, int x_max, int y_min, int y_max,
                         int iter) {
  std::lock_guard<std::mutex> host_lock(locks_here().host);
  int rc = check_fields(fields, T, R, C);
  if (rc) return rc;
  DeviceCtx* c;
  rc = get_ctx(&c);
  if (rc) return rc;
  const size_t fb = (size_t)4 * T * R * C * sizeof(float);
  rc = ensure(&c->d_frames, &c->frames_cap, fb);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(c->d_frames, fields, fb, hipMemcpyHostToDevice, c->stream));
  rc = lspiv_window_replace_dev((float*)c->d_frames, T, R, C, x_min, x_max, y_min, y_max, iter, c->stream);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(fields, c->d_frames, fb, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return LSPIV_OK;
}

int lspiv_scale_velocity_dev(float* d_fields, int64_t T, int64_t n_vec, double res_x, double res_y, const double* dt,
                             void* stream) {
  if (!d_fields || !dt) return fail(LSPIV_EINVAL, "NULL argument");
  if (T < 1 || n_vec < 1) return fail(LSPIV_ESHAPE, "bad shape");
  DeviceCtx* c;
  int rc = get_ctx(&c);
  if (rc) return rc;
  hipStream_t s = stream ? (hipStream_t)stream : c->stream;
  double* d_dt = nullptr;
  HIP_TRY(hipMalloc((void**)&d_dt, (size_t)T * sizeof(double)));
  hipError_t e = hipMemcpyAsync(d_dt, dt, (size_t)T * sizeof(double), hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = lspiv::launch_scale_velocity(d_fields, T, n_vec, (float)res_x, (float)res_y, d_dt, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);  // dt is a host array and the temporary is freed below
  hipFree(d_dt);
  if (e != hipSuccess) return fail(LSPIV_EHIP, "scale_velocity failed: %s", hipGetErrorString(e));
  return LSPIV_OK;
}

// ---- element-wise filters (N2) ---------------------------------------------------------------------
int lspiv_time_diff_dev(const void* d_frames, int dtype, int64_t T, int64_t H, int64_t W, float thres, int use_abs,
                        float* d_out, void* stream) {
  if (!d_frames || !d_out) return fail(LSPIV_EINVAL, "NULL argument");
  if (dtype < 0 || dtype > 2) return fail(LSPIV_EINVAL, "dtype %d not in {0:u8, 1:f32, 2:f64}", dtype);
  if (T < 2 || H <= 0 || W <= 0) return fail(LSPIV_ESHAPE, "need >= 2 frames of positive size");
  DeviceCtx* c;
  int rc = get_ctx(&c);
  if (rc) return rc;
  hipError_t e = lspiv::launch_time_diff(d_frames, dtype, H * W, T, thres, use_abs, d_out, stream ? (hipStream_t)stream : c->stream);
  if (e != hipSuccess) return fail(LSPIV_EHIP, "kernel launch failed: %s", hipGetErrorString(e));
  return LSPIV_OK;
}

int lspiv_time_diff(const void* frames, int dtype, int64_t T, int64_t H, int64_t W, float thres, int use_abs, float* out) {
  std::lock_guard<std::mutex> host_lock(locks_here().host);
  if (!frames || !out) return fail(LSPIV_EINVAL, "NULL argument");
  if (dtype < 0 || dtype > 2) return fail(LSPIV_EINVAL, "dtype %d not in {0:u8, 1:f32, 2:f64}", dtype);
  if (T < 2 || H <= 0 || W <= 0) return fail(LSPIV_ESHAPE, "need >= 2 frames of positive size");
  DeviceCtx* c;
  int rc = get_ctx(&c);
  if (rc) return rc;
  const size_t ib = (size_t)T * H * W * elem_size(dtype), ob = (size_t)(T - 1) * H * W * sizeof(float);
  rc = ensure(&c->d_frames, &c->frames_cap, ib);
  if (rc) return rc;
  rc = ensure(&c->d_planes, &c->planes_cap, ob);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(c->d_frames, frames, ib, hipMemcpyHostToDevice, c->stream));
  rc = lspiv_time_diff_dev(c->d_frames, dtype, T, H, W, thres, use_abs, c->d_planes, c->stream);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(out, c->d_planes, ob, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return LSPIV_OK;
}

int lspiv_time_range_dev(const void* d_frames, int dtype, int64_t T, int64_t H, int64_t W, void* d_out, void* stream) {
  if (!d_frames || !d_out) return fail(LSPIV_EINVAL, "NULL argument");
  if (dtype < 0 || dtype > 2) return fail(LSPIV_EINVAL, "dtype %d not in {0:u8, 1:f32, 2:f64}", dtype);
  if (T < 1 || H <= 0 || W <= 0) return fail(LSPIV_ESHAPE, "need >= 1 frame of positive size");
  DeviceCtx* c;
  int rc = get_ctx(&c);
  if (rc) return rc;
  hipError_t e = lspiv::launch_time_range(d_frames, dtype, H * W, T, d_out, stream ? (hipStream_t)stream : c->stream);
  if (e != hipSuccess) return fail(LSPIV_EHIP, "kernel launch failed: %s", hipGetErrorString(e));
  return LSPIV_OK;
}

int lspiv_time_range(const void* frames, int dtype, int64_t T, int64_t H, int64_t W, void* out) {
  std::lock_guard<std::mutex> host_lock(locks_here().host);
  if (!frames || !out) return fail(LSPIV_EINVAL, "NULL argument");
  if (dtype < 0 || dtype > 2) return fail(LSPIV_EINVAL, "dtype %d not in {0:u8, 1:f32, 2:f64}", dtype);
  if (T < 1 || H <= 0 || W <= 0) return fail(LSPIV_ESHAPE, "need >= 1 frame of positive size");
  DeviceCtx* c;
  int rc = get_ctx(&c);
  if (rc) return rc;
  const size_t ib = (size_t)T * H * W * elem_size(dtype), ob = (size_t)H * W * elem_size(dtype);
  rc = ensure(&c->d_frames, &c->frames_cap, ib);
  if (rc) return rc;
  rc = ensure(&c->d_planes, &c->planes_cap, ob);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(c->d_frames, frames, ib, hipMemcpyHostToDevice, c->stream));
  rc = lspiv_time_range_dev(c->d_frames, dtype, T, H, W, c->d_planes, c->stream);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(out, c->d_planes, ob, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return LSPIV_OK;
}

int lspiv_minmax_dev(const float* d_frames, int64_t n, float lo, float hi, float* d_out, void* stream) {
  if (!d_frames || !d_out) return fail(LSPIV_EINVAL, "NULL argument");
  if (n < 0) return fail(LSPIV_EINVAL, "bad n");
  DeviceCtx* c;
  int rc = get_ctx(&c);
  if (rc) return rc;
  hipError_t e = lspiv::launch_minmax(d_frames, n, lo, hi, d_out, stream ? (hipStream_t)stream : c->stream);
  if (e != hipSuccess) return fail(LSPIV_EHIP, "kernel launch failed: %s", hipGetErrorString(e));
  return LSPIV_OK;
}

int lspiv_minmax(const float* frames, int64_t n, float lo, float hi, float* out) {
  std::lock_guard<std::mutex> host_lock(locks_here().host);
  if (!frames || !out) return fail(LSPIV_EINVAL, "NULL argument");
  if (n <= 0) return n == 0 ? LSPIV_OK : fail(LSPIV_EINVAL, "bad n");
  DeviceCtx* c;
  int rc = get_ctx(&c);
  if (rc) return rc;
  const size_t b = (size_t)n * sizeof(float);
  rc = ensure(&c->d_planes, &c->planes_cap, b);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(c->d_planes, frames, b, hipMemcpyHostToDevice, c->stream));
  rc = lspiv_minmax_dev(c->d_planes, n, lo, hi, c->d_planes, c->stream);  // in place
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(out, c->d_planes, b, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return LSPIV_OK;
}

// Frames.normalize's sampling interval round(T / samples), Python rounding (half to even); 0 = too few frames
static long normalize_interval(int64_t T, int samples) {
  const double ratio = (double)T / (double)samples;
  long iv = std::lround(ratio);
  if (ratio - std::floor(ratio) == 0.5) iv = ((long)std::floor(ratio) % 2 == 0) ? (long)std::floor(ratio) : (long)std::floor(ratio) + 1;
  return iv;
}

int lspiv_normalize_mean_dev(const uint8_t* d_frames, int64_t T, int64_t H, int64_t W, int samples, float* d_mean, void* stream) {
  if (!d_frames || !d_mean) return fail(LSPIV_EINVAL, "NULL argument");
  if (T < 1 || H <= 0 || W <= 0 || samples < 1 || T >= 65536) return fail(LSPIV_ESHAPE, "bad shape");
  const long iv = normalize_interval(T, samples);
  if (iv == 0) return fail(LSPIV_EINVAL, "Amount of frames is too small to provide %d samples", samples);
  DeviceCtx* c;
  int rc = get_ctx(&c);
  if (rc) return rc;
  HIP_TRY(lspiv::launch_sample_mean(d_frames, H * W, (int)T, (int)iv, d_mean, stream ? (hipStream_t)stream : c->stream));
  return LSPIV_OK;
}

int lspiv_normalize_apply_dev(const uint8_t* d_frames, int64_t T, int64_t H, int64_t W, const float* d_mean, uint8_t* d_out,
                              void* stream) {
  if (!d_frames || !d_mean || !d_out) return fail(LSPIV_EINVAL, "NULL argument");
  if (T < 1 || H <= 0 || W <= 0 || T >= 65536) return fail(LSPIV_ESHAPE, "bad shape");
  DeviceCtx* c;
  int rc = get_ctx(&c);
  if (rc) return rc;
  const size_t mm_bytes = ((size_t)2 * T * sizeof(int) + 255) & ~(size_t)255;
  rc = ensure(&c->d_scratch, &c->scratch_cap, mm_bytes + lspiv::normalize_part_bytes(H * W, (int)T));   // same-stream rule below
  if (rc) return rc;
  int* d_mm = (int*)c->d_scratch;
  float* d_part = (float*)((char*)c->d_scratch + mm_bytes);
  hipError_t e = lspiv::launch_normalize_apply(d_frames, H * W, (int)T, d_mean, d_mm, d_mm + T, d_part, d_out,
                                               stream ? (hipStream_t)stream : c->stream);
  if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? LSPIV_ENOMEM : LSPIV_EHIP, "normalize failed: %s", hipGetErrorString(e));
  return LSPIV_OK;
}

int lspiv_normalize_dev(const uint8_t* d_frames, int64_t T, int64_t H, int64_t W, int samples, uint8_t* d_out, void* stream) {
  if (!d_frames || !d_out) return fail(LSPIV_EINVAL, "NULL argument");
  if (T < 1 || H <= 0 || W <= 0 || samples < 1 || T >= 65536) return fail(LSPIV_ESHAPE, "bad shape");
  const long iv = normalize_interval(T, samples);
  if (iv == 0) return fail(LSPIV_EINVAL, "Amount of frames is too small to provide %d samples", samples);
  DeviceCtx* c;
  int rc = get_ctx(&c);
  if (rc) return rc;
  hipStream_t s = stream ? (hipStream_t)stream : c->stream;
  // temporaries (mean plane + per-frame min / max) live in the context's grow-only scratch buffer: no allocation and
  // no synchronisation per call.  (hipMallocAsync / hipFreeAsync on this stream gave run-to-run different outputs on
  // ROCm 7.2 and were dropped.)  Calls that overlap in time must therefore be issued on one stream.
  const size_t mean_bytes = ((size_t)H * W * sizeof(float) + 255) & ~(size_t)255;
  const size_t mm_bytes = ((size_t)2 * T * sizeof(int) + 255) & ~(size_t)255;
  rc = ensure(&c->d_scratch, &c->scratch_cap, mean_bytes + mm_bytes + lspiv::normalize_part_bytes(H * W, (int)T));
  if (rc) return rc;
  float* d_mean = (float*)c->d_scratch;
  int* d_mm = (int*)((char*)c->d_scratch + mean_bytes);
  float* d_part = (float*)((char*)c->d_scratch + mean_bytes + mm_bytes);
  hipError_t e = lspiv::launch_normalize(d_frames, H * W, (int)T, (int)iv, d_mean, d_mm, d_mm + T, d_part, d_out, s);
  if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? LSPIV_ENOMEM : LSPIV_EHIP, "normalize failed: %s", hipGetErrorString(e));
  return LSPIV_OK;
}

int lspiv_normalize(const uint8_t* frames, int64_t T, int64_t H, int64_t W, int samples, uint8_t* out) {
  std::lock_guard<std::mutex> host_lock(locks_here().host);
  if (!frames || !out) return fail(LSPIV_EINVAL, "NULL argument");
  if (T < 1 || H <= 0 || W <= 0) return fail(LSPIV_ESHAPE, "bad shape");
  DeviceCtx* c;
  int rc = get_ctx(&c);
  if (rc) return rc;
  const size_t b = (size_t)T * H * W;
  rc = ensure(&c->d_frames, &c->frames_cap, b);
  if (rc) return rc;
  rc = ensure(&c->d_planes, &c->planes_cap, b);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(c->d_frames, frames, b, hipMemcpyHostToDevice, c->stream));
  rc = lspiv_normalize_dev((const uint8_t*)c->d_frames, T, H, W, samples, (uint8_t*)c->d_planes, c->stream);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(out, c->d_planes, b, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return LSPIV_OK;
}

int lspiv_reduce_rolling_dev(const uint8_t* d_frames, int64_t T, int64_t H, int64_t W, int samples, uint8_t* d_out, void* stream) {
  if (!d_frames || !d_out) return fail(LSPIV_EINVAL, "NULL argument");
  if (T < 1 || H <= 0 || W <= 0 || samples < 1 || T >= (1 << 24)) return fail(LSPIV_ESHAPE, "bad shape");
  if (T < samples) return fail(LSPIV_EINVAL, "Amount of frames is smaller than requested rolling of %d samples", samples);
  DeviceCtx* c;
  int rc = get_ctx(&c);
  if (rc) return rc;
  hipStream_t s = stream ? (hipStream_t)stream : c->stream;
  rc = ensure(&c->d_scratch, &c->scratch_cap, lspiv::reduce_rolling_scratch_bytes(H * W, (int)T));
  if (rc) return rc;
  hipError_t e = lspiv::launch_reduce_rolling(d_frames, H * W, (int)T, samples, (double*)c->d_scratch, d_out, s);
  if (e != hipSuccess) return fail(LSPIV_EHIP, "reduce_rolling failed: %s", hipGetErrorString(e));
  return LSPIV_OK;
}

int lspiv_reduce_rolling(const uint8_t* frames, int64_t T, int64_t H, int64_t W, int samples, uint8_t* out) {
  std::lock_guard<std::mutex> host_lock(locks_here().host);
  if (!frames || !out) return fail(LSPIV_EINVAL, "NULL argument");
  if (T < 1 || H <= 0 || W <= 0) return fail(LSPIV_ESHAPE, "bad shape");
  DeviceCtx* c;
  int rc = get_ctx(&c);
  if (rc) return rc;
  const size_t b = (size_t)T * H * W;
  rc = ensure(&c->d_frames, &c->frames_cap, b);
  if (rc) return rc;
  rc = ensure(&c->d_planes, &c->planes_cap, b);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(c->d_frames, frames, b, hipMemcpyHostToDevice, c->stream));
  rc = lspiv_reduce_rolling_dev((const uint8_t*)c->d_frames, T, H, W, samples, (uint8_t*)c->d_planes, c->stream);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(out, c->d_planes, b, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return LSPIV_OK;
}

static int blur_common_dev(const void* d_frames, int dtype, int64_t T, int64_t H, int64_t W, int k1, int k2, float* d_out,
                           void* stream, float lo = -INFINITY, float hi = INFINITY) {
  if (!d_frames || !d_out) return fail(LSPIV_EINVAL, "NULL argument");
  if (dtype < 0 || dtype > 2) return fail(LSPIV_EINVAL, "dtype %d not in {0:u8, 1:f32, 2:f64}", dtype);
  if (T < 1 || H <= 0 || W <= 0 || T > 65535 || H >= (1 << 30) || W >= (1 << 30)) return fail(LSPIV_ESHAPE, "bad shape");
  for (int k : {k1, k2})
    if (k != 0 && (k < 1 || k > 31 || k % 2 == 0)) return fail(LSPIV_EINVAL, "kernel size %d must be odd, 1..31", k);
  if (k1 == 0) return fail(LSPIV_EINVAL, "kernel size missing");
  DeviceCtx* c;
  int rc = get_ctx(&c);
  if (rc) return rc;
  hipError_t e = lspiv::launch_blur_clip(d_frames, dtype, (int)T, (int)H, (int)W, k1, k2, lo, hi, d_out, stream ? (hipStream_t)stream : c->stream);
  if (e != hipSuccess) return fail(LSPIV_EHIP, "kernel launch failed: %s", hipGetErrorString(e));
  return LSPIV_OK;
}

static int blur_common_host(const void* frames, int dtype, int64_t T, int64_t H, int64_t W, int k1, int k2, float* out) {
  if (!frames || !out) return fail(LSPIV_EINVAL, "NULL argument");
  if (dtype < 0 || dtype > 2 || T < 1 || H <= 0 || W <= 0) return fail(LSPIV_EINVAL, "bad argument");
  std::lock_guard<std::mutex> host_lock(locks_here().host);
  DeviceCtx* c;
  int rc = get_ctx(&c);
  if (rc) return rc;
  const size_t ib = (size_t)T * H * W * elem_size(dtype), ob = (size_t)T * H * W * sizeof(float);
  rc = ensure(&c->d_frames, &c->frames_cap, ib);
  if (rc) return rc;
  rc = ensure(&c->d_planes, &c->planes_cap, ob);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(c->d_frames, frames, ib, hipMemcpyHostToDevice, c->stream));
  rc = blur_common_dev(c->d_frames, dtype, T, H, W, k1, k2, c->d_planes, c->stream);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(out, c->d_planes, ob, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return LSPIV_OK;
}

int lspiv_gaussian_blur(const void* frames, int dtype, int64_t T, int64_t H, int64_t W, int ksize, float* out) {
  return blur_common_host(frames, dtype, T, H, W, ksize, 0, out);
}
int lspiv_gaussian_blur_dev(const void* d_frames, int dtype, int64_t T, int64_t H, int64_t W, int ksize, float* d_out,
                            void* stream) {
  return blur_common_dev(d_frames, dtype, T, H, W, ksize, 0, d_out, stream);
}
int lspiv_edge_detect(const void* frames, int dtype, int64_t T, int64_t H, int64_t W, int ksize_1, int ksize_2, float* out) {
  if (ksize_2 < ksize_1) return fail(LSPIV_EINVAL, "edge_detect expects ksize_2 >= ksize_1");
  return blur_common_host(frames, dtype, T, H, W, ksize_1, ksize_2, out);
}
int lspiv_edge_detect_dev(const void* d_frames, int dtype, int64_t T, int64_t H, int64_t W, int ksize_1, int ksize_2,
                          float* d_out, void* stream) {
  if (ksize_2 < ksize_1) return fail(LSPIV_EINVAL, "edge_detect expects ksize_2 >= ksize_1");
  return blur_common_dev(d_frames, dtype, T, H, W, ksize_1, ksize_2, d_out, stream);
}

int lspiv_edge_detect_clip_dev(const void* d_frames, int dtype, int64_t T, int64_t H, int64_t W, int ksize_1, int ksize_2, float lo,
                               float hi, float* d_out, void* stream) {
  if (ksize_2 < ksize_1) return fail(LSPIV_EINVAL, "edge_detect expects ksize_2 >= ksize_1");
  if (lo != lo || hi != hi) return fail(LSPIV_EINVAL, "NaN limit");
  return blur_common_dev(d_frames, dtype, T, H, W, ksize_1, ksize_2, d_out, stream, lo, hi);
}

// ---- device-resident helpers ------------------------------------------------------------------
int lspiv_dev_malloc(void** d_ptr, size_t bytes) {
  if (!d_ptr) return fail(LSPIV_EINVAL, "d_ptr is NULL");
  DeviceCtx* c;
  int rc = get_ctx(&c);
  if (rc) return rc;
  HIP_TRY(hipMalloc(d_ptr, bytes));
  return LSPIV_OK;
}
int lspiv_dev_free(void* d_ptr) { if (d_ptr) HIP_TRY(hipFree(d_ptr)); return LSPIV_OK; }
// The three helpers below run on the library's stream and wait for it: the kernels are launched on a NON-BLOCKING
// stream, which does not synchronise with the null stream, and a null-stream hipMemcpy from pageable memory / hipMemset
// may return before the data has landed (seen as a flaky first launch reading a half-written stack).
int lspiv_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes) {
  DeviceCtx* c;
  int rc = get_ctx(&c);
  if (rc) return rc;
  if (bytes >= ((size_t)16 << 20) && !is_pinned(h_src)) {
    // a large pageable source (a frame stack, or a time chunk of one): through the two-slot pinned ring, staging threads
    // overlapped with the DMA of the previous slice, like the host entry points -- ~2x the rate of a plain pageable
    // hipMemcpy.  At least four slices per call, so that the un-overlapped first staging step stays a small part of it.
    std::lock_guard<std::mutex> host_lock(locks_here().host);
    rc = stage_ring(c, 1);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));   // the DMA runs on the copy stream: earlier kernels may still use d_dst
    const size_t slice = std::min(c->pinned_cap, std::max((size_t)4 << 20, ((bytes / 4) + 4095) & ~(size_t)4095));
    int batch = 0;
    for (size_t off = 0; off < bytes; off += slice, ++batch) {
      const int slot = batch & 1;
      const size_t nb = std::min(slice, bytes - off);
      if (batch >= 2) HIP_TRY(hipEventSynchronize(c->staged[slot]));
      staged_copy(c->pinned[slot], (const char*)h_src + off, nb);
      HIP_TRY(hipMemcpyAsync((char*)d_dst + off, c->pinned[slot], nb, hipMemcpyHostToDevice, c->copy_stream));
      HIP_TRY(hipEventRecord(c->staged[slot], c->copy_stream));
    }
    HIP_TRY(hipStreamSynchronize(c->copy_stream));
    return LSPIV_OK;
  }
  HIP_TRY(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return LSPIV_OK;
}
int lspiv_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes) {
  DeviceCtx* c;
  int rc = get_ctx(&c);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return LSPIV_OK;
}
// Host frames into a slice of an HBM-resident stack, the way the PIV host entry points bring them in (the same staging threads, the
// same float64 -> float32 conversion and DC-offset guard: a stack filled by this call holds the bytes lspiv_piv_pairs would have
// computed on), without launching anything.  Blocking.
int lspiv_upload_frames(void* d_dst, const void* frames, int dtype, int64_t n_frames, int64_t H, int64_t W, float signal_threshold) {
  if (!d_dst || !frames) return fail(LSPIV_EINVAL, "NULL argument");
  if (dtype < 0 || dtype > 2) return fail(LSPIV_EINVAL, "dtype %d not in {0:u8, 1:f32, 2:f64}", dtype);
  if (n_frames < 0 || H <= 0 || W <= 0) return fail(LSPIV_ESHAPE, "bad shape (%lld, %lld, %lld)", (long long)n_frames, (long long)H, (long long)W);
  if (n_frames == 0) return LSPIV_OK;
  std::lock_guard<std::mutex> host_lock(locks_here().host);
  DeviceCtx* c;
  int rc = get_ctx(&c);
  if (rc) return rc;
  const int dev_dtype = dtype == LSPIV_F64 ? LSPIV_F32 : dtype;
  const size_t frame_elems = (size_t)H * W, frame_bytes = frame_elems * elem_size(dev_dtype), src_frame_bytes = frame_elems * elem_size(dtype);
  rc = stage_ring(c, frame_bytes);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(c->stream));   // the DMA runs on the copy stream: earlier kernels may still use d_dst
  const bool src_pinned = dtype != LSPIV_F64 && is_pinned(frames);
  // at least four slices per call, so that the un-overlapped first staging step stays a small part of it
  const int64_t fpb = std::max<int64_t>(1, std::min<int64_t>((int64_t)(c->pinned_cap / frame_bytes), (n_frames + 3) / 4));
  int batch = 0;
  for (int64_t f0 = 0; f0 < n_frames; ++batch) {
    const int64_t f1 = std::min<int64_t>(n_frames, f0 + fpb);
    const int slot = batch & 1;
    if (batch >= 2) HIP_TRY(hipEventSynchronize(c->staged[slot]));
    const size_t nb = (size_t)(f1 - f0) * frame_bytes;
    const void* dma_src = c->pinned[slot];
    if (dtype == LSPIV_F64) {
      const double* src64 = (const double*)((const char*)frames + (size_t)f0 * src_frame_bytes);
      const std::vector<double> off = narrow_offsets(src64, frame_elems, f1 - f0, signal_threshold);
      lspiv_host::staged_narrow((float*)c->pinned[slot], src64, frame_elems, (size_t)(f1 - f0), off.data());
    } else if (src_pinned)
      dma_src = (const char*)frames + (size_t)f0 * src_frame_bytes;
    else
      staged_copy(c->pinned[slot], (const char*)frames + (size_t)f0 * src_frame_bytes, nb);
    HIP_TRY(hipMemcpyAsync((char*)d_dst + (size_t)f0 * frame_bytes, dma_src, nb, hipMemcpyHostToDevice, c->copy_stream));
    HIP_TRY(hipEventRecord(c->staged[slot], c->copy_stream));
    f0 = f1;
  }
  HIP_TRY(hipStreamSynchronize(c->copy_stream));
  return LSPIV_OK;
}

int lspiv_trace(int enable) {
  DeviceCtx* c;
  int rc = get_ctx(&c);
  if (rc) return rc;
  Trace& t = g_trace[current_device_slot()];
  HIP_TRY(hipDeviceSynchronize());
  std::lock_guard<std::mutex> lk(t.mu);
  for (TraceRec& r : t.recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  t.recs.clear();
  if (enable) {
    if (!t.base) HIP_TRY(hipEventCreate(&t.base));
    HIP_TRY(hipEventRecord(t.base, c->stream));
    HIP_TRY(hipEventSynchronize(t.base));
  }
  t.on.store(enable != 0);
  return LSPIV_OK;
}
int lspiv_trace_read(int64_t cap, int32_t* kind, double* start_ms, double* end_ms, int64_t* n) {
  if (!n || cap < 0 || (cap > 0 && (!kind || !start_ms || !end_ms))) return fail(LSPIV_EINVAL, "bad argument");
  DeviceCtx* c;
  int rc = get_ctx(&c);
  if (rc) return rc;
  Trace& t = g_trace[current_device_slot()];
  HIP_TRY(hipDeviceSynchronize());
  std::lock_guard<std::mutex> lk(t.mu);
  *n = (int64_t)t.recs.size();
  if (!t.base) return LSPIV_OK;
  for (int64_t i = 0; i < std::min<int64_t>(cap, *n); ++i) {
    float a = 0.0f, b = 0.0f;
    HIP_TRY(hipEventElapsedTime(&a, t.base, t.recs[(size_t)i].e0));
    HIP_TRY(hipEventElapsedTime(&b, t.base, t.recs[(size_t)i].e1));
    kind[i] = t.recs[(size_t)i].kind; start_ms[i] = a; end_ms[i] = b;
  }
  return LSPIV_OK;
}

int lspiv_host_alloc(void** h_ptr, size_t bytes) {
  if (!h_ptr) return fail(LSPIV_EINVAL, "h_ptr is NULL");
  DeviceCtx* c;
  int rc = get_ctx(&c);
  if (rc) return rc;
  HIP_TRY(hipHostMalloc(h_ptr, bytes ? bytes : 1, hipHostMallocDefault));
  return LSPIV_OK;
}
int lspiv_host_free(void* h_ptr) {
  if (h_ptr) HIP_TRY(hipHostFree(h_ptr));
  return LSPIV_OK;
}
int lspiv_memset_dev(void* d_ptr, int value, size_t bytes) {
  DeviceCtx* c;
  int rc = get_ctx(&c);
  if (rc) return rc;
  HIP_TRY(hipMemsetAsync(d_ptr, value, bytes, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return LSPIV_OK;
}

int lspiv_event_create(void** ev) {
  if (!ev) return fail(LSPIV_EINVAL, "ev is NULL");
  hipEvent_t e;
  HIP_TRY(hipEventCreate(&e));
  *ev = (void*)e;
  return LSPIV_OK;
}
int lspiv_event_record(void* ev) {
  DeviceCtx* c;
  int rc = get_ctx(&c);
  if (rc) return rc;
  HIP_TRY(hipEventRecord((hipEvent_t)ev, c->stream));
  return LSPIV_OK;
}
int lspiv_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms) {
  if (!ms) return fail(LSPIV_EINVAL, "ms is NULL");
  HIP_TRY(hipEventSynchronize((hipEvent_t)ev_stop));
  HIP_TRY(hipEventElapsedTime(ms, (hipEvent_t)ev_start, (hipEvent_t)ev_stop));
  return LSPIV_OK;
}
int lspiv_event_destroy(void* ev) { if (ev) HIP_TRY(hipEventDestroy((hipEvent_t)ev)); return LSPIV_OK; }

int lspiv_stream_create(void** stream) {
  if (!stream) return fail(LSPIV_EINVAL, "stream is NULL");
  DeviceCtx* c;
  int rc = get_ctx(&c);
  if (rc) return rc;
  hipStream_t s;
  HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  *stream = s;
  return LSPIV_OK;
}
int lspiv_stream_create_priority(void** stream, int priority) {
  if (!stream) return fail(LSPIV_EINVAL, "stream is NULL");
  if (priority == 0) return lspiv_stream_create(stream);
  DeviceCtx* c;
  int rc = get_ctx(&c);
  if (rc) return rc;
  int least = 0, greatest = 0;   // HIP: numerically LOWER = higher priority
  HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
  hipStream_t s;
  HIP_TRY(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, priority > 0 ? greatest : least));
  *stream = s;
  return LSPIV_OK;
}
int lspiv_stream_destroy(void* stream) {
  if (!stream) return LSPIV_OK;
  // the rescue lists of that stream go with it (a later stream may get the same handle value)
  std::vector<DeviceCtx*> ctxs;
  { std::lock_guard<std::mutex> lk(g_mu); ctxs = g_ctx; }
  for (size_t d = 0; d < ctxs.size() && d < (size_t)kMaxDevices; ++d) {
    DeviceCtx* c = ctxs[d];
    if (!c) continue;
    std::lock_guard<std::mutex> lk(g_locks[d].lists);
    for (size_t k = 0; k < c->rescue.size(); ++k) {
      if (c->rescue[k].stream != (hipStream_t)stream) continue;
      (void)hipStreamSynchronize((hipStream_t)stream);
      if (c->rescue[k].base) (void)hipFree(c->rescue[k].base);
      c->rescue.erase(c->rescue.begin() + (long)k);
      break;
    }
  }
  HIP_TRY(hipStreamDestroy((hipStream_t)stream));
  return LSPIV_OK;
}
int lspiv_stream_release(void* stream) {
  // a stream the caller created itself (hipStreamCreate) and handed to "_dev" entry points: drop what the library keeps for it
  // (the rescue lists) -- lspiv_stream_destroy does the same for streams of lspiv_stream_create
  std::vector<DeviceCtx*> ctxs;
  { std::lock_guard<std::mutex> lk(g_mu); ctxs = g_ctx; }
  for (size_t d = 0; d < ctxs.size() && d < (size_t)kMaxDevices; ++d) {
    DeviceCtx* c = ctxs[d];
    if (!c) continue;
    std::lock_guard<std::mutex> launch_lock(g_locks[d].dispatch);
    std::lock_guard<std::mutex> lk(g_locks[d].lists);
    const hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    for (size_t k = 0; k < c->rescue.size(); ++k) {
      if (c->rescue[k].stream != s) continue;
      (void)hipStreamSynchronize(s);
      if (c->rescue[k].base) (void)hipFree(c->rescue[k].base);
      c->rescue.erase(c->rescue.begin() + (long)k);
      break;
    }
  }
  return LSPIV_OK;
}
int lspiv_stream_synchronize(void* stream) {
  DeviceCtx* c;
  int rc = get_ctx(&c);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(stream ? (hipStream_t)stream : c->stream));
  return LSPIV_OK;
}
int lspiv_event_record_on(void* ev, void* stream) {
  if (!ev) return fail(LSPIV_EINVAL, "event is NULL");
  DeviceCtx* c;
  int rc = get_ctx(&c);
  if (rc) return rc;
  HIP_TRY(hipEventRecord((hipEvent_t)ev, stream ? (hipStream_t)stream : c->stream));
  return LSPIV_OK;
}
int lspiv_stream_wait_event(void* stream, void* ev) {
  if (!ev) return fail(LSPIV_EINVAL, "event is NULL");
  DeviceCtx* c;
  int rc = get_ctx(&c);
  if (rc) return rc;
  HIP_TRY(hipStreamWaitEvent(stream ? (hipStream_t)stream : c->stream, (hipEvent_t)ev, 0));
  return LSPIV_OK;
}

int lspiv_synth_particles_dev(void* d_frames, int64_t T, int64_t H, int64_t W, uint64_t seed, float density) {
  if (!d_frames || T < 1 || H < 8 || W < 8 || !(density > 0.0f)) return fail(LSPIV_EINVAL, "bad argument");
  DeviceCtx* c;
  int rc = get_ctx(&c);
  if (rc) return rc;
  hipError_t e = lspiv::launch_synth_particles((uint8_t*)d_frames, T, (int)H, (int)W, seed, density, c->stream);
  if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? LSPIV_ENOMEM : LSPIV_EHIP, "synth failed: %s", hipGetErrorString(e));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return LSPIV_OK;
}

int lspiv_debug_narrow(const double* frames, int64_t frame_elems, int64_t n_frames, int min_abs, float* out, double* offsets) {
  // test hook, host only (no device needed): the float64 -> float32 staging conversion of the host entry points
  if (!frames || !out || frame_elems < 0 || n_frames < 0) return fail(LSPIV_EINVAL, "bad argument");
  std::vector<double> off((size_t)n_frames, 0.0);
  if (min_abs >= 0)
    for (int64_t f = 0; f < n_frames; ++f) off[(size_t)f] = lspiv_host::frame_offset(frames + (size_t)f * frame_elems, (size_t)frame_elems, (double)min_abs);
  lspiv_host::staged_narrow(out, frames, (size_t)frame_elems, (size_t)n_frames, off.data());
  if (offsets) memcpy(offsets, off.data(), off.size() * sizeof(double));
  return lspiv_host::stage_threads();
}
int lspiv_debug_project_division(int* mismatches) {
  // test hook: project_mix_kernel's division-free quotient against the division, every sum 0 .. 255 c for every count c = 1 .. 255
  if (!mismatches) return fail(LSPIV_EINVAL, "NULL argument");
  DeviceCtx* c;
  int rc = get_ctx(&c);
  if (rc) return rc;
  rc = ensure(&c->d_scratch, &c->scratch_cap, sizeof(int));
  if (rc) return rc;
  HIP_TRY(hipMemsetAsync(c->d_scratch, 0, sizeof(int), c->stream));
  const hipError_t e = lspiv::launch_division_check((int*)c->d_scratch, c->stream);
  if (e != hipSuccess) return fail(LSPIV_EHIP, "kernel launch failed: %s", hipGetErrorString(e));
  HIP_TRY(hipMemcpyAsync(mismatches, c->d_scratch, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return LSPIV_OK;
}
int lspiv_debug_hold_lock(int device, int which, int milliseconds) {
  if (device < 0 || device >= kMaxDevices || which < 0 || which > 2 + DeviceCtx::kProjSlots || milliseconds < 0 || milliseconds > 10000)
    return fail(LSPIV_EINVAL, "device %d / lock %d / %d ms out of range", device, which, milliseconds);
  DeviceLocks& l = g_locks[device];
  std::lock_guard<std::mutex> lk(which == 0 ? l.host : which == 1 ? l.dispatch : which == 2 ? l.lists : l.project[which - 3]);
  std::this_thread::sleep_for(std::chrono::milliseconds(milliseconds));
  return LSPIV_OK;
}
int lspiv_debug_segments(int64_t n_pairs, int64_t pair_offset, int seg_len, int64_t* seg_first, int64_t* n_seg) {
  if (n_pairs < 1 || n_pairs > 0x7fffffff || pair_offset < 0 || seg_len < 1 || !seg_first || !n_seg) return LSPIV_EINVAL;
  const lspiv::WalkSegments w = lspiv::walk_segments((uint32_t)n_pairs, pair_offset, (uint32_t)seg_len);
  *seg_first = w.seg_first;
  *n_seg = w.n_seg;
  return LSPIV_OK;
}

int lspiv_debug_fft(int n, int inverse, const float* in, float* out, int64_t count) {
  if (!in || !out || count < 1 || count > (1 << 20)) return fail(LSPIV_EINVAL, "bad argument");
  DeviceCtx* c;
  int rc = get_ctx(&c);
  if (rc) return rc;
  const size_t bytes = (size_t)count * 2 * n * sizeof(float);
  rc = ensure(&c->d_scratch, &c->scratch_cap, 2 * bytes);
  if (rc) return rc;
  float* d_in = (float*)c->d_scratch;
  float* d_out = d_in + (size_t)count * 2 * n;
  HIP_TRY(hipMemcpyAsync(d_in, in, bytes, hipMemcpyHostToDevice, c->stream));
  hipError_t e = lspiv::launch_fft_debug(n, inverse != 0, d_in, d_out, (int)count, c->stream);
  if (e == hipErrorInvalidValue) return fail(LSPIV_EUNSUPPORTED, "no register FFT of length %d", n);
  if (e != hipSuccess) return fail(LSPIV_EHIP, "kernel launch failed: %s", hipGetErrorString(e));
  HIP_TRY(hipMemcpyAsync(out, d_out, bytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return LSPIV_OK;
}

}  // extern "C"
