// ectrans_mi.hip -- host orchestration + C-ABI (include/ectrans_mi.h) of libectrans_mi.so.
// Built with hipcc --offload-arch=gfx950 (product) or g++ -x c++ -DEMI_CPU_EMU (test emulator).
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <map>
#include <string>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "../../include/ectrans_mi.h"
#include "emi_kernels.h"
#include "emi_setup.h"
#include <dlfcn.h>

// ---- roctx ranges named after the reference's GSTATS phases (gpu/internal/tpm_stats.F90:33-55 turns GSTATS into NVTX ranges;
// labels of ectrans-benchmark.F90:1681-1697), so that a rocprofv3 --marker-trace of a Fortran or C host shows INV_TRANS /
// LTINV_CTL / FTINV_CTL ... around the kernel launches.  librocprofiler-sdk-roctx is opened at run time (no link dependency:
// absent library or EMI_ROCTX=0 = no ranges); a range covers the host-side enqueue of its phase.
struct EmiRoctx {
  int (*push)(const char *) = nullptr;
  int (*pop)() = nullptr;
  bool tried = false;
  void load() {
    tried = true;
#ifndef EMI_CPU_EMU
    const char *e = getenv("EMI_ROCTX");
    if (e && atoi(e) == 0) return;
    void *h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librocprofiler-sdk-roctx.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return;
    push = (int (*)(const char *))dlsym(h, "roctxRangePushA");
    pop = (int (*)())dlsym(h, "roctxRangePop");
    if (!push || !pop) push = nullptr, pop = nullptr;
#endif
  }
};
static EmiRoctx g_roctx;
struct EmiRange {  // RAII: one nested range
  bool on;
  explicit EmiRange(const char *label) {
    if (!g_roctx.tried) g_roctx.load();
    on = g_roctx.push != nullptr;
    if (on) g_roctx.push(label);
  }
  ~EmiRange() {
    if (on) g_roctx.pop();
  }
};
extern "C" int emi_roctx_active(void) {
  if (!g_roctx.tried) g_roctx.load();
  return g_roctx.push != nullptr;
}
static const char *const EMI_LBL_SETUP = "SETUP_TRANS    - Setup ecTrans handle", *const EMI_LBL_SULEG = "SULEG          - Comp. of Leg. poly.",
                         *const EMI_LBL_INV = "INV_TRANS      - Inverse transform", *const EMI_LBL_DIR = "DIR_TRANS      - Direct transform",
                         *const EMI_LBL_LTINV = "LTINV_CTL      - Inv. Legendre transform", *const EMI_LBL_LTDIR = "LTDIR_CTL      - Dir. Legendre transform",
                         *const EMI_LBL_FTDIR = "FTDIR_CTL      - Dir. Fourier transform", *const EMI_LBL_FTINV = "FTINV_CTL      - Inv. Fourier transform",
                         *const EMI_LBL_TRMTOL = "LTINV_CTL      - M to L transposition", *const EMI_LBL_TRLTOM = "LTDIR_CTL      - L to M transposition";

#ifdef EMI_CPU_EMU
thread_local EmuCtx *emu_ctx = nullptr;
#endif

// ------------------------------------------------------------------------------------------
// errors + runtime helpers
// ------------------------------------------------------------------------------------------
static char g_err[1024] = "";
void emi_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char *emi_last_error(void) { return g_err; }

#define EMI_FAIL(code, ...)     \
  do {                          \
    emi_set_error(__VA_ARGS__); \
    return (code);              \
  } while (0)

#ifndef EMI_CPU_EMU
static void (*g_oom_hook)() = nullptr;  // frees what the library holds only as a cache (emi_stage.h: idle staging buffers)
int emi_dev_malloc(void **p, size_t bytes) {
  if (hipMalloc(p, bytes ? bytes : 16) == hipSuccess) return 0;
  (void)hipGetLastError();
  if (g_oom_hook) g_oom_hook();
  EMI_CHECK(hipMalloc(p, bytes ? bytes : 16));
  return 0;
}
int emi_dev_free(void *p) {
  if (p) EMI_CHECK(hipFree(p));
  return 0;
}
int emi_dev_memset(void *p, int v, size_t bytes, emi_stream_t s) {
  EMI_CHECK(hipMemsetAsync(p, v, bytes, s));
  return 0;
}
int emi_h2d(void *d, const void *s, size_t b, emi_stream_t st) {
  EMI_CHECK(hipMemcpyAsync(d, s, b, hipMemcpyHostToDevice, st));
  return 0;
}
int emi_d2h(void *d, const void *s, size_t b, emi_stream_t st) {
  EMI_CHECK(hipMemcpyAsync(d, s, b, hipMemcpyDeviceToHost, st));
  return 0;
}
int emi_d2d(void *d, const void *s, size_t b, emi_stream_t st) {
  EMI_CHECK(hipMemcpyAsync(d, s, b, hipMemcpyDeviceToDevice, st));
  return 0;
}
int emi_stream_sync(emi_stream_t s) {
  EMI_CHECK(hipStreamSynchronize(s));
  return 0;
}
int emi_mem_info(size_t *f, size_t *t) {
  EMI_CHECK(hipMemGetInfo(f, t));
  return 0;
}
#else
int emi_dev_malloc(void **p, size_t bytes) {
  *p = aligned_alloc(64, ((bytes ? bytes : 16) + 63) / 64 * 64);
  return *p ? 0 : -1;
}
int emi_dev_free(void *p) {
  free(p);
  return 0;
}
int emi_dev_memset(void *p, int v, size_t bytes, emi_stream_t) {
  memset(p, v, bytes);
  return 0;
}
int emi_h2d(void *d, const void *s, size_t b, emi_stream_t) {
  memcpy(d, s, b);
  return 0;
}
int emi_d2h(void *d, const void *s, size_t b, emi_stream_t) {
  memcpy(d, s, b);
  return 0;
}
int emi_d2d(void *d, const void *s, size_t b, emi_stream_t) {
  memmove(d, s, b);
  return 0;
}
int emi_stream_sync(emi_stream_t) { return 0; }
int emi_mem_info(size_t *f, size_t *t) {
  *f = (size_t)8 << 30;
  *t = (size_t)8 << 30;
  return 0;
}
#endif

#include "emi_stage.h"
#ifndef EMI_CPU_EMU
static const bool g_oom_hook_set = (g_oom_hook = emi_stage::trim, true);
#endif

// ---- EMI_MEM_AUTO: where do the caller's arrays live?  The reference GPU back-end treats its caller's arrays as present-or-copyin
// (gpu/internal/trltog_mod.F90:501-523, trgtol_mod.F90:444-448, ltinv_mod.F90:334-338, updsp_mod.F90:96-97): arrays already on the
// device are used in place, host arrays are copied.  Device and managed allocations of any visible GPU count as device memory;
// pageable, pinned and registered host memory is staged (kernels reading rows over PCIe piece by piece would be far slower).
extern "C" int emi_ptr_space(const void *p) {
#ifdef EMI_CPU_EMU
  (void)p;
  return EMI_MEM_HOST;  // the emulator's "device" memory is host memory: staging is a memcpy
#else
  if (!p) return EMI_MEM_HOST;
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, p) != hipSuccess) {  // older runtimes: "invalid value" for memory HIP has never seen
    (void)hipGetLastError();
    return EMI_MEM_HOST;
  }
  return (at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged || at.type == hipMemoryTypeArray) ? EMI_MEM_DEVICE : EMI_MEM_HOST;
#endif
}
// mem_space of a call -> EMI_MEM_HOST or EMI_MEM_DEVICE; AUTO classifies every array that is present
static int agree_on_mixed_arrays(const char *who, int ndev, int nhost);
// `collective`: the routine is called by every task together (the transforms): with several tasks a task whose arrays are in both memories must
// not fail alone -- its peers, whose arrays are all in one place, would go on into the exchange and wait for it forever -- so the outcome of
// the classification is agreed over the host collectives (one word per task) before anybody starts.
static int resolve_space(const char *who, int mem_space, std::initializer_list<const void *> arrays, int *out, bool collective = false) {
  if (mem_space == EMI_MEM_HOST || mem_space == EMI_MEM_DEVICE) {
    *out = mem_space;
    return 0;
  }
  if (mem_space != EMI_MEM_AUTO) EMI_FAIL(EMI_ERR_ARG, "%s: mem_space = %d (EMI_MEM_HOST, EMI_MEM_DEVICE or EMI_MEM_AUTO)", who, mem_space);
  int ndev = 0, nhost = 0;
  for (const void *q : arrays)
    if (q) (emi_ptr_space(q) == EMI_MEM_DEVICE ? ndev : nhost)++;
  if (collective) {
    if (agree_on_mixed_arrays(who, ndev, nhost)) return EMI_ERR_ARG;
  } else if (ndev && nhost) {
    EMI_FAIL(EMI_ERR_ARG, "%s: %d ARRAYS OF THE CALL ARE IN DEVICE MEMORY AND %d IN HOST MEMORY (all of them must live in one place)", who, ndev, nhost);
  }
  *out = ndev ? EMI_MEM_DEVICE : EMI_MEM_HOST;
  return 0;
}

template <class T>
static int upload(const std::vector<T> &h, T **d) {
  void *p = nullptr;
  if (emi_dev_malloc(&p, h.size() * sizeof(T))) return -1;
  if (!h.empty() && emi_h2d(p, h.data(), h.size() * sizeof(T), 0)) return -1;
  if (emi_stream_sync(0)) return -1;  // the host vector may die right after this call
  *d = (T *)p;
  return 0;
}

// launch the fp64 or the fp32 instantiation of a kernel; RT names the real type inside the argument list
#define EMI_LAUNCH_P(esz, kern, grid, block, lds, st, ...)                     \
  do {                                                                         \
    if ((esz) == 8) {                                                          \
      typedef double RT;                                                       \
      EMI_LAUNCH(emi_f64::kern, grid, block, lds, st, __VA_ARGS__);            \
    } else {                                                                   \
      typedef float RT;                                                        \
      EMI_LAUNCH(emi_f32::kern, grid, block, lds, st, __VA_ARGS__);            \
    }                                                                          \
  } while (0)

// ------------------------------------------------------------------------------------------
// per-resolution plan
// ------------------------------------------------------------------------------------------
struct FftClass {  // one launch group of the FFT kernels: latitudes sharing a workgroup size, fields per workgroup and kernel
  int nthr = 0, fbk = 0;
  int hot = 0;  // > 0: specialised kernel k_fft_*_hot<hot> (EMI_HOT_PLAN_LIST)
  int r16 = 0;  // > 0: register-resident kernel k_fft_*_r16<r16> (EMI_R16_LIST)
  int split = 0;  // 2: k_fft_*_r16p<r16> (EMI_R16S_LIST): the row as two convolutions of half its half-length
  int mr = 0;   // 1: direct mixed-radix kernels k_fft_*_mr (EMI_MR_RADICES)
  int gmem = 0;  // 1: the work array does not fit the LDS; k_fft_*_gm on a global scratch buffer (elems: complex numbers per workgroup)
  long long gm_elems = 0;
  std::vector<int> lats;
  int *d_lats = nullptr;
  FftRowDev *d_rows = nullptr;  // one record per latitude of `lats`
  size_t lds = 0;
};
// Legendre tile maps for one column-tile count: block id -> (ml, row tile, column tile).
// Workgroups are dealt round-robin over the 8 XCDs (block b runs on XCD b % 8).  Each XCD gets, for
// every wavenumber, a fixed range of column tiles (and, when there are fewer column tiles than XCDs,
// a residue class of row tiles): the ~64 tiles an XCD has in flight then share a few operand column
// blocks and the Legendre-panel row blocks of one m in its 4 MiB L2, instead of streaming 26 different
// column blocks per row tile (measured: 13x the algorithmic HBM bytes).  Tiles stay m-ascending
// (longest K first) within every XCD, so all XCDs progress through m together.
struct LegMaps {
  int2 *d_inv = nullptr, *d_dir = nullptr, *d_inv_wide = nullptr, *d_dir_wide = nullptr;  // *_wide: fp32 library, the tiles of zonal wavenumber 0 (k_leg_*_wide)
  long long n_inv = 0, n_dir = 0, n_inv_wide = 0, n_dir_wide = 0;
};

struct Plan {
  bool active = false;
  int nsmax = 0, ndgl = 0, ndgnh = 0;
  bool reduced = false;
  double ra = 6371229.0;
  // ---- global geometry (identical on every task)
  std::vector<int> nloen, nmen, ndglu, procm;  // procm[m]: owning task (0-based) of wavenumber m
  std::vector<int> latlo;                      // [nproc+1] latitude band of every task
  std::vector<double> rmu, rw, racthe, cos2;
  int ngptotg = 0, nspec2g = 0;
  // ---- this task's share
  int nproc = 1, me = 0;  // W-set decomposition (NPRTRW, MYSETW - 1)
  // V-sets: the band of W-set w (latitudes latlo[w] .. latlo[w+1]) is cut into NPRTRV sub-bands of whole latitudes, the grid
  // shares of its tasks; vlat[w * nprv + v]: first latitude (0-based, global) of task (w, v), vlat[nproc * nprv] = NDGL
  int nprv = 1, mev = 0;
  std::vector<int> vlat;
  int vfirst(int w, int v) const { return nprv > 1 ? vlat[(size_t)w * nprv + v] : latlo[w]; }        // first latitude of task (w, v)
  int vlast(int w, int v) const { return nprv > 1 ? vlat[(size_t)w * nprv + v + 1] : latlo[w + 1]; }  // one past its last latitude
  long long vpoints(int w, int v) const {  // grid points of task (w, v)
    long long c = 0;
    for (int j = vfirst(w, v); j < vlast(w, v); j++) c += nloen[j];
    return c;
  }
  long long voffset(int v) const {  // first point of sub-band v inside this task's band
    long long c = 0;
    for (int j = latlo[me]; j < vfirst(me, v); j++) c += nloen[j];
    return c;
  }
  int nump = 0, nlat = 0, lat0 = 0;
  int ngptot = 0, nspec2 = 0;
  std::vector<int> mval, nasm0;            // [nump]
  std::vector<int> l_nmen, l_gpoff, l_fbase;  // [nlat]
  std::vector<int> wbase, wrows, ldp, ldk, lattile_pref, ktile_pref, lbase;
  std::vector<long long> offS, offA, offTS, offTA;
  long long frows = 0, lrows = 0, wrows_total = 0, p_elems = 0, pt_elems = 0;
  // exchange (rows of the Fourier buffers per peer); inverse sends Legendre-side -> FFT-side
  std::vector<long long> leg_rows, leg_disp, fft_rows, fft_disp;  // [nproc]
  // device
  EmiGeomDev g{};
  std::vector<void *> dev_allocs;
  int esz = 8;  // bytes per real: 8 (fp64 library, the reference's _dp build) or 4 (fp32, _sp)
  char *d_P = nullptr, *d_PT = nullptr;  // esz-sized reals
  // fft
  std::vector<FftPlanDev> fplans;
  std::vector<int> planid;  // [nlat]
  FftTabDev ftab{};
  std::vector<FftClass> fclass;
  std::map<int, LegMaps> legmaps;  // by column-tile count
  // work buffers (grown on demand): W, Legendre-side Fourier buffer, FFT-side Fourier buffer
  // (the same allocation when nproc == 1)
  char *d_W = nullptr, *d_FBL = nullptr, *d_FBF = nullptr;  // esz-sized reals; capacities in reals
  size_t cap_W = 0, cap_FBL = 0, cap_FBF = 0;
  void *d_desc = nullptr;
  size_t cap_desc = 0;
  char *d_fftscr = nullptr;  // work arrays of the rows that exceed the LDS (k_fft_*_gm)
  size_t cap_fftscr = 0;
  // Calls on one resolution share d_desc, W and the Fourier buffers, whatever stream each call names: every call
  // starts by making its stream wait for the event the previous call of this resolution recorded at its end.
#ifndef EMI_CPU_EMU
  hipEvent_t ev_done = nullptr;
#endif
  bool ev_valid = false;
};

static int roundup(int a, int b) { return (a + b - 1) / b * b; }
// EMI_TEST_PATHS (tests only; read at every SETUP_TRANS / call, so a test can flip it): bit 0 = exchange-order row tables on ONE task
// (g.fftrow / legN / legS as with several tasks), bit 1 = the 4-batch three-stream pipeline on one task, bit 2 = the scalar fields of
// DIR_TRANS through k_postpack_dir instead of k_leg_dir's epilogue.  Each is the code several tasks (or the adjoint options) run; the
// switch keeps it under the one-GPU test tier.  Results are identical with and without.
static int test_paths() {
  const char *e = getenv("EMI_TEST_PATHS");
  return e ? atoi(e) : 0;
}
// dynamic LDS of k_leg_dir: the operand stage of its two-parity tile (panel and Fourier rows of 16 latitudes: 56 KiB in fp64, 28 in fp32) and the
// staged destinations of the tile's 64 fields (1 KiB).  A constant per precision since round 5 (no row-number tables, nothing depends on
// NDGNH), so the check is a compile-time one: two workgroups per CU need it below 64 KiB each.
static_assert(LG_LDS_BYTES_DIR + 1024 + 64 <= 65536, "k_leg_dir: stage image + epilogue table above 64 KiB, two workgroups per CU no longer fit");
static size_t leg_dir_lds_bytes(const Plan &P) {
  return (size_t)LG_LDS_BYTES_DIR * P.esz / 8 + 1024 + 64;
}

// order a call on `st` behind the previous call of the same resolution (device-side wait, nothing blocks the host)
static int plan_begin(Plan &P, emi_stream_t st) {
#ifndef EMI_CPU_EMU
  if (P.ev_valid) EMI_CHECK(hipStreamWaitEvent(st, P.ev_done, 0));
#else
  (void)P;
  (void)st;
#endif
  return 0;
}
static int plan_end(Plan &P, emi_stream_t st) {
#ifndef EMI_CPU_EMU
  if (!P.ev_done) EMI_CHECK(hipEventCreateWithFlags(&P.ev_done, hipEventDisableTiming));
  EMI_CHECK(hipEventRecord(P.ev_done, st));
  P.ev_valid = true;
#else
  (void)P;
  (void)st;
#endif
  return 0;
}
// host waits until the last call of the resolution has finished (before its buffers are freed or regrown)
static const char *emi_rt_errstr(int rc) {
#ifndef EMI_CPU_EMU
  return hipGetErrorString((hipError_t)rc);
#else
  (void)rc;
  return "";
#endif
}
// Returns the status of the wait (0 = the stream reached the event): a kernel fault or a lost device surfaces here, and emi_wait -- the
// only completion point of the Fortran shim and transi for device-resident arrays -- turns it into EMI_ERR_RUNTIME.
static int plan_quiesce(Plan &P) {
#ifndef EMI_CPU_EMU
  if (P.ev_valid) return (int)hipEventSynchronize(P.ev_done);
#else
  (void)P;
#endif
  return 0;
}

static struct {
  bool init = false;
  int max_resol = 1;
  double ra = 6371229.0;
  int nproc = 1, myproc = 1;  // the W-set decomposition the plans are built on: NPRTRW tasks, this one is MYSETW
  // NPRTRV > 1 (sump_trans0_mod.F90:49, pe2set_mod.F90:111-112): nproc_all = NPRTRW x NPRTRV tasks; task myproc_all is
  // (MYSETW, MYSETV) = ((myproc_all - 1) / NPRTRV + 1, mod(myproc_all - 1, NPRTRV) + 1); the tasks of a W-set are neighbours
  int nproc_all = 1, myproc_all = 1, nprtrv = 1, mysetv = 1, nprtrv_preset = 0;
  std::vector<Plan *> plans;
  int max_batch = 0;
  int profile = 0;  // 1: phase timers per call; 2: accumulated over the calls since emi_set_profile(2)
  emi_alltoallv_fn a2a = nullptr;
  void *a2a_user = nullptr;
  emi_bcast_fn hc_bcast = nullptr;  // host collectives for DIST_x / GATH_x with several tasks
  emi_allgatherv_fn hc_gather = nullptr;
  void *hc_user = nullptr;
} G;

static Plan *get_plan(int kresol) {
  if (!G.init || kresol < 1 || kresol > (int)G.plans.size() || !G.plans[kresol - 1] || !G.plans[kresol - 1]->active) return nullptr;
  return G.plans[kresol - 1];
}

extern "C" int emi_set_nprtrv(int nprtrv) {
  if (G.init) EMI_FAIL(EMI_ERR_STATE, "emi_set_nprtrv: after SETUP_TRANS0");
  if (nprtrv < 1) EMI_FAIL(EMI_ERR_ARG, "emi_set_nprtrv: NPRTRV = %d", nprtrv);
  G.nprtrv_preset = nprtrv;
  return EMI_SUCCESS;
}
extern "C" int emi_init(const emi_init_t *cfg) {
  // SETUP_TRANS0 is idempotent (setup_trans0.F90:108-111)
  if (G.init) return EMI_SUCCESS;
  emi_init_t c{};
  if (cfg) c = *cfg;
  G.max_resol = c.kmax_resol > 0 ? c.kmax_resol : 1;
  G.ra = c.prad > 0 ? c.prad : 6371229.0;
  G.nproc_all = c.nproc > 0 ? c.nproc : 1;
  G.myproc_all = c.myproc > 0 ? c.myproc : 1;
  if (G.myproc_all > G.nproc_all) EMI_FAIL(EMI_ERR_ARG, "emi_init: myproc %d > nproc %d", G.myproc_all, G.nproc_all);
  G.nprtrv = c.nprtrv > 0 ? c.nprtrv : (G.nprtrv_preset > 0 ? G.nprtrv_preset : 1);
  if (G.nproc_all % G.nprtrv != 0)
    EMI_FAIL(EMI_ERR_ARG, "SUMP_TRANS0: NPROC INCONSISTENT WITH NPRTRW (NPROC = %d is not a multiple of NPRTRV = %d)", G.nproc_all, G.nprtrv);
  G.nproc = G.nproc_all / G.nprtrv;                 // NPRTRW
  G.myproc = (G.myproc_all - 1) / G.nprtrv + 1;     // MYSETW
  G.mysetv = (G.myproc_all - 1) % G.nprtrv + 1;     // MYSETV
#ifndef EMI_CPU_EMU
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    EMI_FAIL(EMI_ERR_RUNTIME, "emi_init: no HIP device visible -- libectrans_mi has no CPU path");
  if (c.device >= 0) EMI_CHECK(hipSetDevice(c.device));
#endif
  G.plans.assign(G.max_resol, nullptr);
  const char *pe = getenv("EMI_PROFILE");
  G.profile = pe ? atoi(pe) : 0;
  const char *mb = getenv("EMI_MAX_BATCH");
  if (mb) G.max_batch = atoi(mb);
  G.init = true;
  return EMI_SUCCESS;
}

// ------------------------------------------------------------------------------------------
// SETUP_TRANS
// ------------------------------------------------------------------------------------------

// Direct mixed-radix plan of k_fft_*_mr for a complex length sz: radices A, B, C (B, C may be 1) from EMI_MR_RADICES and the
// workgroup threads for `nf` fields per workgroup, by a lane-time model: a pass of radix R costs ops(R) / R + 15 lane
// operations per point (butterfly; LDS round trip, twiddle, addressing), times the idle share of its last sweep.
static double mr_radix_ops(int R) {
  switch (R) {
    case 2: return 4; case 3: return 14; case 4: return 16; case 5: return 36; case 7: return 66; case 11: return 150;
    case 13: return 204; case 17: return 336; case 19: return 414; case 23: return 594;
    case 6: return 3 * 4 + 2 * 14 + 12; case 8: return 4 * 4 + 2 * 16 + 18; case 9: return 6 * 14 + 24; case 10: return 5 * 4 + 2 * 36 + 24;
    case 12: return 4 * 14 + 3 * 16 + 36; case 14: return 7 * 4 + 2 * 66 + 36; case 15: return 5 * 14 + 3 * 36 + 48; case 16: return 8 * 16 + 54;
    case 18: return 9 * 4 + 2 * 108 + 48; case 20: return 5 * 16 + 4 * 36 + 72; case 21: return 7 * 14 + 3 * 66 + 72;
    default: return -1;
  }
}
static bool mr_choose(int sz, int esz, int fac[3], int &fbk, int &nthr) {
  std::vector<int> rs = {
#define EMI_MR_ROW(r_) r_,
      EMI_MR_RADICES(EMI_MR_ROW)
  };
#ifdef EMI_MR_RADICES_F32
  if (esz == 4)
    for (int r : {EMI_MR_RADICES_F32(EMI_MR_ROW)}) rs.push_back(r);
#endif
#undef EMI_MR_ROW
  double best = 1e300;
  bool found = false;
  auto consider = [&](int A, int B, int C) {
    const int P1 = (B * C) | 1;
    const size_t per_field = (size_t)A * P1 * 2 * esz;
    if (per_field > 160 * 1024 || A * P1 > 65535) return;
    int fb = 16;
    while (fb > 1 && fb * per_field > 40960) fb >>= 1;  // (80 / 160 KiB of fields per workgroup were measured: slower)
    const int R[3] = {A, B, C};
    for (int nt = 128; nt <= 256; nt += 64) {
      double cost = 0;
      for (int ip = 0; ip < 3; ip++) {
        if (R[ip] == 1) continue;
        const long long nb = (long long)fb * (sz / R[ip]);
        const long long sweeps = (nb + nt - 1) / nt;
        cost += (double)(sweeps * nt) / (double)nb * (mr_radix_ops(R[ip]) / R[ip] + 15.0);  // (the 15: 0 ... 60 change the phase by < 1 %)
      }
      // The LDS footprint fixes the workgroups per CU, so a smaller workgroup means fewer waves to hide latencies behind, and the
      // Fourier-space stages run over all threads of the workgroup whatever the passes use.  Measured at TCo1279 (per pair, the
      // rows these kernels carry): one size for all rows 64 threads 66 ms, 128: 51.2, 192: 42.1, 256: 38.8 (first version); final
      // kernels with a premium of 0 / 0.12 / 0.25 / 0.5 per wave less: 28.5 / 26.9 / 25.9 / 25.3 ms (0.5 = 256 threads for every
      // row); 320 / 384 / 512 threads: 35.7 / 35.2 / 31.1 ms.
      cost *= 1.0 + 0.5 * (256 - nt) / 64.0;
      if (cost < best * (1.0 - 1e-9) || (cost <= best * (1.0 + 1e-9) && nt > nthr)) {
        best = cost, found = true;
        fac[0] = A, fac[1] = B, fac[2] = C, fbk = fb, nthr = nt;
      }
    }
  };
  for (int A : rs) {
    if (sz % A) continue;
    const int m1 = sz / A;
    if (m1 == 1) consider(A, 1, 1);
    for (int B : rs) {
      if (m1 % B) continue;
      const int C = m1 / B;
      if (C == 1) consider(A, B, 1);
      for (int Cc : rs)
        if (Cc == C) consider(A, B, C);
    }
  }
  return found;
}

static int build_fft_plans(Plan &P) {
  // Pass 1 (serial, cheap): one plan per distinct row length; table offsets.  The twiddle tables and
  // the digit-reversal table depend only on the work size S and its factor list, so all plans with the
  // same S share one copy (TCo1279: ~20 work sizes for 1280 row lengths -- 2 MB of twiddles that stay in
  // L2 instead of 130 MB); the real-pack twiddles, the chirp and the filter spectrum are per length.
  // Pass 2 fills the tables on the host threads, one task per table.
  struct Shared {
    int S, tw_off, perm_off, ptw_off[14];
    std::vector<int> fac;
    int r16 = 0, mr = 0;
  };
  std::map<int, int> idx;            // row length -> plan
  std::map<int, int> sidx;           // work size -> shared tables
  std::vector<Shared> shared;
  size_t n_tw = 0, n_ptw = 0, n_perm = 0, n_rtw = 0, n_chirp = 0, n_bhat = 0;
  P.planid.assign(P.nlat, 0);
  for (int j = 0; j < P.nlat; j++) {  // local latitudes
    int n = P.nloen[P.lat0 + j];
    auto it = idx.find(n);
    if (it != idx.end()) {
      P.planid[j] = it->second;
      continue;
    }
    FftPlanDev pl{};
    pl.n = n;
    pl.cmode = (n % 2 != 0);
    pl.sz = pl.cmode ? n : n / 2;
    std::vector<int> fac;
    pl.blue = !emi::factorize_smooth(pl.sz, fac);
    pl.S = pl.sz;
    // Register-resident kernels (k_fft_*_r16<R1>, round 3): rows of even length whose Bluestein work length fits 256 R1, R1 from
    // EMI_R16_LIST and at least 8 -- at TCo1279 every row of 1540 to 4098 points.  EMI_FFT_R16=0 keeps the in-place LDS kernels (the
    // tests run the same rows through both families).
    // Rows of that range with a 7-smooth half-length take them too: per
    // point the generic kernels cost 8.7 ps and row, the convolution 5.5 - 6 ps per work point (profiles/r3c_pmc_fft.txt), although it
    // does four times the arithmetic -- every one of those 100-odd row lengths has its own factor list, which the generic kernels
    // walk at run time.
    // Direct mixed-radix kernels (k_fft_*_mr, round 3): even rows whose half-length is a product of at most three radices of
    // EMI_MR_RADICES -- no convolution.  EMI_FFT_MR=0: off (tests: the same rows through the convolution kernels).
    int mr_fbk = 0, mr_nthr = 0, mr_fac[3] = {1, 1, 1};
    if (!pl.cmode && pl.sz >= 2 && !(getenv("EMI_FFT_MR") && atoi(getenv("EMI_FFT_MR")) == 0))
      pl.mr = mr_choose(pl.sz, P.esz, mr_fac, mr_fbk, mr_nthr) ? 1 : 0;
    if (pl.mr) {
      pl.blue = 0;
      fac.assign(mr_fac, mr_fac + 3);
    }
    if (!pl.mr && !pl.cmode && !(getenv("EMI_FFT_R16") && atoi(getenv("EMI_FFT_R16")) == 0)) {
      static const int r1s[] = {
#define EMI_R16_ROW(r_) r_,
          EMI_R16_LIST(EMI_R16_ROW)
#undef EMI_R16_ROW
      };
      const int need = 2 * pl.sz - 1;
      if (need > 1536)  // shorter rows: in-place kernels with several fields per workgroup
        for (int r : r1s)
          if (!pl.r16 && 256 * r >= need) pl.r16 = r;
      // Split kernels (k_fft_*_r16p<R1>, round 6): a row too long for one register-resident convolution whose half-length is even runs as
      // two convolutions of length sz / 2, one after the other, joined by one decimation step (emi_kernels_body.h).  TCo1279 fp64: 44.6
      // against 48.2 ms per pair for the rows of k_fft_*_hot<23 | 24>; TCo2559 fp32: the rows of 4100 .. 8192 points 192 against 249 ms per
      // pair (profiles/r6_fft_experiments.txt).  EMI_FFT_R16S=0: off (tests and A/B runs: the same rows through the in-place LDS kernels).
      const bool use_r16s = !(getenv("EMI_FFT_R16S") && atoi(getenv("EMI_FFT_R16S")) == 0);
      if (!pl.r16 && need > 4096 && pl.sz % 2 == 0 && use_r16s) {
        static const int r1ss[] = {
#define EMI_R16S_ROW(r_) r_,
            EMI_R16S_LIST(EMI_R16S_ROW)
#undef EMI_R16S_ROW
        };
        for (int r : r1ss)
          if (!pl.r16 && 256 * r >= pl.sz - 1) pl.r16 = r, pl.split = 2;
      }
      if (pl.r16) pl.blue = 1;
    }
    if (pl.r16) {
      pl.S = 256 * pl.r16;
      fac.assign(1, pl.r16);  // bookkeeping only: the factor list of these kernels is R1, 16, 16 at compile time
    } else if (pl.blue) {
      pl.S = emi::next_235(2 * pl.sz - 1);
      emi::factorize_smooth(pl.S, fac);
      // Specialised kernels (EMI_HOT_PLAN_LIST): the work length and factor list of the cheapest one that is long
      // enough -- cost model S x (passes + 1), as next_235 -- replace the generic choice when they cost no more: that
      // covers the merged tails (8, 8, 8, 6 | 9 | 10: one LDS round trip fewer than 8, 8, 8, 2, 3 ...) and any work length of the
      // list that is not of the form 2^a {1,3,5,9,15} (there are none at present, see emi_types.h).
      // Even NLOEN only: odd rows run the generic kernels, which have no composite butterflies.  The plan must be the
      // one for the fields-per-workgroup this work length gets (the 40-KiB rule further down).
      if (!pl.cmode && !getenv("EMI_FFT_NO_HOT")) {
        static const int hp[][9] = {
#define EMI_HOT_ROW(pc_, S_, nf_, a_, b_, c_, d_, e_, nfl_) {pc_, S_, nf_, a_, b_, c_, d_, e_, nfl_},
            EMI_HOT_PLAN_LIST(EMI_HOT_ROW)
#undef EMI_HOT_ROW
        };
        long long best = (long long)pl.S * (long long)(fac.size() + 1);
        const int *pick = nullptr;
        for (const auto &r : hp) {
          if (r[1] < 2 * pl.sz - 1) continue;
          int fbk_r = 16;
          while (fbk_r > 1 && (size_t)fbk_r * FFT_LDS_ELEMS(r[1]) * 2 * P.esz > 40960) fbk_r >>= 1;
          if (fbk_r < r[8]) continue;  // (the fp32 library has room for twice the fields: it runs the plan with the list's count, below)
          const long long cost = (long long)r[1] * (r[2] + 1);
          if (cost <= best) {  // list order: a merged tail (ids 22-27) comes after the plain list of its length
            best = cost;
            pick = r;
          }
        }
        if (pick) {
          pl.S = pick[1];
          fac.assign(pick + 3, pick + 3 + pick[2]);
        }
      }
    }
    if (pl.S > 65535 || fac.size() > 14) EMI_FAIL(EMI_ERR_UNSUPPORTED, "FFT length %d not supported (work size %d)", n, pl.S);
    pl.nfac = (int)fac.size();
    for (int i = 0; i < pl.nfac; i++) pl.fac[i] = fac[i];
    const int skey = pl.S * 16 + (pl.r16 ? 15 : (pl.mr ? 14 : pl.nfac));  // the tables depend on the factor list: merged and plain lists differ in length
    auto si = sidx.find(skey);
    if (si == sidx.end()) {
      Shared sh{};
      sh.S = pl.S;
      sh.fac = fac;
      sh.r16 = pl.r16;
      sh.mr = pl.mr;
      sh.tw_off = (int)n_tw;
      sh.perm_off = (int)n_perm;
      if (pl.mr) {
        // twiddle bases of pass 2, [k1]: w^(C k1), and of pass 3, [k1 + A k2]: w^(k1 + A k2) (input r of a butterfly times base^r);
        // perm[k] = LDS position of coefficient k
        const int A = fac[0], B = fac[1];
        sh.ptw_off[0] = (int)n_ptw;
        n_ptw += (size_t)A;
        sh.ptw_off[1] = (int)n_ptw;
        n_ptw += (size_t)A * B;
        sh.ptw_off[2] = (int)n_ptw;
        n_perm += pl.S;
      } else if (pl.r16) {
        // the digit twiddles of A1 / B1: rows 0..2 = w^(t qa), qa = 1..3; rows 3..6 = w^(4 t qb), qb = 1..4; w = exp(-2 pi i / S), t < 256
        sh.ptw_off[0] = (int)n_ptw;
        n_ptw += 7 * 256;
      } else {
      n_tw += pl.S;
      n_perm += pl.S;
      long long lenp = 1;
      for (int ip = 0; ip < pl.nfac; ip++) {
        sh.ptw_off[ip] = (int)n_ptw;
        if (lenp > 1) n_ptw += (size_t)(fac[ip] - 1) * lenp;
        lenp *= fac[ip];
      }
      }
      si = sidx.emplace(skey, (int)shared.size()).first;
      shared.push_back(sh);
    }
    const Shared &sh = shared[si->second];
    pl.tw_off = sh.tw_off;
    pl.perm_off = sh.perm_off;
    for (int ip = 0; ip < pl.nfac; ip++) pl.ptw_off[ip] = sh.ptw_off[ip];
    pl.rtw_off = (int)n_rtw;
    n_rtw += pl.sz + 1;
    if (pl.blue) {
      pl.chirp_off = (int)n_chirp;
      n_chirp += pl.split ? pl.sz / 2 : pl.sz;
      pl.bhat_off = (int)n_bhat;
      n_bhat += pl.S;
    }
    if (n_tw > 0x7fffffffULL || n_ptw > 0x7fffffffULL || n_bhat > 0x7fffffffULL) EMI_FAIL(EMI_ERR_UNSUPPORTED, "FFT tables too large");
    // fields per workgroup: as many as fit ~40 KiB of LDS (power of two, <= 16); longer rows get one
    // field per workgroup and more threads.  Threads per workgroup follow the LDS footprint: 256 up to
    // 40 KiB, 512 up to 80 KiB, else 1024 -- i.e. always 16 waves per CU at 128 VGPRs.  (One thread per
    // radix-8 butterfly, S/8, removes the idle lanes of the second sweep at S = 2560/3072/4608/5120
    // but makes workgroups of 5, 9 and 10 waves, of which only one fits the 16-wave budget: measured
    // 25 % slower.)
    size_t per_field = (size_t)FFT_LDS_ELEMS(pl.S) * 2 * P.esz;
    int fbk = 16;
    while (fbk > 1 && fbk * per_field > 40960) fbk >>= 1;
    pl.fbk = fbk;
    size_t need = fbk * per_field;
    // a row whose work array exceeds the 160 KiB of LDS (fp64: more than 10240 complex points -- the four longest rows of
    // TCo2559, or any caller grid: the reference takes any KLOEN, ftdir_mod.F90:67-84) runs the same passes on a slice
    // of a global scratch buffer (k_fft_*_gm): slow per row, but such rows are few
    const bool gmem = need > 160 * 1024;
    int nthr = need <= 40960 ? 256 : (need <= 81920 ? 512 : 1024);
    // specialised kernel for this work length?  (Bluestein, even NLOEN, one field per workgroup)
    int hot = 0;
    if (pl.blue && !pl.cmode && !gmem && !getenv("EMI_FFT_NO_HOT")) {
      static const int hp[][9] = {
#define EMI_HOT_ROW(pc_, S_, nf_, a_, b_, c_, d_, e_, nfl_) {pc_, S_, nf_, a_, b_, c_, d_, e_, nfl_},
          EMI_HOT_PLAN_LIST(EMI_HOT_ROW)
#undef EMI_HOT_ROW
      };
      // The list's fields-per-workgroup are those of the fp64 library.  The fp32 library could hold twice as many in its 40 KiB, for which no
      // kernel is compiled: until round 6 its short rows (work lengths <= 1536) therefore fell through to the generic kernels -- 14.7 of the
      // 117 ms of FFT per pair at TCo1279 (profiles/r6_pmc_fft_fp32.txt: k_fft_dir 8.35 + k_fft_inv 6.32 ms).  They take the list's plan with
      // the list's field count now (half the LDS per workgroup; the wave budget of the CU is what limits both).
      int hot_fbk = 0;
      for (const auto &r : hp) {
        bool same = r[1] == pl.S && r[2] == pl.nfac && r[8] <= fbk;
        for (int i = 0; same && i < pl.nfac; i++) same = r[3 + i] == pl.fac[i];
        if (same) hot = r[0], hot_fbk = r[8];
      }
      if (hot) {
        fbk = hot_fbk;
        pl.fbk = fbk;
        need = fbk * per_field;
        nthr = need <= 40960 ? 256 : (need <= 81920 ? 512 : 1024);  // = hot_threads(pc): what the specialised kernels are compiled for
      }
    }
    if (pl.r16) {  // one field per workgroup, 256 or 320 threads, one plane + the 240 small twiddles of LDS
      hot = 0;
      fbk = 1;
      pl.fbk = 1;
      nthr = 16 * pl.r16 > 256 ? roundup(16 * pl.r16, 64) : 256;
    }
    if (pl.mr) {
      hot = 0;
      fbk = mr_fbk;
      pl.fbk = fbk;
      nthr = mr_nthr;
    }
    for (int i = 0; i < pl.nfac; i++)
      if (!hot && !pl.r16 && !pl.mr && (pl.fac[i] == 6 || pl.fac[i] > 8)) EMI_FAIL(EMI_ERR_RUNTIME, "internal: composite FFT radix %d without a specialised kernel (length %d)", pl.fac[i], n);
    int cls = -1;
    for (size_t c = 0; c < P.fclass.size(); c++)
      if (P.fclass[c].nthr == nthr && P.fclass[c].fbk == fbk && P.fclass[c].hot == hot && P.fclass[c].r16 == pl.r16 && P.fclass[c].split == pl.split && P.fclass[c].mr == pl.mr && P.fclass[c].gmem == (gmem && !pl.mr ? 1 : 0)) cls = (int)c;
    if (cls < 0) {
      cls = (int)P.fclass.size();
      P.fclass.emplace_back();
      P.fclass[cls].nthr = nthr;
      P.fclass[cls].fbk = fbk;
      P.fclass[cls].hot = hot;
      P.fclass[cls].r16 = pl.r16;
      P.fclass[cls].split = pl.split;
      P.fclass[cls].mr = pl.mr;
      P.fclass[cls].gmem = gmem && !pl.mr ? 1 : 0;
    }
    pl.lds_class = cls;
    int id = (int)P.fplans.size();
    P.fplans.push_back(pl);
    idx[n] = id;
    P.planid[j] = id;
  }
  // ---- pass 2: fill the tables
  std::vector<d2> tw(n_tw), rtw(n_rtw), chirp(n_chirp), bhat(n_bhat), ptw(n_ptw);
  std::vector<uint16_t> perm(n_perm);
  emi::parallel_for((int)shared.size(), [&](int is) {
    const Shared &sh = shared[is];
    const int S = sh.S;
    if (sh.mr) {
      const int A = sh.fac[0], B = sh.fac[1], C = sh.fac[2], P1 = (B * C) | 1;
      auto w = [&](long long e) {
        long double a = 2.0L * (long double)M_PIl * (long double)(e % S) / (long double)S;
        return d2{(double)cosl(a), (double)-sinl(a)};
      };
      d2 *t1 = ptw.data() + sh.ptw_off[0], *t2 = ptw.data() + sh.ptw_off[1];
      for (int k1 = 0; k1 < A; k1++) t1[k1] = w((long long)C * k1);
      for (int q = 0; q < A * B; q++) t2[q] = w((long long)q);
      for (int k = 0; k < S; k++) perm[sh.perm_off + k] = (uint16_t)((k % A) * P1 + ((k / A) % B) * C + k / (A * B));
      return;
    }
    if (sh.r16) {
      d2 *dst = ptw.data() + sh.ptw_off[0];
      for (int row = 0; row < 7; row++) {
        const int mult = row < 3 ? row + 1 : 4 * (row - 2);
        for (int t = 0; t < 256; t++) {
          long double a = 2.0L * (long double)M_PIl * (long double)(((long long)t * mult) % S) / (long double)S;
          *dst++ = d2{(double)cosl(a), (double)-sinl(a)};
        }
      }
      return;
    }
    for (int k = 0; k < S; k++) {
      long double a = 2.0L * (long double)M_PIl * (long double)k / (long double)S;
      tw[sh.tw_off + k] = d2{(double)cosl(a), (double)-sinl(a)};
    }
    // per-pass twiddle tables, [t-1][j] with j fastest (coalesced reads)
    long long lenp = 1;
    for (size_t ip = 0; ip < sh.fac.size(); ip++) {
      const int R = sh.fac[ip];
      d2 *dst = ptw.data() + sh.ptw_off[ip];
      if (lenp > 1)
        for (int t = 1; t < R; t++)
          for (long long j = 0; j < lenp; j++) {
            long double a = 2.0L * (long double)M_PIl * (long double)((j * t) % (lenp * R)) / (long double)(lenp * R);
            *dst++ = d2{(double)cosl(a), (double)-sinl(a)};
          }
      lenp *= R;
    }
    std::vector<uint16_t> pm;
    emi::dit_positions(S, sh.fac, pm);
    std::copy(pm.begin(), pm.end(), perm.begin() + sh.perm_off);
  });
  // longest rows first: their O(L * sz) filter sums dominate
  std::vector<int> order(P.fplans.size());
  for (size_t i = 0; i < order.size(); i++) order[i] = (int)i;
  std::sort(order.begin(), order.end(), [&](int a, int b) { return P.fplans[a].n > P.fplans[b].n; });
  emi::parallel_for((int)order.size(), [&](int io) {
    const FftPlanDev &pl = P.fplans[order[io]];
    const int n = pl.n;
    const double tpi = 2.0 * M_PI;
    for (int k = 0; k <= pl.sz; k++) {
      double a = tpi * (double)k / (double)n;
      rtw[pl.rtw_off + k] = d2{std::cos(a), -std::sin(a)};
    }
    if (!pl.blue) return;
    d2 *c = chirp.data() + pl.chirp_off;
    const int csz = pl.split ? pl.sz / 2 : pl.sz;  // length of the chirp-z transform(s) of the row: the split kernels run two of sz / 2
    for (int k = 0; k < csz; k++) {
      long long k2 = ((long long)k * k) % (2LL * csz);
      double a = M_PI * (double)k2 / (double)csz;
      c[k] = d2{std::cos(a), -std::sin(a)};  // exp(-i pi k^2/csz)
    }
    // filter b_j = conj(c_|j|) wrapped to length L; Bhat = DFT_L(b) (direct O(L*sz) sum in
    // long double: setup only, keeps the table accurate to ~1e-17), stored at the DIT positions
    const int L = pl.S, r0 = pl.fac[0];
    const uint16_t *pm = pl.r16 ? nullptr : perm.data() + pl.perm_off;
    std::vector<long double> cr(L);
    for (int k = 0; k < L; k++) cr[k] = cosl(2.0L * (long double)M_PIl * (long double)k / (long double)L);
    d2 *bh = bhat.data() + pl.bhat_off;
    for (int k = 0; k < L; k++) {
      long double sr = c[0].x, si = -c[0].y;
      long long jk = 0;
      for (int jj = 1; jj < csz; jj++) {
        // b_j + b_{L-j} term: conj(c_j) * (w^{jk} + w^{-jk}) = conj(c_j) * 2 cos(2 pi j k/L)
        jk += k;
        if (jk >= L) jk -= L;
        long double cs = 2.0L * cr[jk];
        sr += (long double)c[jj].x * cs;
        si += -(long double)c[jj].y * cs;
      }
      // position p = pm[k] of the DIT-ordered spectrum belongs to the middle butterfly q = p / R0 as its
      // element t = p % R0; stored [t][q] so that a wave reads its filter values coalesced
      if (pl.r16) {
        // k = k0 + R1 (k1 + 16 k2) sits in register k2 of thread 16 k0 + k1 of the fused middle pass: table [k2][16 k0 + k1]
        const int k0 = k % pl.r16, kk = k / pl.r16, k1 = kk % 16, k2 = kk / 16;
        bh[(size_t)k2 * (16 * pl.r16) + 16 * k0 + k1] = d2{(double)(sr / (long double)L), (double)(si / (long double)L)};  // the 1/S of the convolution folded in
        continue;
      }
      const int p = pm[k];
      bh[(size_t)(p % r0) * (L / r0) + p / r0] = d2{(double)sr, (double)si};
    }
  });
  for (int j = 0; j < P.nlat; j++) {
    const FftPlanDev &pl = P.fplans[P.planid[j]];
    FftClass &fc = P.fclass[pl.lds_class];
    fc.lats.push_back(j);
    if (fc.gmem)
      fc.gm_elems = std::max(fc.gm_elems, (long long)pl.fbk * FFT_LDS_ELEMS(pl.S));
    else if (fc.mr)
      fc.lds = std::max(fc.lds, (size_t)pl.fbk * pl.fac[0] * ((pl.fac[1] * pl.fac[2]) | 1) * 2 * P.esz);
    else if (fc.r16)
      fc.lds = (size_t)fc.r16 * 272 * 8 + (size_t)((fc.split ? 128 * fc.r16 : 0) + 240) * 2 * P.esz;  // plane (+ the parked half of k_fft_*_r16p) + small twiddles
    else
      fc.lds = std::max(fc.lds, (size_t)pl.fbk * FFT_LDS_ELEMS(pl.S) * 2 * P.esz);
  }
  // exp(-2 pi i c k1 / 256), [k1 - 1][c]: the small twiddles of k_fft_*_r16
  std::vector<d2> tw256(15 * 16);
  for (int k1 = 1; k1 < 16; k1++)
    for (int c = 0; c < 16; c++) {
      long double a = 2.0L * (long double)M_PIl * (long double)(c * k1) / 256.0L;
      tw256[(k1 - 1) * 16 + c] = d2{(double)cosl(a), (double)-sinl(a)};
    }
  void *d_tw, *d_rtw, *d_chirp, *d_bhat, *d_ptw, *d_tw256;
  uint16_t *d_perm;
  FftPlanDev *d_plans;
  int *d_planid;
  // the tables are computed in (long) double and rounded once for the fp32 library
  auto upload_c = [&](const std::vector<d2> &v, void **d) {
    if (P.esz == 8) return upload(v, (d2 **)d);
    std::vector<f2> w(v.size());
    for (size_t i = 0; i < v.size(); i++) w[i] = f2{(float)v[i].x, (float)v[i].y};
    return upload(w, (f2 **)d);
  };
  if (upload_c(tw, &d_tw) || upload_c(ptw, &d_ptw) || upload_c(rtw, &d_rtw) || upload_c(chirp, &d_chirp) || upload_c(bhat, &d_bhat) || upload_c(tw256, &d_tw256) || upload(perm, &d_perm) ||
      upload(P.fplans, &d_plans) || upload(P.planid, &d_planid))
    return EMI_ERR_RUNTIME;
  for (void *p : {d_tw, d_ptw, d_rtw, d_chirp, d_bhat, d_tw256, (void *)d_perm, (void *)d_plans, (void *)d_planid})
    P.dev_allocs.push_back(p);
  P.ftab.tw = d_tw;
  P.ftab.ptw = d_ptw;
  P.ftab.rtw = d_rtw;
  P.ftab.chirp = d_chirp;
  P.ftab.bhat = d_bhat;
  P.ftab.tw256 = d_tw256;
  P.ftab.perm = d_perm;
  P.ftab.plans = d_plans;
  P.ftab.planid = d_planid;
  for (FftClass &fc : P.fclass) {
    if (upload(fc.lats, &fc.d_lats)) return EMI_ERR_RUNTIME;
    P.dev_allocs.push_back(fc.d_lats);
    std::vector<FftRowDev> rows(fc.lats.size());
    for (size_t i = 0; i < fc.lats.size(); i++) {
      const int j = fc.lats[i];
      const FftPlanDev &pl = P.fplans[P.planid[j]];
      FftRowDev &r = rows[i];
      r.lat = j, r.planid = P.planid[j], r.n = pl.n, r.sz = pl.sz;
      r.nmen = P.l_nmen[j], r.fb0 = P.l_fbase[j], r.gpoff = P.l_gpoff[j], r.chirp_off = pl.chirp_off;
      r.rtw_off = pl.rtw_off, r.bhat_off = pl.bhat_off, r.ptw_off0 = pl.ptw_off[0];
      r.mr_abc = pl.mr ? (pl.fac[0] | (pl.fac[1] << 8) | (pl.fac[2] << 16) | (pl.fbk << 24)) : 0;
      r.racthe = P.racthe[P.lat0 + j], r.rw = P.rw[P.lat0 + j];
    }
    if (upload(rows, &fc.d_rows)) return EMI_ERR_RUNTIME;
    P.dev_allocs.push_back(fc.d_rows);
  }
  return 0;
}


// ------------------------------------------------------------------------------------------
// CDIO_LEGPOL: the reference's Legendre-polynomial file / memory segment (NPROC = 1 only,
// setup_trans.F90:360-384).  Byte format of write_legpol_mod.F90:66-158 / read_legpol_mod.F90:78-215:
//   4 x int32   'LEGP' 'OL  ' NSMAX NDGNH
//   2*NDGNH x int32   (NLOEN(jgl), NMEN(jgl)), jgl = 1..NDGNH
//   per wavenumber in MYMS order:  RPNMA(IDGLU, ILA) then RPNMS(IDGLU, ILS), 8-byte reals, column-major,
//   IDGLU = MIN(NDGNH, NDGLU(m)), ILA = (NSMAX-m+2)/2, ILS = (NSMAX-m+3)/2, columns with n descending
// ('LEGPOLBF' files carry butterfly-compressed matrices of the FLT option, which this library refuses.)
// ------------------------------------------------------------------------------------------
struct LegpolSource {
  const char *base = nullptr;
  size_t len = 0;
  void *map = nullptr;
  size_t maplen = 0;
  std::vector<size_t> offA, offS;  // byte offsets of RPNMA / RPNMS per wavenumber
  ~LegpolSource() {
    if (map) munmap(map, maplen);
  }
};

static int legpol_open(LegpolSource &src, const emi_legpol_io_t *io, bool membuf) {
  if (membuf) {
    if (!io->ptr) EMI_FAIL(EMI_ERR_ARG, "SETUP_TRANS: KLEGPOLPTR NULL POINTER");
    if (!io->len) EMI_FAIL(EMI_ERR_ARG, "SETUP_TRANS: KLEGPOLPTR_LEN ARGUMENT MISSING");
    src.base = (const char *)io->ptr;
    src.len = io->len;
    return 0;
  }
  if (!io->fname || !io->fname[0]) EMI_FAIL(EMI_ERR_ARG, "SETUP_TRANS: CDLEGPOLFNAME ARGUMENT MISSING");
  int fd = open(io->fname, O_RDONLY);
  struct stat st;
  if (fd < 0 || fstat(fd, &st) != 0) {
    if (fd >= 0) close(fd);
    EMI_FAIL(EMI_ERR_ARG, "READ_LEGPOL: BYTES_IO_OPEN FAILED (%s)", io->fname);
  }
  src.maplen = (size_t)st.st_size;
  src.map = src.maplen ? mmap(nullptr, src.maplen, PROT_READ, MAP_PRIVATE, fd, 0) : nullptr;
  close(fd);
  if (src.map == MAP_FAILED || !src.map) {
    src.map = nullptr;
    EMI_FAIL(EMI_ERR_ARG, "READ_LEGPOL:BYTES_IO_READ FAILED (%s)", io->fname);
  }
  src.base = (const char *)src.map;
  src.len = src.maplen;
  return 0;
}

// header checks of read_legpol_mod.F90:78-118 and the per-wavenumber offsets
static int legpol_index(LegpolSource &src, int nsmax, int ndgnh, const std::vector<int> &nloen, const std::vector<int> &nmen,
                        const std::vector<int> &ndglu) {
  const size_t head = 16 + (size_t)8 * ndgnh;
  if (src.len < 16) EMI_FAIL(EMI_ERR_ARG, "READ_LEGPOL:BYTES_IO_READ FAILED (segment shorter than its header)");
  int ib[4];
  memcpy(ib, src.base, 16);
  if (memcmp(src.base, "LEGPOL  ", 8) != 0) EMI_FAIL(EMI_ERR_ARG, "READ_LEGPOL:WRONG LABEL");
  if (ib[2] != nsmax) EMI_FAIL(EMI_ERR_ARG, "READ_LEGPOL:WRONG SPECTRAL TRUNCATION");
  if (ib[3] != ndgnh) EMI_FAIL(EMI_ERR_ARG, "READ_LEGPOL:WRONG NO OF GAUSSIAN LATITUDES");
  if (src.len < head) EMI_FAIL(EMI_ERR_ARG, "READ_LEGPOL:BYTES_IO_READ FAILED (segment shorter than its latitude table)");
  for (int j = 0; j < ndgnh; j++) {
    int v[2];
    memcpy(v, src.base + 16 + (size_t)8 * j, 8);
    if (v[0] != nloen[j]) EMI_FAIL(EMI_ERR_ARG, "READ_LEGPOL:WRONG NLOEN (latitude %d: %d, file %d)", j + 1, nloen[j], v[0]);
    if (v[1] != nmen[j]) EMI_FAIL(EMI_ERR_ARG, "READ_LEGPOL:WRONG NMEN (latitude %d: %d, file %d)", j + 1, nmen[j], v[1]);
  }
  size_t off = head;
  src.offA.assign(nsmax + 1, 0);
  src.offS.assign(nsmax + 1, 0);
  for (int m = 0; m <= nsmax; m++) {
    const size_t nd = (size_t)std::min(ndgnh, ndglu[m]);
    src.offA[m] = off;
    off += nd * (size_t)((nsmax - m + 2) / 2) * 8;
    src.offS[m] = off;
    off += nd * (size_t)((nsmax - m + 3) / 2) * 8;
  }
  if (off > src.len) EMI_FAIL(EMI_ERR_ARG, "READ_LEGPOL:BYTES_IO_READ FAILED (%zu bytes needed, %zu present)", off, src.len);
  return 0;
}

static int legpol_write(int kresol, const char *fname);

extern "C" int emi_setup_legpol(const emi_setup_t *cfg, const emi_legpol_io_t *io, int *kresol) {
  EmiRange rg_setup(EMI_LBL_SETUP);  // GSTATS 2
  if (!G.init) EMI_FAIL(EMI_ERR_STATE, "SETUP_TRANS: SETUP_TRANS0 HAS TO BE CALLED BEFORE SETUP_TRANS");
  if (!cfg) EMI_FAIL(EMI_ERR_ARG, "emi_setup: null config");
  if (cfg->kdgl <= 0 || cfg->kdgl % 2 != 0) EMI_FAIL(EMI_ERR_ARG, "SETUP_TRANS: KDGL IS NOT A POSITIVE, EVEN NUMBER");
  if (cfg->lduseflt) EMI_FAIL(EMI_ERR_UNSUPPORTED, "SETUP_TRANS: LDUSEFLT not supported (as gpu/external/setup_trans.F90:442)");
  if (cfg->ldll) EMI_FAIL(EMI_ERR_UNSUPPORTED, "SETUP_TRANS: LDLL lat-lon grids not supported (as gpu/external/setup_trans.F90:309)");
  if (cfg->ldstretch) EMI_FAIL(EMI_ERR_UNSUPPORTED, "SETUP_TRANS: PSTRET stretching not supported");
  if (cfg->precision != 0 && cfg->precision != 8 && cfg->precision != 4)
    EMI_FAIL(EMI_ERR_ARG, "emi_setup: precision must be 8 (fp64, the _dp library) or 4 (fp32, _sp), got %d", cfg->precision);
  if (cfg->ksmax < 0) EMI_FAIL(EMI_ERR_ARG, "SETUP_TRANS: KSMAX < 0");
  // CDIO_LEGPOL (setup_trans.F90:360-384)
  enum { LP_NONE, LP_READF, LP_WRITEF, LP_MEMBUF } lp_mode = LP_NONE;
  if (io && io->io && io->io[0]) {
    std::string mode(io->io);
    while (!mode.empty() && mode.back() == ' ') mode.pop_back();
    if (G.nproc_all > 1) EMI_FAIL(EMI_ERR_UNSUPPORTED, "SETUP_TRANS:CDIO_LEGPOL OPTIONS ONLY FOR NPROC=1 ");
    if (mode == "readf" || mode == "READF")
      lp_mode = LP_READF;
    else if (mode == "writef" || mode == "WRITEF")
      lp_mode = LP_WRITEF;
    else if (mode == "membuf" || mode == "MEMBUF")
      lp_mode = LP_MEMBUF;
    else
      EMI_FAIL(EMI_ERR_ARG, "SETUP_TRANS:CDIO_LEGPOL UNKNOWN METHOD (%s)", mode.c_str());
    if (lp_mode != LP_MEMBUF && (!io->fname || !io->fname[0])) EMI_FAIL(EMI_ERR_ARG, "SETUP_TRANS: CDLEGPOLFNAME ARGUMENT MISSING");
  }
  LegpolSource lp_src;
  if ((lp_mode == LP_READF || lp_mode == LP_MEMBUF) && legpol_open(lp_src, io, lp_mode == LP_MEMBUF)) return EMI_ERR_ARG;
  const bool lp_read = lp_src.base != nullptr;
  int slot = -1;
  for (int i = 0; i < G.max_resol; i++)
    if (!G.plans[i] || !G.plans[i]->active) {
      slot = i;
      break;
    }
  if (slot < 0) EMI_FAIL(EMI_ERR_STATE, "SETUP_TRANS:IDEF_RESOL > NMAX_RESOL");
  Plan *pp = new Plan();
  Plan &P = *pp;
  P.nsmax = cfg->ksmax;
  P.ndgl = cfg->kdgl;
  P.ndgnh = (P.ndgl + 1) / 2;
  P.ra = G.ra;
  P.esz = cfg->precision == 4 ? 4 : 8;
  P.nproc = G.nproc;
  P.me = G.myproc - 1;
  const int N = P.nsmax, L = P.ndgl, NP = P.nproc, me = P.me;
  if (G.nproc_all > 1 && !G.a2a) {
    delete pp;
    EMI_FAIL(EMI_ERR_STATE, "SETUP_TRANS: %d tasks but no all-to-all-v hook registered (emi_set_alltoallv)", G.nproc_all);
  }
  if (NP > L / 2 || NP > N + 1) {
    delete pp;
    EMI_FAIL(EMI_ERR_ARG, "SETUP_TRANS: too many tasks (%d) for NDGL=%d, NSMAX=%d", NP, L, N);
  }
  int ndlon = cfg->kdlon > 0 ? cfg->kdlon : 2 * L;
  P.nloen.assign(L, ndlon);
  if (cfg->kloen) {
    ndlon = 0;
    for (int j = 0; j < L; j++) {
      if (cfg->kloen[j] <= 0) {
        delete pp;
        EMI_FAIL(EMI_ERR_ARG, "SETUP_TRANS: KLOEN INVALID (ONE or MORE POINTS <= 0)");
      }
      ndlon = std::max(ndlon, cfg->kloen[j]);
    }
    for (int j = 0; j < L; j++) {
      P.nloen[j] = cfg->kloen[j];
      if (cfg->kloen[j] != ndlon) P.reduced = true;
    }
  }
  P.nspec2g = (N + 1) * (N + 2);
  std::vector<long long> cum(L + 1, 0);
  for (int j = 0; j < L; j++) cum[j + 1] = cum[j] + P.nloen[j];
  if (cum[L] > 2000000000LL) {
    delete pp;
    EMI_FAIL(EMI_ERR_UNSUPPORTED, "grid too large for 32-bit point offsets");
  }
  P.ngptotg = (int)cum[L];
  // Gaussian latitudes / weights, cos^2, 1/(a cos)  (suleg_mod.F90:264-293, 386-394)
  // EMI_SETUP_TIMING=1 prints where SETUP_TRANS spends its wall time
  const bool timing = getenv("EMI_SETUP_TIMING") && atoi(getenv("EMI_SETUP_TIMING"));
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_ph = now();
  auto phase = [&](const char *what) {
    if (timing) fprintf(stderr, "[emi_setup] %-28s %8.3f s\n", what, now() - t_ph);
    t_ph = now();
  };
  emi::gauss_latitudes(L, P.rmu, P.rw);
  phase("gaussian latitudes");
  P.cos2.assign(L, 0.0);
  P.racthe.assign(L, 0.0);
  for (int j = 0; j < L; j++) {
    double th = std::asin(P.rmu[j]), c = std::cos(th);
    P.cos2[j] = c * c;
    P.racthe[j] = 1.0 / c / P.ra;
  }
  emi::wavenumber_cutoffs(N, L, P.nloen, P.reduced, P.cos2, P.nmen, P.ndglu);
  if (lp_read && legpol_index(lp_src, N, P.ndgnh, P.nloen, P.nmen, P.ndglu)) {
    delete pp;
    return EMI_ERR_ARG;
  }

  // ---- distribution over tasks
  // wavenumbers: the reference's zig-zag W-set assignment (suwavedi_mod.F90:118-137)
  P.procm.assign(N + 1, 0);
  {
    int ik = 0, ind = 1;
    for (int m = 0; m <= N; m++) {
      ik += ind;
      if (ik > NP) {
        ik = NP;
        ind = -1;
      } else if (ik < 1) {
        ik = 1;
        ind = 1;
      }
      P.procm[m] = ik - 1;
    }
  }
  // latitudes: contiguous bands of whole latitudes (the Fourier-space distribution of sumplatb_mod.F90 with
  // LDSPLIT=.FALSE., which weighs a latitude by NLOEN); the grid-point distribution is chosen identical, so
  // TRGTOL/TRLTOG stay local copies.  The weight here is the measured cost of a row in the FFT kernels (ps per
  // field, both directions, TCo1279 kernel statistics by work length): 20 ps per point up to NLOEN = 1500, falling
  // linearly to 16.9 ps at 4900, plus 1.75 ns per row -- with equal numbers of points the polar tasks of an 8-task
  // TCo1279 job spent 18 % longer in the FFT phase than the equatorial ones (26.8 against 22.6 ms) and set the
  // job's pace (the reference's weight is NLOEN).
  std::vector<long long> wcum(L + 1, 0);
  for (int j = 0; j < L; j++) {
    const double n = (double)P.nloen[j];
    const double w = n * (n <= 1500.0 ? 20.0 : std::max(15.0, 20.0 - (n - 1500.0) * 0.00091)) + 1750.0;
    wcum[j + 1] = wcum[j] + (long long)(w + 0.5);
  }
  P.latlo.assign(NP + 1, 0);
  for (int r = 1; r < NP; r++) {
    long long target = wcum[L] * r / NP;
    int j = (int)(std::lower_bound(wcum.begin(), wcum.end(), target) - wcum.begin());
    j = std::max(j, P.latlo[r - 1] + 1);
    j = std::min(j, L - (NP - r));
    P.latlo[r] = j;
  }
  P.latlo[NP] = L;
  // V-sets: every band in NPRTRV sub-bands of whole latitudes with (nearly) equal numbers of points -- the grid-point
  // distribution over all NPROC tasks (the reference: sumplat with NPRGPNS = NPROC bands; whole latitudes here as for the bands)
  P.nprv = G.nprtrv;
  P.mev = G.mysetv - 1;
  if (P.nprv > 1) {
    P.vlat.assign((size_t)NP * P.nprv + 1, L);
    for (int w = 0; w < NP; w++) {
      const int a0 = P.latlo[w], a1 = P.latlo[w + 1];
      if (a1 - a0 < P.nprv) {
        delete pp;
        EMI_FAIL(EMI_ERR_ARG, "SETUP_TRANS: too many tasks: band %d has %d latitudes for %d V-sets", w + 1, a1 - a0, P.nprv);
      }
      P.vlat[(size_t)w * P.nprv] = a0;
      for (int v = 1; v < P.nprv; v++) {
        const long long target = cum[a0] + (cum[a1] - cum[a0]) * v / P.nprv;
        int j = (int)(std::lower_bound(cum.begin() + a0, cum.begin() + a1, target) - cum.begin());
        j = std::max(j, P.vlat[(size_t)w * P.nprv + v - 1] + 1);
        j = std::min(j, a1 - (P.nprv - v));
        P.vlat[(size_t)w * P.nprv + v] = j;
      }
    }
  }
  P.lat0 = P.latlo[me];
  P.nlat = P.latlo[me + 1] - P.lat0;
  P.ngptot = (int)(cum[P.latlo[me + 1]] - cum[P.lat0]);
  for (int m = 0; m <= N; m++)
    if (P.procm[m] == me) P.mval.push_back(m);
  P.nump = (int)P.mval.size();
  const int NU = P.nump, NL = P.nlat;
  auto band_of = [&](int lat) { return (int)(std::upper_bound(P.latlo.begin(), P.latlo.end(), lat) - P.latlo.begin()) - 1; };

  // ---- local index tables
  P.nasm0.assign(NU, 0);
  P.wbase.assign(NU + 1, 0);
  P.wrows.assign(NU, 0);
  P.ldp.assign(NU, 0);
  P.ldk.assign(NU, 0);
  P.offS.assign(NU, 0);
  P.offA.assign(NU, 0);
  P.offTS.assign(NU, 0);
  P.offTA.assign(NU, 0);
  P.lattile_pref.assign(NU + 1, 0);
  P.ktile_pref.assign(NU + 1, 0);
  P.lbase.assign(NU + 1, 0);
  std::vector<int> ebase(NU, 0);
  std::vector<double> eps, specw;
  {
    int ipos = 0;
    long long poff = 0, ptoff = 0;
    for (int ml = 0; ml < NU; ml++) {
      const int m = P.mval[ml];
      P.nasm0[ml] = ipos;  // 0-based (D%NASM0 - 1, suwavedi_mod.F90:128-133)
      ipos += (N - m + 1) * 2;
      for (int n = m; n <= N; n++) {  // SPNORMD weights (spnormd_mod.F90:40-57)
        specw.push_back(m == 0 ? 1.0 : 2.0);
        specw.push_back(m == 0 ? 0.0 : 2.0);
      }
      P.wrows[ml] = roundup(N + 2 - m, P.esz == 4 ? 32 : 16);  // whole stages of k_leg_inv: 8 (fp64) | 16 (fp32) rows per parity (LG_KR)
      P.wbase[ml + 1] = P.wbase[ml] + P.wrows[ml];
      int nd = std::min(P.ndgnh, P.ndglu[m]);
      P.lbase[ml + 1] = P.lbase[ml] + nd;
      P.ldp[ml] = roundup(std::max(nd, 1), 64);
      long long pan = (long long)(P.wrows[ml] / 2) * P.ldp[ml];
      P.offS[ml] = poff;
      P.offA[ml] = poff + pan;
      poff += 2 * pan;
      P.ldk[ml] = roundup(P.wrows[ml] / 2, 128);  // whole 128-k tiles of k_leg_dir (zero padded)
      long long pant = (long long)roundup(std::max(nd, 1), 16) * P.ldk[ml];  // k_leg_dir reads stages of 16 latitudes in both precisions (LG_LS; the fp32 kernel's 32-latitude stages went in round 5)
      P.offTS[ml] = ptoff;
      P.offTA[ml] = ptoff + pant;
      ptoff += 2 * pant;
      P.lattile_pref[ml + 1] = P.lattile_pref[ml] + (nd + 63) / 64;
      {  // k_leg_dir's row tiles.  fp64: 2 per 128 n-pairs (one parity each), and for the rest none | one two-parity tile (<= 64 n-pairs) | two;
         // fp32: two-parity tiles of 64 n-pairs
        const int nk = P.wrows[ml] / 2, r = nk % 128;
        P.ktile_pref[ml + 1] = P.ktile_pref[ml] + (P.esz == 4 ? (nk + 63) / 64 : 2 * (nk / 128) + (r == 0 ? 0 : r <= 64 ? 1 : 2));
      }
      ebase[ml] = (int)eps.size();
      for (int n = m; n <= N + 2; n++)  // REPSNM (pre_suleg_mod.F90:55-63)
        eps.push_back(std::sqrt((double)(n * n - m * m) / (double)(4 * n * n - 1)));
    }
    P.nspec2 = ipos;
    P.p_elems = poff;
    P.pt_elems = ptoff;
    P.wrows_total = P.wbase[NU];
  }
  std::vector<int> rowm(P.wrows_total);
  for (int ml = 0; ml < NU; ml++)
    for (int r = 0; r < P.wrows[ml]; r++) rowm[P.wbase[ml] + r] = ml;
  P.l_nmen.assign(NL, 0);
  P.l_gpoff.assign(NL, 0);
  P.l_fbase.assign(NL + 1, 0);
  std::vector<double> l_rw(NL), l_racthe(NL);
  for (int jl = 0; jl < NL; jl++) {
    const int j = P.lat0 + jl;
    P.l_nmen[jl] = P.nmen[j];
    P.l_gpoff[jl] = (int)(cum[j] - cum[P.lat0]);
    P.l_fbase[jl + 1] = P.l_fbase[jl] + P.nmen[j] + 1;
    l_rw[jl] = P.rw[j];
    l_racthe[jl] = P.racthe[j];
  }
  P.frows = P.l_fbase[NL];
  P.lrows = 2LL * P.lbase[NU];

  // ---- Fourier-buffer row tables.  One task: both sides share one latitude-major buffer
  // (row = fbase[lat] + m).  Several tasks: the Legendre-side buffer is cut into one block per
  // destination task, the FFT-side buffer into one block per source task with exactly the same row
  // order, so the all-to-all-v moves whole blocks.  Inside a block the rows are latitude-major, then
  // wavenumber -- the order of the one-task buffer restricted to the block: an FFT workgroup then finds the
  // wavenumbers of its latitude in NPROC short contiguous runs.  (The other order, wavenumber-major, makes
  // every Fourier row of a latitude a separate far-apart line: measured on one task, EMI_FB_ORDER=m with
  // EMI_FB_TABLE=1, the FFT kernels are 13-20 % slower and the Legendre kernels 1 % faster; the row table
  // itself costs the FFT kernels 1.5 %.)
  std::vector<int> legN(std::max(P.lbase[NU], 1)), legS(std::max(P.lbase[NU], 1)), fftrow(P.frows);  // never empty: k_leg_dir clamps its look-ups to an entry that exists
  P.leg_rows.assign(NP, 0);
  P.leg_disp.assign(NP, 0);
  P.fft_rows.assign(NP, 0);
  P.fft_disp.assign(NP, 0);
  const bool tables = NP > 1 || (test_paths() & 1);  // one task: plain affine rows, no table
  if (!tables) {
    for (int ml = 0; ml < NU; ml++) {
      const int m = P.mval[ml], nd = P.lbase[ml + 1] - P.lbase[ml], isl0 = P.ndgnh - nd;
      for (int j = 0; j < nd; j++) {
        legN[P.lbase[ml] + j] = P.l_fbase[isl0 + j] + m;
        legS[P.lbase[ml] + j] = P.l_fbase[L - 1 - isl0 - j] + m;
      }
    }
    for (long long i = 0; i < P.frows; i++) fftrow[i] = (int)i;
    P.leg_rows[0] = P.fft_rows[0] = P.frows;
  } else {
    for (int ml = 0; ml < NU; ml++)
      if (ml > 0 && P.mval[ml] <= P.mval[ml - 1]) {
        delete pp;
        EMI_FAIL(EMI_ERR_RUNTIME, "internal: local wavenumbers not ascending");
      }
    // Legendre side: block d holds the rows (lat in band d, local wavenumber ml with m <= NMEN(lat))
    for (int ml = 0; ml < NU; ml++) {
      const int m = P.mval[ml];
      for (int lat = 0; lat < L; lat++)
        if (P.nmen[lat] >= m) P.leg_rows[band_of(lat)]++;
    }
    for (int d = 1; d < NP; d++) P.leg_disp[d] = P.leg_disp[d - 1] + P.leg_rows[d - 1];
    {
      std::vector<long long> pos(P.leg_disp);
      {
        // the local wavenumbers are ascending, so those present at a latitude are ml = 0 .. cnt-1 and the
        // row of (lat, ml) is the first row of the latitude + ml
        std::vector<long long> latbase(L);
        for (int lat = 0; lat < L; lat++) {
          const int cnt = (int)(std::upper_bound(P.mval.begin(), P.mval.end(), P.nmen[lat]) - P.mval.begin());
          latbase[lat] = pos[band_of(lat)];
          pos[band_of(lat)] += cnt;
        }
        for (int ml = 0; ml < NU; ml++) {
          const int nd = P.lbase[ml + 1] - P.lbase[ml], isl0 = P.ndgnh - nd;
          for (int j = 0; j < nd; j++) {
            legN[P.lbase[ml] + j] = (int)(latbase[isl0 + j] + ml);
            legS[P.lbase[ml] + j] = (int)(latbase[L - 1 - isl0 - j] + ml);
          }
        }
      }
    }
    // FFT side: block s holds the rows (local lat, wavenumber m of task s with m <= NMEN(lat)), same order
    for (int m = 0; m <= N; m++)
      for (int jl = 0; jl < NL; jl++)
        if (P.l_nmen[jl] >= m) P.fft_rows[P.procm[m]]++;
    for (int sr = 1; sr < NP; sr++) P.fft_disp[sr] = P.fft_disp[sr - 1] + P.fft_rows[sr - 1];
    {
      std::vector<long long> pos(P.fft_disp);
      for (int jl = 0; jl < NL; jl++)
        for (int m = 0; m <= P.l_nmen[jl]; m++) fftrow[P.l_fbase[jl] + m] = (int)pos[P.procm[m]]++;
    }
    long long tl = 0, tf = 0;
    for (int r = 0; r < NP; r++) tl += P.leg_rows[r], tf += P.fft_rows[r];
    if (tl != P.lrows || tf != P.frows) {
      delete pp;
      EMI_FAIL(EMI_ERR_RUNTIME, "internal: exchange tables inconsistent (%lld/%lld, %lld/%lld)", tl, P.lrows, tf, P.frows);
    }
  }
  std::vector<double> lapin(N + 4, 0.0);  // RLAPIN(-1:N+2) (pre_suleg_mod.F90:64-69)
  for (int n = 1; n <= N + 2; n++) lapin[n + 1] = -(P.ra * P.ra / (double)(n * (n + 1)));

  phase("distribution + index tables");
  // ---- Legendre panels of the local wavenumbers: PS[k][j] = P_{m+2k}^m(mu_{isl0+j}),
  // PA[k][j] = P_{m+2k+1}^m, zero padded; plus the [j][k] transposed copy for the direct transform
  void *dP = nullptr, *dPT = nullptr;
  const size_t esz = P.esz;
  if (emi_dev_malloc(&dP, (size_t)P.p_elems * esz) || emi_dev_malloc(&dPT, (size_t)P.pt_elems * esz)) {
    delete pp;
    EMI_FAIL(EMI_ERR_RUNTIME, "cannot allocate %.2f GiB for the Legendre panels", (P.p_elems + P.pt_elems) * (double)esz / (1 << 30));
  }
  P.d_P = (char *)dP;
  P.d_PT = (char *)dPT;
  P.dev_allocs.push_back(dP);
  P.dev_allocs.push_back(dPT);
  // m >= 2: k_legpol on the device (below, once the device tables exist); m = 0, 1 (ordinary Legendre
  // recurrence, supolf_mod.F90:124-142) on the host.  EMI_LEGPOL_HOST=1 computes every panel on the
  // host threads instead (the two paths agree bit for bit, tests/test_gpu_parity.py).
  // With CDIO_LEGPOL = readf / membuf every panel is taken from the file or segment instead (read_legpol_mod.F90:120-215).
  const bool belousov = cfg->lduserpnm != 0 && !lp_read;  // LDUSERPNM: SUPOL per latitude on the host threads
  const bool legpol_host = lp_read || belousov || (getenv("EMI_LEGPOL_HOST") && atoi(getenv("EMI_LEGPOL_HOST")));
  std::vector<std::vector<double>> belpan;  // [ml] the panels [par][k][j] filled latitude by latitude
  if (belousov) {
    const int nmaxb = N + 1;  // INSMAX = NTMAX + 1 (suleg_mod.F90:402-410)
    const emi::BelousovTables BT = emi::belousov_tables(nmaxb);
    belpan.resize(NU);
    for (int ml = 0; ml < NU; ml++) belpan[ml].assign((size_t)2 * (P.wrows[ml] / 2) * P.ldp[ml], 0.0);
    emi::parallel_for(P.ndgnh, [&](int j) {
      std::vector<double> pol((size_t)(nmaxb + 1) * (nmaxb + 1), 0.0);
      emi::belousov_latitude(BT, P.rmu[j], pol.data());
      for (int ml = 0; ml < NU; ml++) {
        const int m = P.mval[ml], nd = P.lbase[ml + 1] - P.lbase[ml], isl0 = P.ndgnh - nd;
        if (j < isl0) continue;
        const int ld = P.ldp[ml], nk = P.wrows[ml] / 2;
        double *pan = belpan[ml].data();
        for (int par = 0; par < 2; par++)
          for (int k = 0; m + 2 * k + par <= N + 1; k++)
            pan[((size_t)par * nk + k) * ld + (j - isl0)] = pol[(size_t)m * (nmaxb + 1) + (m + 2 * k + par)];
      }
    });
  }
  if (emi_dev_memset(dP, 0, (size_t)P.p_elems * esz, 0) || emi_dev_memset(dPT, 0, (size_t)P.pt_elems * esz, 0)) {
    delete pp;
    return EMI_ERR_RUNTIME;
  }
  emi_stream_sync(0);
  {
    std::atomic<int> bad{0};
    emi::parallel_for(NU, [&](int ml) {
      const int m = P.mval[ml];
      if (m >= 2 && !legpol_host) return;
      const int nd = P.lbase[ml + 1] - P.lbase[ml], isl0 = P.ndgnh - nd;
      const int ld = P.ldp[ml], nk = P.wrows[ml] / 2;
      const int nmax = N + 2;
      std::vector<double> pan((size_t)2 * nk * ld, 0.0), col(nmax + 1);
      std::vector<int> corr(nmax + 1);
      if (belousov) {
        pan.swap(belpan[ml]);
      } else if (lp_read) {
        // reference column c holds n descending: row k of the panel (n = m + 2k + par) is column nc-1-k
        for (int par = 0; par < 2; par++) {
          const int nc = par ? (N - m + 2) / 2 : (N - m + 3) / 2;
          const char *mat = lp_src.base + (par ? lp_src.offA[m] : lp_src.offS[m]);
          double *dst = pan.data() + (size_t)par * nk * ld;
          for (int k = 0; k < nc; k++) memcpy(dst + (size_t)k * ld, mat + ((size_t)(nc - 1 - k) * nd) * 8, (size_t)nd * 8);
        }
      } else {
        emi::LegCoef lc = emi::legendre_coefficients(m, nmax);
        for (int j = 0; j < nd; j++) {
          double mu = P.rmu[isl0 + j];
          for (int par = 0; par < 2; par++) {
            emi::legendre_column(lc, mu, par, col.data(), corr.data());
            double *dst = pan.data() + (size_t)par * nk * ld;
            for (int k = 0; m + 2 * k + par <= N + 1; k++) dst[(size_t)k * ld + j] = col[m + 2 * k + par];
          }
        }
      }
      // the recurrences always run in double (as the reference's _sp build does: the JPRD work arrays of suleg_mod.F90:130-162);
      // the fp32 library rounds the finished panel once
      std::vector<float> cvt;
      auto put = [&](char *dst, const std::vector<double> &v) {
        if (esz == 8) return emi_h2d(dst, v.data(), v.size() * 8, 0);
        cvt.resize(v.size());
        for (size_t i = 0; i < v.size(); i++) cvt[i] = (float)v[i];
        int rc = emi_h2d(dst, cvt.data(), cvt.size() * 4, 0);
        emi_stream_sync(0);  // cvt is reused
        return rc;
      };
      if (put(P.d_P + P.offS[ml] * esz, pan)) bad = 1;
      const int ldk = P.ldk[ml], ndp = roundup(std::max(nd, 1), 16);  // = the panel extent behind offTA (pant)
      std::vector<double> pt((size_t)2 * ndp * ldk, 0.0);
      for (int par = 0; par < 2; par++)
        for (int k = 0; k < nk; k++)
          for (int j = 0; j < nd; j++) pt[((size_t)par * ndp + j) * ldk + k] = pan[((size_t)par * nk + k) * ld + j];
      if (put(P.d_PT + P.offTS[ml] * esz, pt)) bad = 1;
      emi_stream_sync(0);
    });
    if (bad) {
      delete pp;
      return EMI_ERR_RUNTIME;
    }
  }
  phase("legendre panels (host part)");
  // ---- device tables
  int *d_mval, *d_nmen, *d_gpoff, *d_nasm0, *d_fbase, *d_fftrow, *d_lbase, *d_legN, *d_legS, *d_wbase, *d_wrows, *d_rowm, *d_ebase,
      *d_ldp, *d_ldk, *d_ltp, *d_ktp;
  double *d_eps, *d_lapin, *d_rw, *d_racthe, *d_specw;
  long long *d_offS, *d_offA, *d_offTS, *d_offTA;
  if (upload(P.mval, &d_mval) || upload(P.l_nmen, &d_nmen) || upload(P.l_gpoff, &d_gpoff) || upload(P.nasm0, &d_nasm0) ||
      upload(P.l_fbase, &d_fbase) || upload(fftrow, &d_fftrow) || upload(P.lbase, &d_lbase) || upload(legN, &d_legN) ||
      upload(legS, &d_legS) || upload(P.wbase, &d_wbase) || upload(P.wrows, &d_wrows) || upload(rowm, &d_rowm) ||
      upload(ebase, &d_ebase) || upload(P.ldp, &d_ldp) || upload(P.ldk, &d_ldk) || upload(P.lattile_pref, &d_ltp) ||
      upload(P.ktile_pref, &d_ktp) || upload(eps, &d_eps) || upload(lapin, &d_lapin) || upload(l_rw, &d_rw) ||
      upload(l_racthe, &d_racthe) || upload(specw, &d_specw) || upload(P.offS, &d_offS) || upload(P.offA, &d_offA) ||
      upload(P.offTS, &d_offTS) || upload(P.offTA, &d_offTA)) {
    delete pp;
    return EMI_ERR_RUNTIME;
  }
  for (void *p : {(void *)d_mval, (void *)d_nmen, (void *)d_gpoff, (void *)d_nasm0, (void *)d_fbase, (void *)d_fftrow, (void *)d_lbase,
                  (void *)d_legN, (void *)d_legS, (void *)d_wbase, (void *)d_wrows, (void *)d_rowm, (void *)d_ebase, (void *)d_ldp,
                  (void *)d_ldk, (void *)d_ltp, (void *)d_ktp, (void *)d_eps, (void *)d_lapin, (void *)d_rw, (void *)d_racthe,
                  (void *)d_specw, (void *)d_offS, (void *)d_offA, (void *)d_offTS, (void *)d_offTA})
    P.dev_allocs.push_back(p);
  EmiGeomDev &g = P.g;
  g.nsmax = N;
  g.nump = NU;
  g.nlat = NL;
  g.ngptot = P.ngptot;
  g.m0_wide = P.esz == 4 ? 1 : 0;  // the fp32 library computes zonal wavenumber 0 in double (ledir_mod.F90:133-171)
  g.mval = d_mval;
  g.nmen = d_nmen;
  g.gpoff = d_gpoff;
  g.nasm0 = d_nasm0;
  g.fbase = d_fbase;
  g.fftrow = tables ? d_fftrow : nullptr;  // one task: rows are fbase[lat] + m, no table
  g.lbase = d_lbase;
  g.legN = d_legN;
  g.legS = d_legS;
  g.wbase = d_wbase;
  g.wrows = d_wrows;
  g.rowm = d_rowm;
  g.ebase = d_ebase;
  g.eps = d_eps;
  g.lapin = d_lapin;
  g.rw = d_rw;
  g.racthe = d_racthe;
  g.P = P.d_P;
  g.offS = d_offS;
  g.offA = d_offA;
  g.ldp = d_ldp;
  g.PT = P.d_PT;
  g.offTS = d_offTS;
  g.offTA = d_offTA;
  g.ldk = d_ldk;
  g.lattile_pref = d_ltp;
  g.ktile_pref = d_ktp;
  g.specw = d_specw;
  phase("device tables");
  if (!legpol_host) {
    // ---- Legendre panels, m >= 2: one thread per (wavenumber, parity, latitude)
    const int nmax = N + 2;
    std::vector<double> dcl((size_t)NU * (nmax + 1), 0.0), ddl((size_t)NU * (nmax + 1), 0.0), zf(NU, 0.0), mu(P.rmu.begin(), P.rmu.begin() + P.ndgnh);
    std::vector<int> blk;
    emi::parallel_for(NU, [&](int ml) {
      if (P.mval[ml] < 2) return;
      emi::LegCoef lc = emi::legendre_coefficients(P.mval[ml], nmax);
      std::copy(lc.dcl.begin(), lc.dcl.end(), dcl.begin() + (size_t)ml * (nmax + 1));
      std::copy(lc.ddl.begin(), lc.ddl.end(), ddl.begin() + (size_t)ml * (nmax + 1));
      zf[ml] = lc.zfac_m;
    });
    for (int ml = 0; ml < NU; ml++) {  // m ascending = longest recurrences first
      if (P.mval[ml] < 2) continue;
      const int nd = P.lbase[ml + 1] - P.lbase[ml];
      for (int par = 0; par < 2; par++)
        for (int jt = 0; jt * 64 < nd; jt++) {
          blk.push_back(ml);
          blk.push_back(par << 16 | jt);
        }
    }
    if (!blk.empty()) {
      double *d_dcl, *d_ddl, *d_zf, *d_mu;
      int *d_blk;
      if (upload(dcl, &d_dcl) || upload(ddl, &d_ddl) || upload(zf, &d_zf) || upload(mu, &d_mu) || upload(blk, &d_blk)) {
        delete pp;
        return EMI_ERR_RUNTIME;
      }
      LegPolDev la{d_mu, d_dcl, d_ddl, d_zf, d_blk, P.ndgnh, nmax};
      EmiRange rg_suleg(EMI_LBL_SULEG);  // GSTATS 140
      EMI_LAUNCH_P(P.esz, k_legpol, blk.size() / 2, 64, 0, (emi_stream_t)0, P.g, la);
      emi_stream_sync(0);
      for (void *q : {(void *)d_dcl, (void *)d_ddl, (void *)d_zf, (void *)d_mu, (void *)d_blk}) emi_dev_free(q);
    }
    phase("legendre panels (device)");
  }
  int rc = build_fft_plans(P);
  phase("fft plans + tables");
  if (rc) {
    delete pp;
    return rc;
  }
  emi_stream_sync(0);
  P.active = true;
  if (G.plans[slot]) delete G.plans[slot];
  G.plans[slot] = pp;
  if (kresol) *kresol = slot + 1;
  if (lp_mode == LP_WRITEF) {  // suleg_mod.F90:1186
    rc = legpol_write(slot + 1, io->fname);
    phase("legendre file written");
    if (rc) {
      emi_release(slot + 1);
      return rc;
    }
  }
  return EMI_SUCCESS;
}

extern "C" int emi_setup(const emi_setup_t *cfg, int *kresol) { return emi_setup_legpol(cfg, nullptr, kresol); }

extern "C" int emi_set_alltoallv(emi_alltoallv_fn fn, void *user) {
  G.a2a = fn;
  G.a2a_user = user;
  return EMI_SUCCESS;
}

extern "C" int emi_release(int kresol) {
  Plan *P = get_plan(kresol);
  if (!P) EMI_FAIL(EMI_ERR_STATE, "TRANS_RELEASE: unknown resolution %d", kresol);
  plan_quiesce(*P);
  emi_stream_sync(0);
#ifndef EMI_CPU_EMU
  if (P->ev_done) (void)hipEventDestroy(P->ev_done);
#endif
  for (void *p : P->dev_allocs) emi_dev_free(p);
  for (auto &kv : P->legmaps) {
    emi_dev_free(kv.second.d_inv);
    emi_dev_free(kv.second.d_inv_wide);
    emi_dev_free(kv.second.d_dir_wide);
    emi_dev_free(kv.second.d_dir);
  }
  emi_dev_free(P->d_W);
  emi_dev_free(P->d_FBL);
  if (P->d_FBF != P->d_FBL) emi_dev_free(P->d_FBF);
  emi_dev_free(P->d_desc);
  emi_dev_free(P->d_fftscr);
  P->active = false;
  delete P;
  G.plans[kresol - 1] = nullptr;
  emi_stage::trim();  // the idle staging buffers of host-array calls were sized for this resolution
  return EMI_SUCCESS;
}

// Frees the idle device staging buffers kept between host-array calls (several GiB after a large call; other allocators of
// the process -- torch, the host model -- cannot see or reclaim them).  Waits for the work queued on the null stream first:
// a buffer is idle once its call has returned, but that call's last copies may still be in flight.
extern "C" int emi_trim_cache(void) {
  emi_stream_sync(0);
  emi_stage::trim();
  return EMI_SUCCESS;
}

extern "C" int emi_finalize(void) {
  if (!G.init) return EMI_SUCCESS;
  for (int i = 0; i < (int)G.plans.size(); i++)
    if (G.plans[i] && G.plans[i]->active) emi_release(i + 1);
  G.plans.clear();
  emi_stage::trim();
  G.init = false;
  return EMI_SUCCESS;
}

// ------------------------------------------------------------------------------------------
// TRANS_INQ
// ------------------------------------------------------------------------------------------
extern "C" int emi_inq_int(int kresol, const char *name, int *value) {
  Plan *P = get_plan(kresol);
  if (!P) EMI_FAIL(EMI_ERR_STATE, "TRANS_INQ: unknown resolution %d", kresol);
  std::string s(name ? name : "");
  if (s == "nspec2")
    *value = P->nspec2;
  else if (s == "nspec2g")
    *value = P->nspec2g;
  else if (s == "nspec2mx") {
    int mx = 0;
    std::vector<int> cnt(P->nproc, 0);
    for (int m = 0; m <= P->nsmax; m++) cnt[P->procm[m]] += 2 * (P->nsmax - m + 1);
    for (int c : cnt) mx = std::max(mx, c);
    *value = mx;
  } else if (s == "nspec")
    *value = P->nspec2 / 2;
  else if (s == "nspecg")
    *value = P->nspec2g / 2;
  else if (s == "ngptot")  // this task's grid points: its sub-band with V-sets, else its band
    *value = P->nprv > 1 ? (int)P->vpoints(P->me, P->mev) : P->ngptot;
  else if (s == "ngptot_band")  // grid points of the whole band of this task's W-set (the FFT work of its V-set)
    *value = P->ngptot;
  else if (s == "ngptotg")
    *value = P->ngptotg;
  else if (s == "ngptotmx") {
    long long mx = 0;
    for (int r = 0; r < P->nproc; r++)
      for (int v = 0; v < P->nprv; v++) mx = std::max(mx, P->vpoints(r, v));
    *value = (int)mx;
  } else if (s == "nump")
    *value = P->nump;
  else if (s == "ndgl")
    *value = P->ndgl;
  else if (s == "nsmax")
    *value = P->nsmax;
  else if (s == "ndlon")
    *value = *std::max_element(P->nloen.begin(), P->nloen.end());
  else if (s == "nproc")
    *value = P->nproc * P->nprv;
  else if (s == "nprtrw")
    *value = P->nproc;
  else if (s == "nprtrv")
    *value = P->nprv;
  else if (s == "myproc")
    *value = P->me * P->nprv + P->mev + 1;
  else if (s == "mysetw")
    *value = P->me + 1;
  else if (s == "mysetv")
    *value = P->mev + 1;
  else if (s == "nfrstlat")  // first (1-based, global) latitude of this task's grid share (its band; its sub-band with V-sets)
    *value = P->vfirst(P->me, P->mev) + 1;
  else if (s == "nlstlat")
    *value = P->vlast(P->me, P->mev);
  else
    EMI_FAIL(EMI_ERR_ARG, "emi_inq_int: unknown name '%s'", s.c_str());
  return EMI_SUCCESS;
}

extern "C" int emi_inq_int_array(int kresol, const char *name, int *out, int len) {
  Plan *P = get_plan(kresol);
  if (!P) EMI_FAIL(EMI_ERR_STATE, "TRANS_INQ: unknown resolution %d", kresol);
  std::string s(name ? name : "");
  const std::vector<int> *v = nullptr;
  std::vector<int> tmp;
  if (s == "nloen")
    v = &P->nloen;
  else if (s == "nmen" || s == "nmeng")
    v = &P->nmen;
  else if (s == "ndglu")
    v = &P->ndglu;
  else if (s == "nasm0") {  // D%NASM0(0:NSMAX): 1-based address of (m, n=m), -99 for foreign m
    tmp.assign(P->nsmax + 1, -99);
    for (int ml = 0; ml < P->nump; ml++) tmp[P->mval[ml]] = P->nasm0[ml] + 1;
    v = &tmp;
  } else if (s == "myms")
    v = &P->mval;
  else if (s == "procm") {  // D%NPROCM: owning W-set (1-based) of every wavenumber
    tmp = P->procm;
    for (auto &x : tmp) x += 1;
    v = &tmp;
  } else if (s == "latlo") {  // [nproc+1] 0-based first latitude of every task's grid share, task order
    if (P->nprv > 1)
      v = &P->vlat;
    else
      v = &P->latlo;
  } else if (s == "fftwork") {  // [ndgl] work length of the FFT of every latitude of this task (0 elsewhere): NLOEN/2 or NLOEN,
    // or the Bluestein length; diagnostic (tests check which specialised kernel a row length selects)
    tmp.assign(P->ndgl, 0);
    for (int j = 0; j < P->nlat; j++) tmp[P->lat0 + j] = P->fplans[P->planid[j]].S;
    v = &tmp;
  } else
    EMI_FAIL(EMI_ERR_ARG, "emi_inq_int_array: unknown name '%s'", s.c_str());
  if (len < (int)v->size()) EMI_FAIL(EMI_ERR_ARG, "TRANS_INQ: %s TOO SMALL (%d < %zu)", s.c_str(), len, v->size());
  std::copy(v->begin(), v->end(), out);
  return EMI_SUCCESS;
}

extern "C" int emi_inq_real_array(int kresol, const char *name, double *out, int len) {
  Plan *P = get_plan(kresol);
  if (!P) EMI_FAIL(EMI_ERR_STATE, "TRANS_INQ: unknown resolution %d", kresol);
  std::string s(name ? name : "");
  const std::vector<double> *v = nullptr;
  if (s == "rmu" || s == "pmu")
    v = &P->rmu;
  else if (s == "rgw" || s == "pgw" || s == "rw")
    v = &P->rw;
  else if (s == "racthe")
    v = &P->racthe;
  else if (s == "rlapin" || s == "plapin") {
    // RLAPIN(-1:NSMAX+2) (pre_suleg_mod.F90:64-69): eigenvalues of the inverse Laplacian, -a^2/(n(n+1)); 0 for n <= 0
    if (len < P->nsmax + 4) EMI_FAIL(EMI_ERR_ARG, "TRANS_INQ: PLAPIN TOO SMALL");
    out[0] = out[1] = 0.0;
    for (int n = 1; n <= P->nsmax + 2; n++) out[n + 1] = -(P->ra * P->ra / (double)(n * (n + 1)));
    return EMI_SUCCESS;
  } else
    EMI_FAIL(EMI_ERR_ARG, "emi_inq_real_array: unknown name '%s'", s.c_str());
  if (len < (int)v->size()) EMI_FAIL(EMI_ERR_ARG, "TRANS_INQ: %s TOO SMALL", s.c_str());
  std::copy(v->begin(), v->end(), out);
  return EMI_SUCCESS;
}

extern "C" int emi_inq_legendre(int kresol, int m, int symmetric, double *out, int *nrows, int *ncols) {
  Plan *P = get_plan(kresol);
  if (!P) EMI_FAIL(EMI_ERR_STATE, "TRANS_INQ: unknown resolution %d", kresol);
  if (m < 0 || m > P->nsmax) EMI_FAIL(EMI_ERR_ARG, "emi_inq_legendre: m out of range");
  if (P->procm[m] != P->me) EMI_FAIL(EMI_ERR_ARG, "emi_inq_legendre: wavenumber %d belongs to task %d", m, P->procm[m] + 1);
  const int ml = (int)(std::lower_bound(P->mval.begin(), P->mval.end(), m) - P->mval.begin());
  const int N = P->nsmax, nd = std::min(P->ndgnh, P->ndglu[m]);
  const int nc = symmetric ? (N - m + 3) / 2 : (N - m + 2) / 2;
  if (nrows) *nrows = nd;
  if (ncols) *ncols = nc;
  if (!out) return EMI_SUCCESS;
  const int ld = P->ldp[ml], nk = P->wrows[ml] / 2;
  std::vector<double> pan((size_t)nk * ld);
  const char *src = P->d_P + (symmetric ? P->offS[ml] : P->offA[ml]) * P->esz;
  if (P->esz == 8) {
    if (emi_d2h(pan.data(), src, pan.size() * 8, 0)) return EMI_ERR_RUNTIME;
    emi_stream_sync(0);
  } else {
    std::vector<float> pf(pan.size());
    if (emi_d2h(pf.data(), src, pf.size() * 4, 0)) return EMI_ERR_RUNTIME;
    emi_stream_sync(0);
    for (size_t i = 0; i < pf.size(); i++) pan[i] = pf[i];
  }
  // reference column c (0-based) holds n descending: k = nc-1-c
  for (int c = 0; c < nc; c++)
    for (int j = 0; j < nd; j++) out[(size_t)c * nd + j] = pan[(size_t)(nc - 1 - c) * ld + j];
  return EMI_SUCCESS;
}

// WRITE_LEGPOL (write_legpol_mod.F90:66-158): the panels as emi_inq_legendre returns them, in MYMS order
static int legpol_write(int kresol, const char *fname) {
  Plan *P = get_plan(kresol);
  if (!P) EMI_FAIL(EMI_ERR_STATE, "WRITE_LEGPOL: unknown resolution %d", kresol);
  FILE *f = fopen(fname, "wb");
  if (!f) EMI_FAIL(EMI_ERR_ARG, "WRITE_LEGPOL: BYTES_IO_OPEN FAILED (%s)", fname);
  int head[4];
  memcpy(head, "LEGPOL  ", 8);
  head[2] = P->nsmax;
  head[3] = P->ndgnh;
  bool ok = fwrite(head, 4, 4, f) == 4;
  std::vector<int> lat(2 * (size_t)P->ndgnh);
  for (int j = 0; j < P->ndgnh; j++) {
    lat[2 * j] = P->nloen[j];
    lat[2 * j + 1] = P->nmen[j];
  }
  ok = ok && fwrite(lat.data(), 4, lat.size(), f) == lat.size();
  std::vector<double> buf;
  for (int m = 0; ok && m <= P->nsmax; m++)
    for (int sym = 0; ok && sym < 2; sym++) {  // anti-symmetric first
      int nr = 0, nc = 0;
      if (emi_inq_legendre(kresol, m, sym, nullptr, &nr, &nc)) {
        fclose(f);
        return EMI_ERR_RUNTIME;
      }
      buf.resize((size_t)nr * nc);
      if (emi_inq_legendre(kresol, m, sym, buf.data(), &nr, &nc)) {
        fclose(f);
        return EMI_ERR_RUNTIME;
      }
      ok = fwrite(buf.data(), 8, buf.size(), f) == buf.size();
    }
  ok = (fclose(f) == 0) && ok;
  if (!ok) EMI_FAIL(EMI_ERR_RUNTIME, "WRITE_LEGPOL:BYTES_IO_WRITE FAILED (%s)", fname);
  return EMI_SUCCESS;
}

// ------------------------------------------------------------------------------------------
// transforms
// ------------------------------------------------------------------------------------------
struct HostStage {  // staging of host arrays through device memory (mem_space == HOST): emi_stage.h
  size_t esz;
  explicit HostStage(int e) : esz((size_t)e) {}
  std::vector<void *> dev;
  bool failed = false;  // a staging buffer could not be allocated or filled: the call must not launch anything
  std::vector<std::pair<void *, std::pair<void *, size_t>>> outs;  // dev -> (host, bytes)
  ~HostStage() {
    for (void *p : dev) emi_stage::release(p);
  }
  const void *in(const void *h, size_t elems, bool host, emi_stream_t s) {
    if (!h || !host) return h;
    void *d = emi_stage::acquire(elems * esz);
    if (!d) {
      failed = true;
      return nullptr;
    }
    dev.push_back(d);
    if (emi_h2d(d, h, elems * esz, s)) failed = true;
    return d;
  }
  // preload: the call does not write every element of the array (padding of the last NPROMA block, more
  // fields in the array than the call produces) -- start from the caller's contents so that the copy back
  // leaves those elements as they were, as the reference does
  void *out(void *h, size_t elems, bool host, bool preload = false, emi_stream_t s = 0) {
    if (!h || !host) return h;
    void *d = emi_stage::acquire(elems * esz);
    if (!d) {
      failed = true;
      return nullptr;
    }
    dev.push_back(d);
    if (preload && emi_h2d(d, h, elems * esz, s)) failed = true;
    outs.push_back({d, {h, elems * esz}});
    return d;
  }
  void flush(emi_stream_t s) {
    for (auto &o : outs) emi_d2h(o.second.first, o.first, o.second.second, s);
    emi_stream_sync(s);
  }
};

static int grow(Plan &P, char **p, size_t *cap, size_t need, const char *what, emi_stream_t st) {
  if (need <= *cap) return 0;
  plan_quiesce(P);  // kernels of the previous call (on whatever stream it named) may still use the old buffer
  emi_stream_sync(0);
  emi_dev_free(*p);
  *p = nullptr;
  *cap = 0;
  void *q;
  if (emi_dev_malloc(&q, need)) EMI_FAIL(EMI_ERR_RUNTIME, "cannot allocate %.2f GiB %s", need / 1073741824.0, what);
  *p = (char *)q;
  *cap = need;
  emi_dev_memset(q, 0, need, st);  // on the caller's stream: ordered ahead of this call's kernels
  return 0;
}

static int ensure_work(Plan &P, int bfpad, int nfb, emi_stream_t st) {
  const size_t rowb = (size_t)2 * bfpad * P.esz;
  if (grow(P, &P.d_W, &P.cap_W, (size_t)P.wrows_total * rowb, "packed-spectral work buffer", st)) return -1;
  if (P.nproc == 1) {
    // + 1 per buffer: a row of zeros behind the Fourier rows (k_leg_dir reads it for latitudes past the last one of a stage)
    if (grow(P, &P.d_FBL, &P.cap_FBL, (size_t)nfb * (P.frows + 1) * rowb, "Fourier work buffer", st)) return -1;
    P.d_FBF = P.d_FBL;
    P.cap_FBF = P.cap_FBL;
  } else {
    if (grow(P, &P.d_FBL, &P.cap_FBL, (size_t)nfb * (P.lrows + 1) * rowb, "Fourier (Legendre-side) exchange buffer", st)) return -1;
    if (grow(P, &P.d_FBF, &P.cap_FBF, (size_t)nfb * P.frows * rowb, "Fourier (FFT-side) exchange buffer", st)) return -1;
  }
  return 0;
}

// Per-phase device timing with HIP event pairs recorded on the stream each kernel runs on.
// Nothing is synchronised inside a transform call; emi_last_phase_ms() resolves the events lazily.
struct PhaseTimer {
  static const int MAXIV = 4096;
  int n = 0, ncreated = 0;
  bool on = false;
  int kinds[MAXIV];        // 0 spectral pack / unpack, 1 Legendre, 2 FFT, 3 exchange (the all-to-all-v hook, on the stream it is queued on)
  double xbytes = 0;       // bytes this task handed to the exchange hook since the measurement started (send side, peers only)
  long long fft_kernels = 0;  // kernel launches of the FFT phases since the measurement started (launch_fft)
#ifndef EMI_CPU_EMU
  hipEvent_t e0[MAXIV], e1[MAXIV];
  // keep: the intervals of earlier calls stay (accumulating mode, emi_set_profile(2)): nothing has to be resolved --
  // i.e. no host synchronisation -- between the calls of a timed loop
  void begin(bool on_, bool keep = false) {
    on = on_;
    if (!keep) n = 0, xbytes = 0, fft_kernels = 0;
  }
  int start(int kind, emi_stream_t s) {
    if (!on || n >= MAXIV) return -1;
    if (n >= ncreated) {  // events are created as the intervals are first used
      (void)hipEventCreate(&e0[n]);
      (void)hipEventCreate(&e1[n]);
      ncreated = n + 1;
    }
    kinds[n] = kind;
    (void)hipEventRecord(e0[n], s);
    return n++;
  }
  void stop(int iv, emi_stream_t s) {
    if (iv >= 0) (void)hipEventRecord(e1[iv], s);
  }
  void resolve(double *ms3, int *launches) {  // four entries each
    for (int i = 0; i < 4; i++) ms3[i] = 0, launches[i] = 0;
    if (!on) return;
    for (int i = 0; i < n; i++) {
      (void)hipEventSynchronize(e1[i]);
      float ms = 0;
      (void)hipEventElapsedTime(&ms, e0[i], e1[i]);
      ms3[kinds[i]] += ms;
      launches[kinds[i]]++;
    }
  }
#else
  void begin(bool, bool = false) {}
  int start(int, emi_stream_t) { return -1; }
  void stop(int, emi_stream_t) {}
  void resolve(double *ms3, int *launches) {
    for (int i = 0; i < 4; i++) ms3[i] = 0, launches[i] = 0;
  }
#endif
};
static PhaseTimer g_pt;

// TRLTOM / TRMTOL (trltom_mod.F90:96-136, trmtol_mod.F90:101-141): one all-to-all-v of whole
// row blocks of the Fourier buffers.  to_fft: Legendre-side -> FFT-side (inverse transform).
// The hook always sees ALL tasks of the job (one entry per task, zero bytes for tasks that are not peers of this exchange): with
// V-sets TRMTOL / TRLTOM run among the NPRTRW tasks that share this task's V-set (task w * NPRTRV + mysetv), TRLTOG / TRGTOL
// among the NPRTRV tasks of its W-set -- no sub-communicators, any transport that handles empty blocks works unchanged.
// Displacements of the empty blocks continue the dense order (the Python transport checks it).
static int hook_alltoallv(const void *sb, const std::vector<long long> &psc, void *rb, const std::vector<long long> &prc, int npeers, int stride, int first,
                          emi_stream_t st) {
  const int NA = G.nproc_all;
  std::vector<long long> sc(NA, 0), sd(NA, 0), rc(NA, 0), rd(NA, 0);
  for (int k = 0; k < npeers; k++) sc[first + k * stride] = psc[k], rc[first + k * stride] = prc[k];
  for (int r = 1; r < NA; r++) sd[r] = sd[r - 1] + sc[r - 1], rd[r] = rd[r - 1] + rc[r - 1];
  const int iv = g_pt.start(3, st);
  for (int r = 0; r < NA; r++)
    if (g_pt.on && r != G.myproc_all - 1) g_pt.xbytes += (double)sc[r];
  const int rc_hook = G.a2a(G.a2a_user, sb, sc.data(), sd.data(), rb, rc.data(), rd.data(), NA, (void *)st);
  g_pt.stop(iv, st);
  if (rc_hook != 0) EMI_FAIL(EMI_ERR_RUNTIME, "all-to-all-v hook failed");
  return 0;
}
static int exchange(Plan &P, bool to_fft, int ldf, emi_stream_t st, char *FBl, char *FBf) {
  if (P.nproc == 1) return 0;
  const int NP = P.nproc;
  std::vector<long long> sc(NP), rc(NP);
  const long long rowb = (long long)ldf * P.esz;
  for (int r = 0; r < NP; r++) {
    // the blocks of both buffers are dense and in task order (leg_disp / fft_disp are the running sums of the rows)
    const long long lr = P.leg_rows[r] * rowb, fr = P.fft_rows[r] * rowb;
    sc[r] = to_fft ? lr : fr;
    rc[r] = to_fft ? fr : lr;
  }
  const void *sb = to_fft ? FBl : FBf;
  void *rb = to_fft ? FBf : FBl;
  return hook_alltoallv(sb, sc, rb, rc, NP, P.nprv, P.mev, st);
}

static int ensure_desc(Plan &P, size_t bytes);
// The field descriptors of a call go to the device through a small ring of pinned host buffers, so that the copy is
// asynchronous and the call never waits for the stream: the host prepares and queues the next call while the
// kernels of this one run (0.6 ms of host work per pair, otherwise exposed between any two calls).  A ring slot is
// reused four calls later, after the event behind its copy.  The device buffer is one per resolution: the copy of
// the next call is ordered behind this call's kernels on the caller's stream.
static int upload_desc(Plan &P, const std::vector<char> &hd, emi_stream_t st) {
  if (ensure_desc(P, hd.size())) return EMI_ERR_RUNTIME;
#ifdef EMI_CPU_EMU
  emi_h2d(P.d_desc, hd.data(), hd.size(), st);
  return 0;
#else
  static struct {
    void *h[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t cap[4] = {0, 0, 0, 0};
    hipEvent_t ev[4];
    bool init[4] = {false, false, false, false};
    unsigned next = 0;
  } ring;
  const int k = (int)(ring.next++ & 3u);
  if (ring.init[k]) {
    EMI_CHECK(hipEventSynchronize(ring.ev[k]));
  } else {
    EMI_CHECK(hipEventCreateWithFlags(&ring.ev[k], hipEventDisableTiming));
    ring.init[k] = true;
  }
  if (ring.cap[k] < hd.size()) {
    if (ring.h[k]) (void)hipHostFree(ring.h[k]);
    ring.h[k] = nullptr;
    ring.cap[k] = 0;
    const size_t cap = std::max(hd.size(), (size_t)1 << 16);
    EMI_CHECK(hipHostMalloc(&ring.h[k], cap, hipHostMallocDefault));
    ring.cap[k] = cap;
  }
  memcpy(ring.h[k], hd.data(), hd.size());
  EMI_CHECK(hipMemcpyAsync(P.d_desc, ring.h[k], hd.size(), hipMemcpyHostToDevice, (hipStream_t)st));
  EMI_CHECK(hipEventRecord(ring.ev[k], (hipStream_t)st));
  return 0;
#endif
}

static int ensure_desc(Plan &P, size_t bytes) {
  if (bytes <= P.cap_desc) return 0;
  plan_quiesce(P);
  emi_stream_sync(0);
  emi_dev_free(P.d_desc);
  P.d_desc = nullptr;
  size_t cap = std::max(bytes, (size_t)1 << 16);
  if (emi_dev_malloc(&P.d_desc, cap)) return EMI_ERR_RUNTIME;
  P.cap_desc = cap;
  return 0;
}

// All eight XCDs share every wavenumber: an XCD takes a range of column tiles (and a residue class of row tiles when there are fewer
// column tiles than XCDs), column tiles innermost, so that consecutive tiles of an XCD share the panel rows.  (Measured and rejected,
// wavenumber groups of 4 / 2 / 1 XCDs -- fetched bytes go up, time never down; row tiles innermost -- 10 % fewer
// bytes for k_leg_dir and 1.4 % more time.)
// which: 0 every wavenumber, 1 all but the wide one (fp32 library: m = 0 accumulates in double and has a kernel of its own), 2 only that one
static int build_tilemap(const Plan &P, const std::vector<int> &pref, int nct, int2 **d_map, long long *nblocks, int which = 0) {
  const int nx = 8;  // XCDs
  int gx = 1;
  while (gx * 2 <= std::min(nct, nx)) gx *= 2;
  const int gy = nx / gx;
  std::vector<std::vector<int2>> per(8);
  for (int ml = 0; ml < P.nump; ml++) {
    const bool wide = P.esz == 4 && P.mval[ml] == 0;
    if ((which == 1 && wide) || (which == 2 && !wide)) continue;
    const int nrt = pref[ml + 1] - pref[ml];
    for (int x = 0; x < nx; x++) {
      // rotate the column ranges and row residues with the wavenumber: the ranges differ by one tile, rotation
      // evens the per-XCD totals out (a fixed assignment leaves XCDs with 4 of 26 column tiles 23 %
      // more work than those with 3)
      const int xc = (x + ml) % gx, xl = (x / gx + ml) % gy;
      const int c0 = (int)((long long)xc * nct / gx), c1 = (int)((long long)(xc + 1) * nct / gx);
      for (int rt = xl; rt < nrt; rt += gy)
        for (int ct = c0; ct < c1; ct++) per[x].push_back(int2{ml, (rt << 16) | ct});
    }
  }
  size_t mx = 0;
  for (auto &v : per) mx = std::max(mx, v.size());
  std::vector<int2> map(mx * 8, int2{-1, 0});
  for (int x = 0; x < 8; x++)
    for (size_t k = 0; k < per[x].size(); k++) map[k * 8 + x] = per[x][k];
  *nblocks = (long long)map.size();
  if (map.empty()) {  // (which == 2 on a task that does not own m = 0)
    *d_map = nullptr;
    return 0;
  }
  return upload(map, d_map);
}

static int leg_tilemaps(Plan &P, int nct, LegMaps **out) {
  auto it = P.legmaps.find(nct);
  if (it == P.legmaps.end()) {
    LegMaps lm;
    const bool split = P.esz == 4;  // fp32 library: k_leg_inv / k_leg_dir without the double-precision tiles, which go to k_leg_*_wide
    if (build_tilemap(P, P.lattile_pref, nct, &lm.d_inv, &lm.n_inv, split ? 1 : 0) || build_tilemap(P, P.ktile_pref, nct, &lm.d_dir, &lm.n_dir, split ? 1 : 0) ||
        (split && (build_tilemap(P, P.lattile_pref, nct, &lm.d_inv_wide, &lm.n_inv_wide, 2) || build_tilemap(P, P.ktile_pref, nct, &lm.d_dir_wide, &lm.n_dir_wide, 2))))
      return EMI_ERR_RUNTIME;
    it = P.legmaps.emplace(nct, lm).first;
  }
  *out = &it->second;
  return 0;
}

static int pick_batch(Plan &P, int nfields, int depth) {
  // fields per batch: bounded by free HBM (W + FB rows x 16 B per field; FB twice when batches are
  // pipelined) and EMI_MAX_BATCH; multiples of 64 fields so the 128-column tiles are full
  size_t fr = 0, tot = 0;
  emi_mem_info(&fr, &tot);
  size_t have = fr + P.cap_W + P.cap_FBL + (P.nproc > 1 ? P.cap_FBF : 0) + emi_stage::idle_bytes();
  const int nfb = depth > 1 ? 2 : 1;
  double per_field = (double)(P.wrows_total + nfb * P.frows + (P.nproc > 1 ? nfb * P.lrows : 0)) * 2.0 * P.esz;
  long long cap = (long long)((double)have * 0.85 / per_field);
  cap = cap / 64 * 64;
  if (cap < 64) cap = 64;
  if (G.max_batch > 0) cap = std::min<long long>(cap, std::max(64, roundup(G.max_batch, 64)));
  long long want = roundup((nfields + depth - 1) / depth, 64);
  return (int)std::min(cap, std::max(64LL, want));
}


static int launch_fft(Plan &P, bool inverse, bool adj, const GridFld *d_flds, int nfld, char *FB, int ldf, int nproma,
                       emi_stream_t st) {
  for (size_t c = 0; c < P.fclass.size(); c++) {
    FftClass &fc = P.fclass[c];
    if (fc.lats.empty() || nfld <= 0) continue;
    const int nchunk = (nfld + fc.fbk - 1) / fc.fbk;
    const long long nblocks = (long long)fc.lats.size() * nchunk;
    FftLaunchDev lc{fc.d_lats, (int)fc.lats.size(), nchunk, nblocks, adj ? 1 : 0, fc.d_rows};
    const int nthr = fc.nthr;
    if (g_pt.on) g_pt.fft_kernels++;  // every class that reaches this point launches exactly one kernel
    if (fc.gmem) {  // work arrays in global memory, one slice per workgroup
      const size_t needb = (size_t)nblocks * fc.gm_elems * 2 * P.esz;
      if (needb > P.cap_fftscr) {
        plan_quiesce(P);
        emi_stream_sync(0);
        emi_dev_free(P.d_fftscr);
        P.d_fftscr = nullptr;
        P.cap_fftscr = 0;
        void *q = nullptr;
        if (emi_dev_malloc(&q, needb)) EMI_FAIL(EMI_ERR_RUNTIME, "cannot allocate %.2f GiB of FFT scratch for the rows that exceed the LDS", needb / 1073741824.0);
        P.d_fftscr = (char *)q;
        P.cap_fftscr = needb;
      }
      if (inverse) {
        if (P.esz == 8)
          EMI_LAUNCH(emi_f64::k_fft_inv_gm, nblocks, nthr, 0, st, P.g, P.ftab, lc, d_flds, nfld, (const double *)FB, ldf, nproma, (d2 *)P.d_fftscr, fc.gm_elems);
        else
          EMI_LAUNCH(emi_f32::k_fft_inv_gm, nblocks, nthr, 0, st, P.g, P.ftab, lc, d_flds, nfld, (const float *)FB, ldf, nproma, (f2 *)P.d_fftscr, fc.gm_elems);
      } else {
        if (P.esz == 8)
          EMI_LAUNCH(emi_f64::k_fft_dir_gm, nblocks, nthr, 0, st, P.g, P.ftab, lc, d_flds, nfld, (double *)FB, ldf, nproma, (d2 *)P.d_fftscr, fc.gm_elems);
        else
          EMI_LAUNCH(emi_f32::k_fft_dir_gm, nblocks, nthr, 0, st, P.g, P.ftab, lc, d_flds, nfld, (float *)FB, ldf, nproma, (f2 *)P.d_fftscr, fc.gm_elems);
      }
      continue;
    }
    if (fc.mr) {
      if (inverse)
        EMI_LAUNCH_P(P.esz, k_fft_inv_mr, nblocks, nthr, fc.lds, st, P.g, P.ftab, lc, d_flds, nfld, (const RT *)FB, ldf, nproma);
      else
        EMI_LAUNCH_P(P.esz, k_fft_dir_mr, nblocks, nthr, fc.lds, st, P.g, P.ftab, lc, d_flds, nfld, (RT *)FB, ldf, nproma);
      continue;
    }
    if (fc.r16 && fc.split) {
      switch (fc.r16) {
#define EMI_R16P_LAUNCH(r_)                                                                                                                 \
  case r_:                                                                                                                                  \
    if (inverse)                                                                                                                            \
      EMI_LAUNCH_P(P.esz, k_fft_inv_r16p<r_>, nblocks, nthr, fc.lds, st, P.g, P.ftab, lc, d_flds, nfld, (const RT *)FB, ldf, nproma);      \
    else                                                                                                                                    \
      EMI_LAUNCH_P(P.esz, k_fft_dir_r16p<r_>, nblocks, nthr, fc.lds, st, P.g, P.ftab, lc, d_flds, nfld, (RT *)FB, ldf, nproma);            \
    break;
        EMI_R16S_LIST(EMI_R16P_LAUNCH)
#undef EMI_R16P_LAUNCH
        default: EMI_FAIL(EMI_ERR_RUNTIME, "internal: no k_fft_*_r16p kernel for R1 = %d", fc.r16);
      }
      continue;
    }
    if (fc.r16) {
      switch (fc.r16) {
#define EMI_R16_LAUNCH(r_)                                                                                                                  \
  case r_:                                                                                                                                  \
    if (inverse)                                                                                                                            \
      EMI_LAUNCH_P(P.esz, k_fft_inv_r16<r_>, nblocks, nthr, fc.lds, st, P.g, P.ftab, lc, d_flds, nfld, (const RT *)FB, ldf, nproma);       \
    else                                                                                                                                    \
      EMI_LAUNCH_P(P.esz, k_fft_dir_r16<r_>, nblocks, nthr, fc.lds, st, P.g, P.ftab, lc, d_flds, nfld, (RT *)FB, ldf, nproma);             \
    break;
        EMI_R16_LIST(EMI_R16_LAUNCH)
#undef EMI_R16_LAUNCH
        default: EMI_FAIL(EMI_ERR_RUNTIME, "internal: no k_fft_*_r16 kernel for R1 = %d", fc.r16);
      }
      continue;
    }
    switch (fc.hot) {
#define EMI_HOT_LAUNCH(pc_, S_, nf_, a_, b_, c_, d_, e_, nfl_)                                                                                       \
  case pc_:                                                                                                                                    \
    if (inverse)                                                                                                                               \
      EMI_LAUNCH_P(P.esz, k_fft_inv_hot<pc_>, nblocks, nthr, fc.lds, st, P.g, P.ftab, lc, d_flds, nfld, (const RT *)FB, ldf, nproma);         \
    else                                                                                                                                       \
      EMI_LAUNCH_P(P.esz, k_fft_dir_hot<pc_>, nblocks, nthr, fc.lds, st, P.g, P.ftab, lc, d_flds, nfld, (RT *)FB, ldf, nproma);               \
    break;
      EMI_HOT_PLAN_LIST(EMI_HOT_LAUNCH)
#undef EMI_HOT_LAUNCH
      default:
        if (inverse)
          EMI_LAUNCH_P(P.esz, k_fft_inv, nblocks, nthr, fc.lds, st, P.g, P.ftab, lc, d_flds, nfld, (const RT *)FB, ldf, nproma);
        else
          EMI_LAUNCH_P(P.esz, k_fft_dir, nblocks, nthr, fc.lds, st, P.g, P.ftab, lc, d_flds, nfld, (RT *)FB, ldf, nproma);
    }
  }
  return 0;
}

// Two library-owned streams software-pipeline the field batches of one call: the Legendre kernels
// (MFMA-bound) of batch b+1 run on stream A while the FFT kernels (fp64-VALU/LDS-bound) of batch b
// run on stream B; the Fourier buffer is double-buffered.  Both are forked from / joined to the
// caller's stream with events, so the call keeps ordinary stream semantics.
struct Pipeline {
#ifndef EMI_CPU_EMU
  hipStream_t sA = nullptr, sB = nullptr, sX = nullptr;  // Legendre, FFT, exchange
  std::vector<hipEvent_t> ev;
  hipEvent_t fork = nullptr, joinA = nullptr, joinB = nullptr, joinX = nullptr;
  int init() {
    if (sA) return 0;
    EMI_CHECK(hipStreamCreateWithFlags(&sA, hipStreamNonBlocking));
    EMI_CHECK(hipStreamCreateWithFlags(&sB, hipStreamNonBlocking));
    EMI_CHECK(hipStreamCreateWithFlags(&sX, hipStreamNonBlocking));
    EMI_CHECK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    EMI_CHECK(hipEventCreateWithFlags(&joinA, hipEventDisableTiming));
    EMI_CHECK(hipEventCreateWithFlags(&joinB, hipEventDisableTiming));
    EMI_CHECK(hipEventCreateWithFlags(&joinX, hipEventDisableTiming));
    return 0;
  }
  hipEvent_t event(size_t i) {
    while (ev.size() <= i) {
      hipEvent_t e;
      (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
      ev.push_back(e);
    }
    return ev[i];
  }
  void begin(emi_stream_t user) {
    (void)hipEventRecord(fork, user);
    (void)hipStreamWaitEvent(sA, fork, 0);
    (void)hipStreamWaitEvent(sB, fork, 0);
    (void)hipStreamWaitEvent(sX, fork, 0);
  }
  void end(emi_stream_t user) {
    (void)hipEventRecord(joinA, sA);
    (void)hipEventRecord(joinB, sB);
    (void)hipEventRecord(joinX, sX);
    (void)hipStreamWaitEvent(user, joinA, 0);
    (void)hipStreamWaitEvent(user, joinB, 0);
    (void)hipStreamWaitEvent(user, joinX, 0);
  }
  void signal(size_t i, emi_stream_t s) { (void)hipEventRecord(event(i), s); }
  void wait(size_t i, emi_stream_t s) { (void)hipStreamWaitEvent(s, event(i), 0); }
#else
  void *sA = nullptr, *sB = nullptr, *sX = nullptr;
  int init() { return 0; }
  void begin(emi_stream_t) {}
  void end(emi_stream_t) {}
  void signal(size_t, emi_stream_t) {}
  void wait(size_t, emi_stream_t) {}
#endif
};
static Pipeline g_pipe;

// number of field batches to pipeline (1 = plain sequential execution on the caller's stream)
static int pipeline_depth(const Plan &P, int nfields) {
  // one task: sequential.  Co-running the two kernel families of neighbouring field batches was measured SLOWER on MI355X
  // (TCo1279/KF=1645: 555 vs 531 ms per pair with 4 batches; round 2: 400.7 vs 391.7) -- the FFT kernels need all 16 waves per CU
  // to hide latency and the Legendre kernels lose MFMA issue slots.
  const int cfg = (test_paths() & 2) ? 4 : 1;
#ifdef EMI_CPU_EMU
  (void)P;
  (void)nfields;
  return 1;
#else
  if (nfields < 256) return 1;
  if (P.nproc > 1) {
    // several tasks: the all-to-all-v of batch b (xGMI) runs on its own stream under the Legendre kernels
    // of batch b+1 and the FFT kernels of batch b-1 -- with 2 or 4 GPUs the exchange is as long as the
    // compute (one or three links per GPU), so hiding it is worth more than the co-running penalty.
    // EMI_PIPELINE_DIST: batches per call (1 = sequential), default 4
    static int dcfg = -1;
    if (dcfg < 0) {
      const char *e = getenv("EMI_PIPELINE_DIST");
      dcfg = e ? atoi(e) : 4;
      if (dcfg < 1) dcfg = 1;
    }
    return dcfg;
  }
  return cfg;
#endif
}

#ifndef EMI_CPU_EMU
static int set_lds_attrs() {
  static bool done = false;
  if (done) return 0;
  EMI_CHECK(hipFuncSetAttribute((const void *)emi_f64::k_leg_dir, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  EMI_CHECK(hipFuncSetAttribute((const void *)emi_f32::k_leg_dir, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
#define EMI_HOT_ATTR(pc_, S_, nf_, a_, b_, c_, d_, e_, nfl_)                                                                                  \
  EMI_CHECK(hipFuncSetAttribute((const void *)emi_f64::k_fft_inv_hot<pc_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));  \
  EMI_CHECK(hipFuncSetAttribute((const void *)emi_f64::k_fft_dir_hot<pc_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));  \
  EMI_CHECK(hipFuncSetAttribute((const void *)emi_f32::k_fft_inv_hot<pc_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));  \
  EMI_CHECK(hipFuncSetAttribute((const void *)emi_f32::k_fft_dir_hot<pc_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  EMI_HOT_PLAN_LIST(EMI_HOT_ATTR)
#undef EMI_HOT_ATTR
  EMI_CHECK(hipFuncSetAttribute((const void *)emi_f64::k_fft_inv, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  EMI_CHECK(hipFuncSetAttribute((const void *)emi_f64::k_fft_dir, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  EMI_CHECK(hipFuncSetAttribute((const void *)emi_f32::k_fft_inv, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  EMI_CHECK(hipFuncSetAttribute((const void *)emi_f32::k_fft_dir, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  // direct mixed-radix plans hold up to 160 KiB per field (mr_choose) and have no global-scratch variant: fp64 rows such as
  // n = 8704 (half-length 4352 = 16 x 16 x 17) need 69 888 B
  EMI_CHECK(hipFuncSetAttribute((const void *)emi_f64::k_fft_inv_mr, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  EMI_CHECK(hipFuncSetAttribute((const void *)emi_f64::k_fft_dir_mr, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  EMI_CHECK(hipFuncSetAttribute((const void *)emi_f32::k_fft_inv_mr, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  EMI_CHECK(hipFuncSetAttribute((const void *)emi_f32::k_fft_dir_mr, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
#define EMI_R16_ATTR(r_)                                                                                                                     \
  EMI_CHECK(hipFuncSetAttribute((const void *)emi_f64::k_fft_inv_r16<r_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));  \
  EMI_CHECK(hipFuncSetAttribute((const void *)emi_f64::k_fft_dir_r16<r_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));  \
  EMI_CHECK(hipFuncSetAttribute((const void *)emi_f32::k_fft_inv_r16<r_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));  \
  EMI_CHECK(hipFuncSetAttribute((const void *)emi_f32::k_fft_dir_r16<r_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  EMI_R16_LIST(EMI_R16_ATTR)
#undef EMI_R16_ATTR
#define EMI_R16P_ATTR(r_)                                                                                                                    \
  EMI_CHECK(hipFuncSetAttribute((const void *)emi_f64::k_fft_inv_r16p<r_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
  EMI_CHECK(hipFuncSetAttribute((const void *)emi_f64::k_fft_dir_r16p<r_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
  EMI_CHECK(hipFuncSetAttribute((const void *)emi_f32::k_fft_inv_r16p<r_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
  EMI_CHECK(hipFuncSetAttribute((const void *)emi_f32::k_fft_dir_r16p<r_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  EMI_R16S_LIST(EMI_R16P_ATTR)
#undef EMI_R16P_ATTR
  done = true;
  return 0;
}
#else
static int set_lds_attrs() { return 0; }
#endif

// scalar field enumeration shared by both directions: PSPSCALAR, or PSPSC2 + PSPSC3A + PSPSC3B
// (ltinv_mod.F90:203-240 / updsp_mod.F90:128-160 ordering)
struct ScalarRef {
  int arr;  // 0 scalar, 1 sc2, 2 sc3a, 3 sc3b
  int lev, var;
};

template <class ARGS>
static int enumerate_scalars(const ARGS &a, const char *who, std::vector<ScalarRef> &sc) {
  sc.clear();
  if (a.spscalar) {
    if (a.spsc3a || a.spsc3b || a.spsc2) EMI_FAIL(EMI_ERR_ARG, "%s : PSPSCALAR AND PSPSC3A/PSPSC3B/PSPSC2 BOTH PRESENT", who);
    for (int i = 0; i < a.nf_scalar; i++) sc.push_back({0, i, 0});
  } else {
    if (a.spsc2)
      for (int i = 0; i < a.nf_sc2; i++) sc.push_back({1, i, 0});
    if (a.spsc3a)
      for (int v = 0; v < a.sc3a_nvar; v++)
        for (int l = 0; l < a.sc3a_nlev; l++) sc.push_back({2, l, v});
    if (a.spsc3b)
      for (int v = 0; v < a.sc3b_nvar; v++)
        for (int l = 0; l < a.sc3b_nlev; l++) sc.push_back({3, l, v});
  }
  return 0;
}

// The reference's extent checks (inv_trans.F90:476-600, dir_trans.F90:370-491) on the extents the caller reports.
// nvar_uv = IF_UV_PAR; dmul = 3 with LDSCDERS (IF_SC2_G, IF_SC3A_G3 and IF_SC3B_G3 are tripled, inv_trans.F90:371-376).
// *uv_dim3: the third extent of PGPUV to address with (>= nvar_uv).
template <class ARGS>
static int check_extents(const char *who, const ARGS &a, int nspec2, int nproma, int ngpblks, int nuv, int nvar_uv, int dmul,
                         int *uv_dim3) {
  *uv_dim3 = nvar_uv;
  const emi_extents_t *e = a.ext;
  if (!e) return 0;
  if (e->sp_dim2 > 0 && e->sp_dim2 < nspec2)
    EMI_FAIL(EMI_ERR_ARG, "%s : SECOND DIMENSION OF A SPECTRAL ARRAY TOO SMALL (%d < NSPEC2 = %d)", who, e->sp_dim2, nspec2);
  auto lead = [&](const char *nm, int d1) {
    if (d1 < nproma) {
      emi_set_error("%s:FIRST DIMENSION OF %s TOO SMALL (%d < %d)", who, nm, d1, nproma);
      return -1;
    }
    if (d1 != nproma) {
      emi_set_error("%s: first extent of %s is %d, NPROMA is %d: pass a packed copy at the C-ABI (the Fortran shim does)", who, nm, d1, nproma);
      return -1;
    }
    return 0;
  };
  if (a.gp) {
    if (lead("PGP", e->gp[0])) return EMI_ERR_ARG;
    if (e->gp[1] != a.gp_nfld) EMI_FAIL(EMI_ERR_ARG, "%s: gp_nfld (%d) is not the second extent of PGP (%d)", who, a.gp_nfld, e->gp[1]);
    if (e->gp[2] < ngpblks) EMI_FAIL(EMI_ERR_ARG, "%s:THIRD DIMENSION OF PGP TOO SMALL (%d < %d)", who, e->gp[2], ngpblks);
  }
  if (a.gpuv) {
    if (lead("PGPUV", e->gpuv[0])) return EMI_ERR_ARG;
    if (e->gpuv[1] != a.nf_uv) EMI_FAIL(EMI_ERR_ARG, "%s:SEC. DIMENSION OF PGPUV INCONSISTENT (%d, IF_UV_G = %d)", who, e->gpuv[1], a.nf_uv);
    if (e->gpuv[2] < nvar_uv) EMI_FAIL(EMI_ERR_ARG, "%s:THIRD DIMENSION OF PGPUV TOO SMALL (%d < %d)", who, e->gpuv[2], nvar_uv);
    if (e->gpuv[3] < ngpblks) EMI_FAIL(EMI_ERR_ARG, "%s:FOURTH DIMENSION OF PGPUV TOO SMALL (%d < %d)", who, e->gpuv[3], ngpblks);
    *uv_dim3 = e->gpuv[2];
    (void)nuv;
  }
  if (a.gp2 && a.nf_sc2 > 0) {
    if (lead("PGP2", e->gp2[0])) return EMI_ERR_ARG;
    if (e->gp2[1] != a.nf_sc2 * dmul) EMI_FAIL(EMI_ERR_ARG, "%s:SEC. DIMENSION OF PGP2 INCONSISTENT (%d, IF_SC2_G = %d)", who, e->gp2[1], a.nf_sc2 * dmul);
    if (e->gp2[2] < ngpblks) EMI_FAIL(EMI_ERR_ARG, "%s:THIRD DIMENSION OF PGP2 TOO SMALL (%d < %d)", who, e->gp2[2], ngpblks);
  }
  if (a.gp3a && a.sc3a_nlev * a.sc3a_nvar > 0) {
    if (lead("PGP3A", e->gp3a[0])) return EMI_ERR_ARG;
    if (e->gp3a[1] != a.sc3a_nlev) EMI_FAIL(EMI_ERR_ARG, "%s:SEC. DIMENSION OF PGP3A INCONSISTENT (%d, IF_SC3A_G2 = %d)", who, e->gp3a[1], a.sc3a_nlev);
    if (e->gp3a[2] != a.sc3a_nvar * dmul)
      EMI_FAIL(EMI_ERR_ARG, "%s:THIRD DIMENSION OF PGP3A INCONSISTENT (%d, IF_SC3A_G3 = %d)", who, e->gp3a[2], a.sc3a_nvar * dmul);
    if (e->gp3a[3] < ngpblks) EMI_FAIL(EMI_ERR_ARG, "%s:FOURTH DIMENSION OF PGP3A TOO SMALL (%d < %d)", who, e->gp3a[3], ngpblks);
  }
  if (a.gp3b && a.sc3b_nlev * a.sc3b_nvar > 0) {
    if (lead("PGP3B", e->gp3b[0])) return EMI_ERR_ARG;
    if (e->gp3b[1] != a.sc3b_nlev) EMI_FAIL(EMI_ERR_ARG, "%s:SEC. DIMENSION OF PGP3B INCONSISTENT (%d, IF_SC3B_G2 = %d)", who, e->gp3b[1], a.sc3b_nlev);
    if (e->gp3b[2] != a.sc3b_nvar * dmul)
      EMI_FAIL(EMI_ERR_ARG, "%s:THIRD DIMENSION OF PGP3B INCONSISTENT (%d, IF_SC3B_G3 = %d)", who, e->gp3b[2], a.sc3b_nvar * dmul);
    if (e->gp3b[3] < ngpblks) EMI_FAIL(EMI_ERR_ARG, "%s:FOURTH DIMENSION OF PGP3B TOO SMALL (%d < %d)", who, e->gp3b[3], ngpblks);
  }
  return 0;
}

// INV_TRANS, and DIR_TRANSAD when adj: the adjoint of DIR_TRANS (for the inner products of the
// reference's adjoint tests: plain sum over grid points, SPECNORM weights in spectral space) is the same
// spectral -> grid pipeline with the Gaussian weight and 1/NLOEN applied per latitude (ledirad_mod.F90:151,183,
// ftdirad_mod.F90:84-89) and the adjoint of UVTVD in place of VDTUV.
static int inv_trans_impl(int kresol, const emi_invtrans_t *ap, bool adj) {
  Plan *Pp = get_plan(kresol);
  if (!Pp) EMI_FAIL(EMI_ERR_STATE, "INV_TRANS: unknown resolution %d", kresol);
  if (!ap) EMI_FAIL(EMI_ERR_ARG, "INV_TRANS: null argument block");
  Plan &P = *Pp;
  const emi_invtrans_t &a = *ap;
  emi_stream_t st = (emi_stream_t)a.stream;
  const bool host = a.mem_space == EMI_MEM_HOST;
  // ---- field accounting (inv_trans.F90:230-387)
  const int nuv = (a.spvor || a.spdiv) ? a.nf_uv : 0;
  if (nuv > 0 && !a.spvor) EMI_FAIL(EMI_ERR_ARG, "INV_TRANS : IF_UV > 0 BUT PSPVOR MISSING");
  if (nuv > 0 && !a.spdiv) EMI_FAIL(EMI_ERR_ARG, "INV_TRANS : IF_UV > 0 BUT PSPDIV MISSING");
  std::vector<ScalarRef> sc;
  if (enumerate_scalars(a, "INV_TRANS", sc)) return EMI_ERR_ARG;
  const int nsc = (int)sc.size();
  bool lscders = a.ldscders && nsc > 0, lvorgp = a.ldvorgp != 0, ldivgp = a.lddivgp != 0, luvder = a.lduvder && nuv > 0;
  if (lvorgp) ldivgp = true;  // inv_trans.F90:350
  const int nproma = a.kproma > 0 ? a.kproma : P.ngptot;
  const int ngpblks = (P.ngptot - 1) / nproma + 1;
  int if_gp = 2 * nuv + nsc + (lscders ? 2 * nsc : 0) + ((nuv && lvorgp) ? nuv : 0) + ((nuv && ldivgp) ? nuv : 0) + (luvder ? 2 * nuv : 0);
  if (a.gp) {
    if (a.gpuv || a.gp3a || a.gp3b || a.gp2) EMI_FAIL(EMI_ERR_ARG, "INV_TRANS:PGP AND PGPUV/PGP3A/PGP3B/PGP2 CAN NOT BOTH BE PRESENT");
    if (a.gp_nfld < if_gp) EMI_FAIL(EMI_ERR_ARG, "INV_TRANS:SECOND DIMENSION OF PGP TOO SMALL (%d < %d)", a.gp_nfld, if_gp);
  } else {
    if (nuv > 0 && !a.gpuv) EMI_FAIL(EMI_ERR_ARG, "INV_TRANS:PGPUV MISSING");
    if (a.spscalar && nsc > 0) EMI_FAIL(EMI_ERR_ARG, "INV_TRANS:PGP MISSING (PSPSCALAR needs PGP)");
    if (a.spsc2 && a.nf_sc2 > 0 && !a.gp2) EMI_FAIL(EMI_ERR_ARG, "INV_TRANS:PGP2 MISSING");
    if (a.spsc3a && a.sc3a_nlev * a.sc3a_nvar > 0 && !a.gp3a) EMI_FAIL(EMI_ERR_ARG, "INV_TRANS:PGP3A MISSING");
    if (a.spsc3b && a.sc3b_nlev * a.sc3b_nvar > 0 && !a.gp3b) EMI_FAIL(EMI_ERR_ARG, "INV_TRANS:PGP3B MISSING");
  }
  if (if_gp == 0) return EMI_SUCCESS;
  const int nvar_uv = ((nuv && lvorgp) ? 1 : 0) + ((nuv && ldivgp) ? 1 : 0) + 2 + (luvder ? 2 : 0);  // IF_UV_PAR
  const int dmul = lscders ? 3 : 1;
  int uv_dim3 = nvar_uv;
  if (check_extents(adj ? "DIR_TRANSAD" : "INV_TRANS", a, P.nspec2, nproma, ngpblks, nuv, nvar_uv, dmul, &uv_dim3)) return EMI_ERR_ARG;
  if (set_lds_attrs()) return EMI_ERR_RUNTIME;
  if (plan_begin(P, st)) return EMI_ERR_RUNTIME;

  // ---- stage host arrays
  HostStage hs(P.esz);
  const size_t ns2 = P.nspec2;
  const void *d_vor = hs.in(a.spvor, ns2 * a.nf_uv, host && nuv, st), *d_div = hs.in(a.spdiv, ns2 * a.nf_uv, host && nuv, st);
  const void *d_sc[4] = {hs.in(a.spscalar, ns2 * a.nf_scalar, host, st), hs.in(a.spsc2, ns2 * a.nf_sc2, host, st),
                           hs.in(a.spsc3a, ns2 * a.sc3a_nlev * a.sc3a_nvar, host, st),
                           hs.in(a.spsc3b, ns2 * a.sc3b_nlev * a.sc3b_nvar, host, st)};
  const size_t gsz = (size_t)nproma * ngpblks;
  const bool gpad = gsz != (size_t)P.ngptot;  // last NPROMA block padded: those elements are not written
  void *d_gp = hs.out(a.gp, gsz * a.gp_nfld, host, gpad || a.gp_nfld > if_gp, st);
  void *d_gpuv = hs.out(a.gpuv, gsz * nuv * uv_dim3, host && nuv, gpad || uv_dim3 > nvar_uv, st);
  void *d_gp2 = hs.out(a.gp2, gsz * a.nf_sc2 * dmul, host, gpad, st);
  void *d_gp3a = hs.out(a.gp3a, gsz * a.sc3a_nlev * a.sc3a_nvar * dmul, host, gpad, st);
  void *d_gp3b = hs.out(a.gp3b, gsz * a.sc3b_nlev * a.sc3b_nvar * dmul, host, gpad, st);
  if (hs.failed) EMI_FAIL(EMI_ERR_RUNTIME, "INV_TRANS: cannot stage the host arrays through device memory (%s)", emi_last_error());

  // ---- Legendre-space fields (ltinv_mod.F90:166-262): [vor][div] u v scalars [nsders]
  std::vector<SpecSrc> lt;
  auto sc_src = [&](const ScalarRef &r, int kind) {
    SpecSrc s{};
    s.kind = kind;
    switch (r.arr) {
      case 0: s.a = d_sc[0]; s.sa = a.nf_scalar; s.ia = r.lev; break;
      case 1: s.a = d_sc[1]; s.sa = a.nf_sc2; s.ia = r.lev; break;
      case 2: s.a = (const char *)d_sc[2] + (size_t)r.var * ns2 * a.sc3a_nlev * P.esz; s.sa = a.sc3a_nlev; s.ia = r.lev; break;
      default: s.a = (const char *)d_sc[3] + (size_t)r.var * ns2 * a.sc3b_nlev * P.esz; s.sa = a.sc3b_nlev; s.ia = r.lev; break;
    }
    return s;
  };
  int i_vor = -1, i_div = -1, i_u = -1, i_v = -1, i_sc = -1, i_nsd = -1;
  if (nuv) {
    auto uvsrc = [&](int i, int kind) {
      SpecSrc s{};
      s.kind = kind;
      s.a = d_vor; s.sa = a.nf_uv; s.ia = i;
      s.b = d_div; s.sb = a.nf_uv; s.ib = i;
      if (kind == SPK_COPY + 100) { s.kind = SPK_COPY; s.a = d_div; }
      return s;
    };
    if (lvorgp) { i_vor = (int)lt.size(); for (int i = 0; i < nuv; i++) lt.push_back(uvsrc(i, SPK_COPY)); }
    if (ldivgp) { i_div = (int)lt.size(); for (int i = 0; i < nuv; i++) lt.push_back(uvsrc(i, SPK_COPY + 100)); }
    i_u = (int)lt.size(); for (int i = 0; i < nuv; i++) lt.push_back(uvsrc(i, adj ? SPK_U_AD : SPK_U));
    i_v = (int)lt.size(); for (int i = 0; i < nuv; i++) lt.push_back(uvsrc(i, adj ? SPK_V_AD : SPK_V));
  }
  if (nsc) {
    i_sc = (int)lt.size();
    for (auto &r : sc) lt.push_back(sc_src(r, SPK_COPY));
    if (lscders) { i_nsd = (int)lt.size(); for (auto &r : sc) lt.push_back(sc_src(r, SPK_NSD)); }
  }
  const int nlt = (int)lt.size();
  // ---- grid fields (ftinv_ctl_mod.F90:228-262 order) with their destination arrays
  struct GOut { GridFld g; int lt; };
  std::vector<GOut> gout;
  int gcount = 0;
  auto dest_uv = [&](int var, int lev) {
    GridFld g{};
    if (d_gp) { g.base = d_gp; g.nf_arr = a.gp_nfld; g.fidx = gcount; }
    else { g.base = d_gpuv; g.nf_arr = nuv * uv_dim3; g.fidx = var * nuv + lev; }
    return g;
  };
  auto dest_sc = [&](int isc, int kder) {
    GridFld g{};
    if (d_gp) { g.base = d_gp; g.nf_arr = a.gp_nfld; g.fidx = gcount; return g; }
    const ScalarRef &r = sc[isc];
    if (r.arr == 1) { g.base = d_gp2; g.nf_arr = a.nf_sc2 * dmul; g.fidx = r.lev + kder * a.nf_sc2; }
    else if (r.arr == 2) { g.base = d_gp3a; g.nf_arr = a.sc3a_nlev * a.sc3a_nvar * dmul; g.fidx = (r.var + kder * a.sc3a_nvar) * a.sc3a_nlev + r.lev; }
    else { g.base = d_gp3b; g.nf_arr = a.sc3b_nlev * a.sc3b_nvar * dmul; g.fidx = (r.var + kder * a.sc3b_nvar) * a.sc3b_nlev + r.lev; }
    return g;
  };
  int uvvar = 0;
  auto push = [&](GridFld g, int mode, int src) { g.mode = mode; g.src = src; gout.push_back({g, src}); gcount++; };
  if (nuv) {
    if (lvorgp) { for (int i = 0; i < nuv; i++) push(dest_uv(uvvar, i), GM_PLAIN, i_vor + i); uvvar++; }
    if (ldivgp) { for (int i = 0; i < nuv; i++) push(dest_uv(uvvar, i), GM_PLAIN, i_div + i); uvvar++; }
    for (int i = 0; i < nuv; i++) push(dest_uv(uvvar, i), GM_ACOS, i_u + i);
    uvvar++;
    for (int i = 0; i < nuv; i++) push(dest_uv(uvvar, i), GM_ACOS, i_v + i);
    uvvar++;
  }
  for (int i = 0; i < nsc; i++) push(dest_sc(i, 0), GM_PLAIN, i_sc + i);
  if (lscders) for (int i = 0; i < nsc; i++) push(dest_sc(i, 1), GM_ACOS, i_nsd + i);
  if (luvder) {
    for (int i = 0; i < nuv; i++) push(dest_uv(uvvar, i), GM_EWDER_UV, i_u + i);
    uvvar++;
    for (int i = 0; i < nuv; i++) push(dest_uv(uvvar, i), GM_EWDER_UV, i_v + i);
    uvvar++;
  }
  if (lscders) for (int i = 0; i < nsc; i++) push(dest_sc(i, 2), GM_EWDER, i_sc + i);

  // ---- batches over Legendre-space fields, software-pipelined over two streams
  const int depth = pipeline_depth(P, nlt);
  const int bsz = pick_batch(P, nlt, depth);
  const int nbat = (nlt + bsz - 1) / bsz;
  const bool piped = depth > 1 && nbat > 1;
  // The 64-field column tiles of the call are dealt evenly to the batches and every batch has its own row width
  // (2 x its tiles x 64 reals): no batch computes, stores or exchanges columns of another batch's width (with
  // batches of ceil(fields / nbat) rounded up to 64 the 4 x 448 columns of a 1645-field call were 9 % padding).
  const int tiles_total = (nlt + 63) / 64;
  const int bfpad = 64 * ((tiles_total + nbat - 1) / nbat);  // widest batch
  if (ensure_work(P, bfpad, piped ? 2 : 1, st)) return EMI_ERR_RUNTIME;
  const int ldw_max = 2 * bfpad;
  // all descriptors of the call in one upload
  struct Bat { size_t off_l, off_g; int nl, ng, ldw; };
  std::vector<Bat> bats;
  std::vector<char> hdesc;
  for (int ibat = 0, b0 = 0; ibat < nbat; ibat++) {
    const int tiles_b = tiles_total / nbat + (ibat < tiles_total % nbat ? 1 : 0);
    const int nb = std::min(64 * tiles_b, nlt - b0);
    std::vector<GridFld> bg;
    for (auto &go : gout)
      if (go.lt >= b0 && go.lt < b0 + nb) {
        GridFld g = go.g;
        g.src = go.lt - b0;
        bg.push_back(g);
      }
    Bat bt{};
    bt.nl = nb;
    bt.ldw = 2 * 64 * tiles_b;
    bt.ng = (int)bg.size();
    bt.off_l = hdesc.size();
    hdesc.resize(bt.off_l + ((size_t)nb * sizeof(SpecSrc) + 255) / 256 * 256);
    memcpy(hdesc.data() + bt.off_l, lt.data() + b0, (size_t)nb * sizeof(SpecSrc));
    bt.off_g = hdesc.size();
    hdesc.resize(bt.off_g + (bg.size() * sizeof(GridFld) + 255) / 256 * 256);
    if (!bg.empty()) memcpy(hdesc.data() + bt.off_g, bg.data(), bg.size() * sizeof(GridFld));
    bats.push_back(bt);
    b0 += nb;
  }
  // Legendre tile maps per batch: a batch with fewer fields than the row width (the last one of a call) only
  // gets the column tiles that hold fields (built before anything is queued or forked: a new map is a blocking upload)
  std::vector<LegMaps *> bmaps(nbat, nullptr);
  for (int ib = 0; ib < nbat; ib++)
    if (leg_tilemaps(P, (bats[ib].nl + 63) / 64, &bmaps[ib])) return EMI_ERR_RUNTIME;
  if (upload_desc(P, hdesc, st)) return EMI_ERR_RUNTIME;
  emi_stream_t sA = st, sB = st, sX = st;
  if (piped) {
    if (g_pipe.init()) return EMI_ERR_RUNTIME;
    sA = (emi_stream_t)g_pipe.sA;
    sB = (emi_stream_t)g_pipe.sB;
    sX = (emi_stream_t)g_pipe.sX;
    g_pipe.begin(st);
  }
  g_pt.begin(G.profile != 0, G.profile == 2);
  // Events of batch ib: 3 ib = Legendre done, 3 ib + 1 = FFT done, 3 ib + 2 = exchange done.  Both
  // Fourier buffers are double buffered ([ib & 1]); one task: FBf == FBl and there is no exchange.
  const bool dist = P.nproc > 1;
  // buffer strides for the widest batch (+ the zero row DIR_TRANS keeps behind the Legendre-side rows)
  const size_t lstride = (size_t)((dist ? P.lrows : P.frows) + 1) * ldw_max * P.esz, fstride = (size_t)P.frows * ldw_max * P.esz;
  for (int ib = 0; ib < nbat; ib++) {
    const Bat &bt = bats[ib];
    const int ldw = bt.ldw, bfpad_b = bt.ldw / 2;  // this batch's row width
    const SpecSrc *d_bl = (const SpecSrc *)((char *)P.d_desc + bt.off_l);
    const GridFld *d_bg = (const GridFld *)((char *)P.d_desc + bt.off_g);
    char *FBl = P.d_FBL + (piped ? (size_t)(ib & 1) * lstride : 0);
    char *FBf = dist ? P.d_FBF + (piped ? (size_t)(ib & 1) * fstride : 0) : FBl;
    // stream A: spectral pack + Legendre; FBl[ib&1] was last read by the FFT (one task) or by the
    // exchange (several tasks) of batch ib-2
    if (piped && ib >= 2) g_pipe.wait(3 * (ib - 2) + (dist ? 2 : 1), sA);
    int iv;
    {
      EmiRange rg(EMI_LBL_LTINV);  // GSTATS 102: PRFI1B / VDTUV / SPNSDE + LEINV + ASRE1B
      iv = g_pt.start(0, sA);
      {
        long long nblk = (long long)P.wrows_total * ((bfpad_b + 255) / 256);
        EMI_LAUNCH_P(P.esz, k_prepack_inv, nblk, 256, 0, sA, P.g, d_bl, bt.nl, bfpad_b, (RT *)P.d_W, ldw, (long long)P.wrows_total);
      }
      g_pt.stop(iv, sA);
      iv = g_pt.start(1, sA);
      LegMaps *lmaps = bmaps[ib];
      // (the wide tiles first: they are the longest of the call)
      if (lmaps->n_inv_wide > 0)
        EMI_LAUNCH(emi_f32::k_leg_inv_wide, lmaps->n_inv_wide, LG_THREADS, LG_LDS_BYTES + 512, sA, P.g, (const int2 *)lmaps->d_inv_wide, (const float *)P.d_W, ldw, (float *)FBl, ldw);
      if (lmaps->n_inv > 0)
        EMI_LAUNCH_P(P.esz, k_leg_inv, lmaps->n_inv, LG_THREADS, LG_LDS_BYTES + 512, sA, P.g, (const int2 *)lmaps->d_inv, (const RT *)P.d_W, ldw, (RT *)FBl, ldw);
      g_pt.stop(iv, sA);
    }
    if (piped) g_pipe.signal(3 * ib, sA);
    if (dist) {
      // stream X: TRMTOL; FBf[ib&1] was last read by the FFT of batch ib-2
      if (piped) g_pipe.wait(3 * ib, sX);
      if (piped && ib >= 2) g_pipe.wait(3 * (ib - 2) + 1, sX);
      EmiRange rg(EMI_LBL_TRMTOL);  // GSTATS 152
      if (exchange(P, true, ldw, sX, FBl, FBf)) {
        if (piped) g_pipe.end(st);  // the three streams were forked from the caller's: join them before giving up
        plan_end(P, st);
        return EMI_ERR_RUNTIME;
      }
      if (piped) g_pipe.signal(3 * ib + 2, sX);
    }
    // stream B: FFTs
    if (piped) g_pipe.wait(3 * ib + (dist ? 2 : 0), sB);
    EmiRange rgf(EMI_LBL_FTINV);  // GSTATS 107: FOURIER_IN + FSC + FTINV + TRLTOG
    iv = g_pt.start(2, sB);
    if (launch_fft(P, true, adj, d_bg, bt.ng, FBf, ldw, nproma, sB)) {
      if (piped) g_pipe.end(st);
      plan_end(P, st);
      return EMI_ERR_RUNTIME;
    }
    g_pt.stop(iv, sB);
    if (piped) g_pipe.signal(3 * ib + 1, sB);
  }
  if (piped) g_pipe.end(st);
  if (plan_end(P, st)) return EMI_ERR_RUNTIME;
  if (host) hs.flush(st);
#ifndef EMI_CPU_EMU
  EMI_CHECK(hipGetLastError());
#endif
  return EMI_SUCCESS;
}

// DIR_TRANS, and INV_TRANSAD when adj: the adjoint of INV_TRANS is the same grid -> spectral pipeline
// without the Gaussian weight and the 1/NLOEN (ftinvad_mod.F90:77-83: "change of metric") and with the
// adjoint of VDTUV (= -RLAPIN x UVTVD) in place of UVTVD.
struct AdjOpts {  // INV_TRANSAD: the options of the INV_TRANS it is the adjoint of
  bool scders = false, vorgp = false, divgp = false, uvder = false;
};
static int dir_trans_impl(int kresol, const emi_dirtrans_t *ap, bool adj, const AdjOpts *ao = nullptr) {
  Plan *Pp = get_plan(kresol);
  if (!Pp) EMI_FAIL(EMI_ERR_STATE, "DIR_TRANS: unknown resolution %d", kresol);
  if (!ap) EMI_FAIL(EMI_ERR_ARG, "DIR_TRANS: null argument block");
  Plan &P = *Pp;
  const emi_dirtrans_t &a = *ap;
  emi_stream_t st = (emi_stream_t)a.stream;
  const bool host = a.mem_space == EMI_MEM_HOST;
  const int nuv = (a.spvor || a.spdiv) ? a.nf_uv : 0;
  if (nuv > 0 && (!a.spvor || !a.spdiv)) EMI_FAIL(EMI_ERR_ARG, "DIR_TRANS : IF_UV > 0 BUT PSPVOR OR PSPDIV MISSING");
  std::vector<ScalarRef> sc;
  if (enumerate_scalars(a, "DIR_TRANS", sc)) return EMI_ERR_ARG;
  const int nsc = (int)sc.size();
  const int nproma = a.kproma > 0 ? a.kproma : P.ngptot;
  const int ngpblks = (P.ngptot - 1) / nproma + 1;
  // INV_TRANSAD with derivative / vorticity / divergence inputs: the grid arrays have INV_TRANS's layout
  // (inv_trans.F90:352-387: [vor][div] u v scalars [N-S derivatives] [u, v E-W derivatives] [scalar E-W derivatives])
  const bool a_scd = ao && ao->scders && nsc > 0, a_uvd = ao && ao->uvder && nuv > 0;
  const bool a_div = ao && (ao->divgp || ao->vorgp) && nuv > 0, a_vor = ao && ao->vorgp && nuv > 0;
  const int nvar_uv = 2 + (a_vor ? 1 : 0) + (a_div ? 1 : 0) + (a_uvd ? 2 : 0), dmul = a_scd ? 3 : 1;
  const int if_gp = 2 * nuv + nsc + (a_scd ? 2 * nsc : 0) + (a_vor ? nuv : 0) + (a_div ? nuv : 0) + (a_uvd ? 2 * nuv : 0);  // dir_trans.F90:303
  if (a.gp) {
    if (a.gpuv || a.gp3a || a.gp3b || a.gp2) EMI_FAIL(EMI_ERR_ARG, "DIR_TRANS:PGP AND PGPUV/PGP3A/PGP3B/PGP2 CAN NOT BOTH BE PRESENT");
    if (a.gp_nfld < if_gp) EMI_FAIL(EMI_ERR_ARG, "DIR_TRANS:SECOND DIMENSION OF PGP TOO SMALL (%d < %d)", a.gp_nfld, if_gp);
  } else {
    if (nuv > 0 && !a.gpuv) EMI_FAIL(EMI_ERR_ARG, "DIR_TRANS:PGPUV MISSING");
    if (a.spscalar && nsc > 0) EMI_FAIL(EMI_ERR_ARG, "DIR_TRANS:PGP MISSING (PSPSCALAR needs PGP)");
    if (a.spsc2 && a.nf_sc2 > 0 && !a.gp2) EMI_FAIL(EMI_ERR_ARG, "DIR_TRANS:PGP2 MISSING");
    if (a.spsc3a && a.sc3a_nlev * a.sc3a_nvar > 0 && !a.gp3a) EMI_FAIL(EMI_ERR_ARG, "DIR_TRANS:PGP3A MISSING");
    if (a.spsc3b && a.sc3b_nlev * a.sc3b_nvar > 0 && !a.gp3b) EMI_FAIL(EMI_ERR_ARG, "DIR_TRANS:PGP3B MISSING");
  }
  if (if_gp == 0) return EMI_SUCCESS;
  int uv_dim3 = nvar_uv;
  if (check_extents(adj ? "INV_TRANSAD" : "DIR_TRANS", a, P.nspec2, nproma, ngpblks, nuv, nvar_uv, dmul, &uv_dim3)) return EMI_ERR_ARG;
  if (set_lds_attrs()) return EMI_ERR_RUNTIME;
  if (plan_begin(P, st)) return EMI_ERR_RUNTIME;

  HostStage hs(P.esz);
  const size_t ns2 = P.nspec2, gsz = (size_t)nproma * ngpblks;
  void *d_vor = hs.out(a.spvor, ns2 * a.nf_uv, host && nuv), *d_div = hs.out(a.spdiv, ns2 * a.nf_uv, host && nuv);
  void *d_sc[4] = {hs.out(a.spscalar, ns2 * a.nf_scalar, host), hs.out(a.spsc2, ns2 * a.nf_sc2, host),
                     hs.out(a.spsc3a, ns2 * a.sc3a_nlev * a.sc3a_nvar, host), hs.out(a.spsc3b, ns2 * a.sc3b_nlev * a.sc3b_nvar, host)};
  const void *d_gp = hs.in(a.gp, gsz * a.gp_nfld, host, st);
  const void *d_gpuv = hs.in(a.gpuv, gsz * nuv * uv_dim3, host && nuv, st);
  const void *d_gp2 = hs.in(a.gp2, gsz * a.nf_sc2 * dmul, host, st);
  const void *d_gp3a = hs.in(a.gp3a, gsz * a.sc3a_nlev * a.sc3a_nvar * dmul, host, st);
  const void *d_gp3b = hs.in(a.gp3b, gsz * a.sc3b_nlev * a.sc3b_nvar * dmul, host, st);
  if (hs.failed) EMI_FAIL(EMI_ERR_RUNTIME, "DIR_TRANS: cannot stage the host arrays through device memory (%s)", emi_last_error());

  // Fourier-space fields: u(nuv) v(nuv) scalars (dir_trans.F90:301, ftdir_ctl_mod.F90); INV_TRANSAD with options adds
  // the grid vorticity / divergence, the N-S derivatives of the scalars and the E-W derivatives as further fields
  std::vector<GridFld> gin;
  int gcount = 0, uvvar = 0;
  auto src_uv = [&](int i, int mode) {  // next u/v-shaped variable of PGP / PGPUV
    GridFld g{};
    if (d_gp) { g.base = (void *)d_gp; g.nf_arr = a.gp_nfld; g.fidx = gcount; }
    else { g.base = (void *)d_gpuv; g.nf_arr = nuv * uv_dim3; g.fidx = uvvar * nuv + i; }
    g.mode = mode;
    gcount++;
    return g;
  };
  auto src_sc = [&](int isc, int kder, int mode) {  // scalar isc, derivative block kder (0 value, 1 N-S, 2 E-W: trltog_mod.F90:632-690)
    GridFld g{};
    const ScalarRef &r = sc[isc];
    if (d_gp) { g.base = (void *)d_gp; g.nf_arr = a.gp_nfld; g.fidx = gcount; }
    else if (r.arr == 1) { g.base = (void *)d_gp2; g.nf_arr = a.nf_sc2 * dmul; g.fidx = r.lev + kder * a.nf_sc2; }
    else if (r.arr == 2) { g.base = (void *)d_gp3a; g.nf_arr = a.sc3a_nlev * a.sc3a_nvar * dmul; g.fidx = (r.var + kder * a.sc3a_nvar) * a.sc3a_nlev + r.lev; }
    else { g.base = (void *)d_gp3b; g.nf_arr = a.sc3b_nlev * a.sc3b_nvar * dmul; g.fidx = (r.var + kder * a.sc3b_nvar) * a.sc3b_nlev + r.lev; }
    g.mode = mode;
    gcount++;
    return g;
  };
  // field numbers: u_i = i, v_i = nuv + i, scalar_j = 2 nuv + j (as DIR_TRANS), then the adjoint-only inputs
  std::vector<GridFld> g_vor, g_div, g_u, g_v, g_sc, g_ns, g_uew, g_vew, g_scew;
  if (nuv) {
    if (a_vor) { for (int i = 0; i < nuv; i++) g_vor.push_back(src_uv(i, GM_PLAIN)); uvvar++; }
    if (a_div) { for (int i = 0; i < nuv; i++) g_div.push_back(src_uv(i, GM_PLAIN)); uvvar++; }
    for (int i = 0; i < nuv; i++) g_u.push_back(src_uv(i, GM_ACOS));
    uvvar++;
    for (int i = 0; i < nuv; i++) g_v.push_back(src_uv(i, GM_ACOS));
    uvvar++;
  }
  for (int i = 0; i < nsc; i++) g_sc.push_back(src_sc(i, 0, GM_PLAIN));
  if (a_scd) for (int i = 0; i < nsc; i++) g_ns.push_back(src_sc(i, 1, GM_ACOS));
  if (a_uvd) {
    for (int i = 0; i < nuv; i++) g_uew.push_back(src_uv(i, GM_EWDER_UV));
    uvvar++;
    for (int i = 0; i < nuv; i++) g_vew.push_back(src_uv(i, GM_EWDER_UV));
    uvvar++;
  }
  if (a_scd) for (int i = 0; i < nsc; i++) g_scew.push_back(src_sc(i, 2, GM_EWDER));
  std::vector<int> i_vor(nuv, -1), i_div(nuv, -1), i_uew(nuv, -1), i_vew(nuv, -1), i_ns(nsc, -1), i_scew(nsc, -1);
  for (auto *v : {&g_u, &g_v, &g_sc}) gin.insert(gin.end(), v->begin(), v->end());
  auto add = [&](const std::vector<GridFld> &v, std::vector<int> &idx) {
    for (size_t i = 0; i < v.size(); i++) {
      idx[i] = (int)gin.size();
      gin.push_back(v[i]);
    }
  };
  add(g_vor, i_vor), add(g_div, i_div), add(g_uew, i_uew), add(g_vew, i_vew), add(g_ns, i_ns), add(g_scew, i_scew);
  const int kf_total = (int)gin.size();
  // batches of whole atoms: {u_i, v_i [, their adjoint-only companions]} and {scalar_j [, its N-S and E-W inputs]}
  const int natoms = nuv + nsc;
  const int depth = pipeline_depth(P, kf_total);
  const int cap = pick_batch(P, kf_total, depth);
  std::vector<std::vector<int>> batches;  // Fourier field indices
  {
    std::vector<int> cur;
    for (int at = 0; at < natoms; at++) {
      std::vector<int> fl;
      if (at < nuv) {
        fl = {at, nuv + at};
        for (int x : {i_vor[at], i_div[at], i_uew[at], i_vew[at]})
          if (x >= 0) fl.push_back(x);
      } else {
        fl = {2 * nuv + (at - nuv)};
        for (int x : {i_ns[at - nuv], i_scew[at - nuv]})
          if (x >= 0) fl.push_back(x);
      }
      if (!cur.empty() && (int)(cur.size() + fl.size()) > cap) {
        batches.push_back(cur);
        cur.clear();
      }
      cur.insert(cur.end(), fl.begin(), fl.end());
    }
    if (!cur.empty()) batches.push_back(cur);
  }
  int maxb = 0;
  for (auto &b : batches) maxb = std::max(maxb, (int)b.size());
  const int nbat = (int)batches.size();
  const bool piped = depth > 1 && nbat > 1;
  const int bfpad = roundup(maxb, 64);
  if (ensure_work(P, bfpad, piped ? 2 : 1, st)) return EMI_ERR_RUNTIME;
  const int ldw_max = 2 * bfpad;  // row width of the widest batch; every batch has its own (2 x its fields rounded up to 64)
  struct Bat { size_t off_g, off_o, off_f; int ng, no, ldw; };
  const bool fuse_dir = !(test_paths() & 4);  // plain-copy fields leave k_leg_dir's epilogue straight for the caller's arrays
  std::vector<Bat> bats;
  std::vector<char> hdesc;
  for (auto &b : batches) {
    std::vector<GridFld> bg;
    std::vector<SpecDst> bo;
    std::vector<FuseDst> bf(bfpad, FuseDst{nullptr, 0, 0});  // per W field of the batch
    std::map<int, int> loc;
    for (size_t i = 0; i < b.size(); i++) {
      loc[b[i]] = (int)i;
      bg.push_back(gin[b[i]]);
    }
    auto at_ = [&](int x) { return x >= 0 ? loc[x] : -1; };  // field number -> position in this batch's W
    for (size_t i = 0; i < b.size(); i++) {
      int f = b[i];
      if (f < nuv) {  // u_i -> vor_i and div_i outputs
        SpecDst v{};
        v.dst = d_vor; v.stride = a.nf_uv; v.idx = f; v.kind = adj ? SPO_VOR_AD : SPO_VOR; v.src0 = loc[f]; v.src1 = loc[nuv + f];
        v.src2 = at_(i_uew[f]); v.src3 = at_(i_vew[f]); v.src4 = at_(i_vor[f]);
        bo.push_back(v);
        v.dst = d_div; v.kind = adj ? SPO_DIV_AD : SPO_DIV; v.src4 = at_(i_div[f]);
        bo.push_back(v);
      } else if (f >= 2 * nuv && f < 2 * nuv + nsc) {
        const int isc = f - 2 * nuv;
        const ScalarRef &r = sc[isc];
        SpecDst sd{};
        sd.kind = SPO_COPY; sd.src0 = (int)i; sd.src1 = sd.src2 = sd.src3 = sd.src4 = -1;
        switch (r.arr) {
          case 0: sd.dst = d_sc[0]; sd.stride = a.nf_scalar; sd.idx = r.lev; break;
          case 1: sd.dst = d_sc[1]; sd.stride = a.nf_sc2; sd.idx = r.lev; break;
          case 2: sd.dst = (char *)d_sc[2] + (size_t)r.var * ns2 * a.sc3a_nlev * P.esz; sd.stride = a.sc3a_nlev; sd.idx = r.lev; break;
          default: sd.dst = (char *)d_sc[3] + (size_t)r.var * ns2 * a.sc3b_nlev * P.esz; sd.stride = a.sc3b_nlev; sd.idx = r.lev; break;
        }
        if (a_scd) {  // value + adjoint of SPNSDE on the N-S input + (-i m) x the E-W input: k_postpack_dir
          sd.kind = SPO_SC_AD; sd.src1 = at_(i_ns[isc]); sd.src2 = at_(i_scew[isc]);
          bo.push_back(sd);
        } else if (fuse_dir) {
          bf[i] = FuseDst{sd.dst, sd.stride, sd.idx};  // written by the epilogue of k_leg_dir
        } else {
          bo.push_back(sd);
        }
      }
    }
    Bat bt{};
    bt.ng = (int)bg.size();
    bt.ldw = 2 * roundup((int)b.size(), 64);
    bt.no = (int)bo.size();
    bt.off_g = hdesc.size();
    hdesc.resize(bt.off_g + (bg.size() * sizeof(GridFld) + 255) / 256 * 256);
    if (!bg.empty()) memcpy(hdesc.data() + bt.off_g, bg.data(), bg.size() * sizeof(GridFld));
    bt.off_o = hdesc.size();
    hdesc.resize(bt.off_o + (bo.size() * sizeof(SpecDst) + 255) / 256 * 256);
    if (!bo.empty()) memcpy(hdesc.data() + bt.off_o, bo.data(), bo.size() * sizeof(SpecDst));
    bt.off_f = hdesc.size();
    hdesc.resize(bt.off_f + (bf.size() * sizeof(FuseDst) + 255) / 256 * 256);
    if (!bf.empty()) memcpy(hdesc.data() + bt.off_f, bf.data(), bf.size() * sizeof(FuseDst));
    bats.push_back(bt);
  }
  std::vector<LegMaps *> bmaps(nbat, nullptr);  // per batch: only the column tiles that hold fields (as INV_TRANS)
  for (int ib = 0; ib < nbat; ib++)
    if (leg_tilemaps(P, (bats[ib].ng + 63) / 64, &bmaps[ib])) return EMI_ERR_RUNTIME;
  if (upload_desc(P, hdesc, st)) return EMI_ERR_RUNTIME;
  emi_stream_t sA = st, sB = st, sX = st;
  if (piped) {
    if (g_pipe.init()) return EMI_ERR_RUNTIME;
    sA = (emi_stream_t)g_pipe.sA;
    sB = (emi_stream_t)g_pipe.sB;
    sX = (emi_stream_t)g_pipe.sX;
    g_pipe.begin(st);
  }
  g_pt.begin(G.profile != 0, G.profile == 2);
  // events of batch ib: 3 ib = FFT done, 3 ib + 1 = Legendre done, 3 ib + 2 = exchange done (as INV_TRANS)
  const bool dist = P.nproc > 1;
  // buffer strides for the widest batch, + the zero row behind the Legendre-side rows of each buffer
  const long long lrows_call = dist ? P.lrows : P.frows;
  const size_t lstride = (size_t)(lrows_call + 1) * ldw_max * P.esz, fstride = (size_t)P.frows * ldw_max * P.esz;
  for (int ib = 0; ib < nbat; ib++) {
    const Bat &bt = bats[ib];
    const int ldw = bt.ldw;  // this batch's row width
    const GridFld *d_bg = (const GridFld *)((char *)P.d_desc + bt.off_g);
    const SpecDst *d_bo = (const SpecDst *)((char *)P.d_desc + bt.off_o);
    char *FBl = P.d_FBL + (piped ? (size_t)(ib & 1) * lstride : 0);
    char *FBf = dist ? P.d_FBF + (piped ? (size_t)(ib & 1) * fstride : 0) : FBl;
    // stream B: FFTs; FBf[ib&1] was last read by the Legendre transform (one task) or by the exchange
    // (several tasks) of batch ib-2
    if (piped && ib >= 2) g_pipe.wait(3 * (ib - 2) + (dist ? 2 : 1), sB);
    int iv;
    {
      EmiRange rg(EMI_LBL_FTDIR);  // GSTATS 106: TRGTOL + FTDIR + FOURIER_OUT
      iv = g_pt.start(2, sB);
      if (launch_fft(P, false, adj, d_bg, bt.ng, FBf, ldw, nproma, sB)) {
        if (piped) g_pipe.end(st);
        plan_end(P, st);
        return EMI_ERR_RUNTIME;
      }
      g_pt.stop(iv, sB);
    }
    if (piped) g_pipe.signal(3 * ib, sB);
    if (dist) {
      // stream X: TRLTOM; FBl[ib&1] was last read by the Legendre transform of batch ib-2
      if (piped) g_pipe.wait(3 * ib, sX);
      if (piped && ib >= 2) g_pipe.wait(3 * (ib - 2) + 1, sX);
      EmiRange rg(EMI_LBL_TRLTOM);  // GSTATS 153
      if (exchange(P, false, ldw, sX, FBl, FBf)) {
        if (piped) g_pipe.end(st);
        plan_end(P, st);
        return EMI_ERR_RUNTIME;
      }
      if (piped) g_pipe.signal(3 * ib + 2, sX);
    }
    // stream A: Legendre + spectral unpack
    if (piped) g_pipe.wait(3 * ib + (dist ? 2 : 0), sA);
    EmiRange rgl(EMI_LBL_LTDIR);  // GSTATS 103: PRFI2B + LEDIR + UVTVD + UPDSP
    iv = g_pt.start(1, sA);
    // the zero row of this batch: row `lrows_call` in the batch's own row width (the buffer held other data before)
    emi_dev_memset(FBl + (size_t)lrows_call * ldw * P.esz, 0, (size_t)ldw * P.esz, sA);
    const FuseDst *d_bf = fuse_dir ? (const FuseDst *)((char *)P.d_desc + bt.off_f) : nullptr;
    LegMaps *lmaps = bmaps[ib];
    if (lmaps->n_dir_wide > 0)  // (the double-precision tiles first: they are the longest of the call)
      EMI_LAUNCH(emi_f32::k_leg_dir_wide, lmaps->n_dir_wide, LG_THREADS, leg_dir_lds_bytes(P), sA, P.g, (const int2 *)lmaps->d_dir_wide, (const float *)FBl, (int)lrows_call, ldw, (float *)P.d_W, ldw,
                 d_bf);
    if (lmaps->n_dir > 0)
      EMI_LAUNCH_P(P.esz, k_leg_dir, lmaps->n_dir, LG_THREADS, leg_dir_lds_bytes(P), sA, P.g, (const int2 *)lmaps->d_dir, (const RT *)FBl, (int)lrows_call, ldw, (RT *)P.d_W, ldw, d_bf);
    g_pt.stop(iv, sA);
    if (piped) g_pipe.signal(3 * ib + 1, sA);
    iv = g_pt.start(0, sA);
    if (bt.no > 0) {  // vorticity / divergence from the wind fields left in W
      long long nblk = ((long long)P.wrows_total + 3) / 4 * ((bt.no + 63) / 64);  // block = 4 rows x 64 fields
      EMI_LAUNCH_P(P.esz, k_postpack_dir, nblk, 256, 0, sA, P.g, d_bo, bt.no, (const RT *)P.d_W, ldw, (long long)P.wrows_total);
    }
    g_pt.stop(iv, sA);
  }
  if (piped) g_pipe.end(st);
  if (plan_end(P, st)) return EMI_ERR_RUNTIME;
  if (host) hs.flush(st);
#ifndef EMI_CPU_EMU
  EMI_CHECK(hipGetLastError());
#endif
  return EMI_SUCCESS;
}

static int specnorm_sumsq(Plan &P, int mem_space, const void *spec, int nfld, double *sumsq) {
  if (resolve_space("SPECNORM", mem_space, {spec}, &mem_space)) return EMI_ERR_ARG;
  HostStage hs(P.esz);
  // SPECNORM has no stream argument: it runs on the null stream behind the last transform of this resolution
  // (which may have been queued on a non-blocking stream, e.g. the DIR_TRANS that produced `spec`)
  if (plan_begin(P, (emi_stream_t)0)) return EMI_ERR_RUNTIME;
  const void *d_sp = hs.in(spec, (size_t)P.nspec2 * nfld, mem_space == EMI_MEM_HOST, 0);
  if (hs.failed) EMI_FAIL(EMI_ERR_RUNTIME, "SPECNORM: cannot stage the host array through device memory");
  void *d_out = nullptr;
  if (emi_dev_malloc(&d_out, (size_t)nfld * 8)) return EMI_ERR_RUNTIME;
  EMI_LAUNCH_P(P.esz, k_specnorm, nfld, 256, 256 * 8, (emi_stream_t)0, P.g, (long long)P.nspec2, (const RT *)d_sp, nfld, (double *)d_out);
  emi_d2h(sumsq, d_out, (size_t)nfld * 8, 0);
  emi_stream_sync(0);
  emi_dev_free(d_out);
  return EMI_SUCCESS;
}

// V-sets: the fields of PSPEC are this V-set's (spnorm_ctl_mod.F90: KVSET); their wavenumbers are spread over the NPRTRW tasks that
// share the V-set.  The host collective spans all tasks, and the V-sets may hold different numbers of fields: field counts first,
// then the partial sums; byv[v][k] = sum over the W-sets of field k of V-set v (the same on every task).
static int specnorm_vsets(Plan &P, const double *partial, int nfld, std::vector<std::vector<double>> &byv) {
  if (!G.hc_gather) EMI_FAIL(EMI_ERR_STATE, "SPECNORM: several tasks and no host collectives (emi_set_host_collectives)");
  const int NA = G.nproc_all;
  std::vector<long long> one(NA, 8), d1(NA);
  for (int t = 0; t < NA; t++) d1[t] = 8LL * t;
  std::vector<long long> nf_all(NA, 0);
  long long mine = nfld;
  if (G.hc_gather(G.hc_user, &mine, 8, nf_all.data(), one.data(), d1.data(), NA)) EMI_FAIL(EMI_ERR_RUNTIME, "SPECNORM: all-gather-v failed");
  std::vector<long long> cnt(NA), dsp(NA);
  long long tot = 0;
  for (int t = 0; t < NA; t++) cnt[t] = nf_all[t] * 8, dsp[t] = tot, tot += cnt[t];
  std::vector<double> all((size_t)(tot / 8) + 1);
  double dummy = 0.0;
  if (G.hc_gather(G.hc_user, nfld ? (const void *)partial : (const void *)&dummy, cnt[G.myproc_all - 1], all.data(), cnt.data(), dsp.data(), NA))
    EMI_FAIL(EMI_ERR_RUNTIME, "SPECNORM: all-gather-v failed");
  byv.assign(P.nprv, {});
  for (int v = 0; v < P.nprv; v++) {
    const long long nfv = nf_all[v];  // task (W-set 1, V-set v)
    byv[v].assign((size_t)nfv, 0.0);
    for (int w = 0; w < P.nproc; w++) {
      const int t = w * P.nprv + v;
      if (nf_all[t] != nfv) EMI_FAIL(EMI_ERR_ARG, "SPECNORM: tasks %d and %d of V-set %d pass %lld and %lld fields", v + 1, t + 1, v + 1, nfv, nf_all[t]);
      for (long long i = 0; i < nfv; i++) byv[v][(size_t)i] += all[(size_t)(dsp[t] / 8 + i)];
    }
  }
  return 0;
}

extern "C" int emi_specnorm(int kresol, int mem_space, const void *spec, int nfld, double *norms) {
  Plan *Pp = get_plan(kresol);
  if (!Pp) EMI_FAIL(EMI_ERR_STATE, "SPECNORM: unknown resolution %d", kresol);
  // V-sets: a task whose V-set holds none of the fields still takes part in the collective (nfld = 0)
  if (nfld < 0 || (nfld > 0 && (!spec || !norms)) || (nfld == 0 && Pp->nprv == 1)) EMI_FAIL(EMI_ERR_ARG, "SPECNORM: bad arguments");
  if (Pp->nproc > 1 && Pp->nprv == 1 && !G.hc_gather)
    EMI_FAIL(EMI_ERR_STATE, "SPECNORM: several tasks and no host collectives (emi_set_host_collectives): use emi_specnorm_partial and sum over tasks");
  if (nfld > 0) {
    const int rc = specnorm_sumsq(*Pp, mem_space, spec, nfld, norms);
    if (rc) return rc;
  }
  if (Pp->nproc > 1 && Pp->nprv == 1) {  // spnormc_mod.F90:49-85 gathers the partial sums on the master; here every task gets the norms
    const int NP = Pp->nproc;
    std::vector<double> all((size_t)NP * nfld);
    std::vector<long long> cnt(NP, (long long)nfld * 8), dsp(NP);
    for (int t = 0; t < NP; t++) dsp[t] = (long long)t * nfld * 8;
    if (G.hc_gather(G.hc_user, norms, cnt[0], all.data(), cnt.data(), dsp.data(), NP)) EMI_FAIL(EMI_ERR_RUNTIME, "SPECNORM: all-gather-v failed");
    for (int i = 0; i < nfld; i++) {
      double s = 0.0;
      for (int t = 0; t < NP; t++) s += all[(size_t)t * nfld + i];  // task order: the same sum on every task
      norms[i] = s;
    }
  } else if (Pp->nprv > 1) {
    std::vector<std::vector<double>> byv;
    if (specnorm_vsets(*Pp, norms, nfld, byv)) return EMI_ERR_RUNTIME;
    for (int i = 0; i < nfld; i++) norms[i] = byv[Pp->mev][i];
  }
  for (int i = 0; i < nfld; i++) norms[i] = std::sqrt(norms[i]);
  return EMI_SUCCESS;
}

// SPECNORM with KVSET (specnorm.h:12, spnorm_ctl_mod.F90): PSPEC holds this V-set's fields, PNORM the norms of ALL nfld_g fields
// (on every task; the reference fills it on the master), kvset[f] the V-set of global field f
extern "C" int emi_specnorm_kvset(int kresol, int mem_space, const void *spec, int nfld, const int *kvset, int nfld_g, double *norms_g) {
  Plan *Pp = get_plan(kresol);
  if (!Pp) EMI_FAIL(EMI_ERR_STATE, "SPECNORM: unknown resolution %d", kresol);
  Plan &P = *Pp;
  if (nfld < 0 || nfld_g < 0 || (nfld_g > 0 && (!kvset || !norms_g)) || (nfld > 0 && !spec)) EMI_FAIL(EMI_ERR_ARG, "SPECNORM: bad arguments");
  int mine = 0;
  for (int f = 0; f < nfld_g; f++) {
    if (kvset[f] < 1 || kvset[f] > P.nprv) EMI_FAIL(EMI_ERR_ARG, "SPECNORM:KVSET CONTAINS VALUES OUTSIDE RANGE");
    mine += kvset[f] == P.mev + 1;
  }
  if (mine != nfld) EMI_FAIL(EMI_ERR_ARG, "SPECNORM: %d fields of KVSET belong to this V-set, PSPEC holds %d", mine, nfld);
  if (P.nprv == 1) return emi_specnorm(kresol, mem_space, spec, nfld, norms_g);
  std::vector<double> part((size_t)nfld + 1, 0.0);
  if (nfld > 0) {
    const int rc = specnorm_sumsq(P, mem_space, spec, nfld, part.data());
    if (rc) return rc;
  }
  std::vector<std::vector<double>> byv;
  if (specnorm_vsets(P, part.data(), nfld, byv)) return EMI_ERR_RUNTIME;
  std::vector<int> k(P.nprv, 0);
  for (int f = 0; f < nfld_g; f++) {
    const int v = kvset[f] - 1;
    if ((size_t)k[v] >= byv[v].size()) EMI_FAIL(EMI_ERR_ARG, "SPECNORM: V-set %d passed fewer fields than KVSET names", v + 1);
    norms_g[f] = std::sqrt(byv[v][(size_t)k[v]++]);
  }
  return EMI_SUCCESS;
}
// V-set decomposition (sump_trans0_mod.F90:49, pe2set_mod.F90:111-112): NPRTRW, NPRTRV, MYSETW, MYSETV
extern "C" int emi_inq_vsets(int *nprtrw, int *nprtrv, int *mysetw, int *mysetv) {
  if (!G.init) EMI_FAIL(EMI_ERR_STATE, "emi_inq_vsets: SETUP_TRANS0 has not been called");
  if (nprtrw) *nprtrw = G.nproc;
  if (nprtrv) *nprtrv = G.nprtrv;
  if (mysetw) *mysetw = G.myproc;
  if (mysetv) *mysetv = G.mysetv;
  return EMI_SUCCESS;
}

// this task's contribution (sum over its wavenumbers of the weighted squares, spnormd_mod.F90);
// the caller sums over tasks and takes the square root (spnormc_mod.F90 gathers to the master)
extern "C" int emi_specnorm_partial(int kresol, int mem_space, const void *spec, int nfld, double *sumsq) {
  Plan *Pp = get_plan(kresol);
  if (!Pp) EMI_FAIL(EMI_ERR_STATE, "SPECNORM: unknown resolution %d", kresol);
  if (!spec || nfld <= 0 || !sumsq) EMI_FAIL(EMI_ERR_ARG, "SPECNORM: bad arguments");
  return specnorm_sumsq(*Pp, mem_space, spec, nfld, sumsq);
}

// GPNORM_TRANS (cpu/external/gpnorm_trans.F90:11-96, cpu/internal/gpnorm_trans_ctl_mod.F90): area-weighted average, minimum and maximum of
// `kfields` grid fields of PGP(nproma, gp_nfld, ngpblks).  Per latitude the sum over the longitudes in double times RW / NLOEN on the
// task that holds the latitude (whole latitudes here, so the reference's TRGTOL is not needed); the per-latitude values of all tasks
// are gathered and summed in latitude order on every task (the reference: on task 1), so the average does not depend on the
// decomposition.  ave_only (LDAVE_ONLY): pmin / pmax come in as the caller's local extrema and are only reduced over the tasks.
// Host arrays are staged whole (all gp_nfld fields of PGP, although only the first kfields are reduced): a diagnostic, not a hot path.
extern "C" int emi_gpnorm(int kresol, int mem_space, const void *gp, int gp_nfld, int kfields, int kproma, double *ave, double *pmin, double *pmax,
                          int ave_only) {
  Plan *Pp = get_plan(kresol);
  if (!Pp) EMI_FAIL(EMI_ERR_STATE, "GPNORM_TRANS: unknown resolution %d", kresol);
  Plan &P = *Pp;
  if (kfields <= 0 || !gp || !ave || !pmin || !pmax) EMI_FAIL(EMI_ERR_ARG, "GPNORM_TRANS: bad arguments");
  if (gp_nfld < kfields) EMI_FAIL(EMI_ERR_ARG, "GPNORM_TRANS_CTL:SECOND DIMENSION OF PGP TOO SMALL (%d < %d)", gp_nfld, kfields);
  if (G.nproc_all > 1 && !G.hc_gather) EMI_FAIL(EMI_ERR_STATE, "GPNORM_TRANS: several tasks and no host collectives (emi_set_host_collectives)");
  if (resolve_space("GPNORM_TRANS", mem_space, {gp}, &mem_space)) return EMI_ERR_ARG;
  const int j0 = P.vfirst(P.me, P.mev), j1 = P.vlast(P.me, P.mev), nl = j1 - j0;
  std::vector<int> rowoff(nl + 1, 0);
  for (int j = 0; j < nl; j++) rowoff[j + 1] = rowoff[j] + P.nloen[j0 + j];
  const long long myp = rowoff[nl];
  const int nproma = kproma > 0 ? kproma : (int)std::max<long long>(myp, 1);
  const long long ngpblks = myp > 0 ? (myp - 1) / nproma + 1 : 0;
  std::vector<double> loc((size_t)3 * kfields * std::max(nl, 1), 0.0);
  int lerr = 0;
  if (nl > 0) {
    HostStage hs(P.esz);
    if (plan_begin(P, (emi_stream_t)0)) return EMI_ERR_RUNTIME;  // behind the transform that produced the fields
    const void *d_gp = hs.in(gp, (size_t)nproma * gp_nfld * ngpblks, mem_space == EMI_MEM_HOST, 0);
    int *d_off = nullptr;
    void *d_out = nullptr;
    // a local failure must not leave the other tasks waiting in the gather below: it travels as a status word of this task's block
    if (hs.failed || upload(rowoff, &d_off) || emi_dev_malloc(&d_out, loc.size() * 8)) {
      lerr = 1;
    } else {
      EMI_LAUNCH_P(P.esz, k_gpnorm, (long long)nl * kfields, 256, 3 * 256 * 8, (emi_stream_t)0, (const int *)d_off, nl, (const RT *)d_gp, gp_nfld, kfields, nproma,
                   (double *)d_out);
      if (emi_d2h(loc.data(), d_out, loc.size() * 8, 0) || emi_stream_sync(0)) lerr = 1;
    }
    emi_dev_free(d_off);
    emi_dev_free(d_out);
  }
  // this task's block: [field][latitude] RW / NLOEN x row sums, then the field minima and maxima, then the status word
  const size_t nn = (size_t)kfields * nl;
  std::vector<double> blk(nn + 2 * (size_t)kfields + 1);
  blk.back() = (double)lerr;
  for (int f = 0; f < kfields; f++) {
    double mn = ave_only ? pmin[f] : 0.0, mx = ave_only ? pmax[f] : 0.0;
    for (int j = 0; j < nl; j++) {
      const double rwj = P.esz == 4 ? (double)(float)P.rw[j0 + j] : P.rw[j0 + j];  // REAL(PW(IGL),JPRB), gpnorm_trans_ctl_mod.F90:206
      blk[(size_t)f * nl + j] = loc[(size_t)f * nl + j] * rwj / (double)P.nloen[j0 + j];
      if (!ave_only) {
        const double a = loc[nn + (size_t)f * nl + j], b = loc[2 * nn + (size_t)f * nl + j];
        mn = j == 0 ? a : std::min(mn, a), mx = j == 0 ? b : std::max(mx, b);
      }
    }
    blk[nn + f] = mn, blk[nn + kfields + f] = mx;
  }
  const int NA = G.nproc_all;
  std::vector<double> all;
  std::vector<long long> cnt(NA, 0), dsp(NA, 0);
  std::vector<int> nlat_of(NA, 0);
  if (NA > 1) {
    long long tot = 0;
    for (int w = 0; w < P.nproc; w++)
      for (int v = 0; v < P.nprv; v++) {
        const int t = w * P.nprv + v;
        nlat_of[t] = P.vlast(w, v) - P.vfirst(w, v);
        cnt[t] = 8LL * ((long long)kfields * nlat_of[t] + 2 * kfields + 1), dsp[t] = tot, tot += cnt[t];
      }
    all.resize((size_t)(tot / 8));
    if (G.hc_gather(G.hc_user, blk.data(), cnt[G.myproc_all - 1], all.data(), cnt.data(), dsp.data(), NA)) EMI_FAIL(EMI_ERR_RUNTIME, "GPNORM_TRANS: all-gather-v failed");
  } else {
    nlat_of[0] = nl, cnt[0] = 8LL * (long long)blk.size();
    all = blk;
  }
  for (int t = 0; t < NA; t++)
    if (all[(size_t)(dsp[t] / 8) + (size_t)kfields * nlat_of[t] + 2 * (size_t)kfields] != 0.0)
      EMI_FAIL(EMI_ERR_RUNTIME, "GPNORM_TRANS: task %d could not stage or reduce its fields (no device memory?)", t + 1);
  for (int f = 0; f < kfields; f++) {  // latitude order = task order (bands and sub-bands ascend with the task number)
    double a = 0.0, mn = 0.0, mx = 0.0;
    bool first = true;
    for (int t = 0; t < NA; t++) {
      const double *b = all.data() + dsp[t] / 8;
      const int nlt = nlat_of[t];
      for (int j = 0; j < nlt; j++) a += b[(size_t)f * nlt + j];
      if (nlt > 0 || ave_only) {
        const double tmn = b[(size_t)kfields * nlt + f], tmx = b[(size_t)kfields * nlt + kfields + f];
        mn = first ? tmn : std::min(mn, tmn), mx = first ? tmx : std::max(mx, tmx), first = false;
      }
    }
    ave[f] = a, pmin[f] = mn, pmax[f] = mx;
  }
  return EMI_SUCCESS;
}

// VORDIV_TO_UV (cpu/external/vordiv_to_uv.F90:11-178): spectral vorticity / divergence -> spectral U = u cos(theta), V = v cos(theta), for the
// wavenumbers of this task's W-set (suwavedi_mod.F90:118-137), n <= KSMAX.  Needs SETUP_TRANS0 only (the reference sets up and releases a
// spectral-only resolution inside the call); precision: 8 or 4 bytes per real of the four arrays PSP*(nfld, nspec2).
extern "C" int emi_vordiv_to_uv(int ksmax, int precision, int mem_space, const void *spvor, const void *spdiv, void *spu, void *spv, int nfld, int nspec2_ext) {
  if (!G.init) EMI_FAIL(EMI_ERR_STATE, "VORDIV_TO_UV: SETUP_TRANS0 has not been called");
  if (ksmax < 0 || (precision != 8 && precision != 4)) EMI_FAIL(EMI_ERR_ARG, "VORDIV_TO_UV: bad arguments (KSMAX = %d, precision = %d)", ksmax, precision);
  if (nfld <= 0) return EMI_SUCCESS;
  if (!spvor || !spdiv || !spu || !spv) EMI_FAIL(EMI_ERR_ARG, "VORDIV_TO_UV : PSPVOR / PSPDIV / PSPU / PSPV MISSING");
  const int N = ksmax, NP = G.nproc, me = G.myproc - 1;
  std::vector<int> mval, nasm0, ebase, pairm;
  std::vector<double> eps, lapin(N + 4, 0.0);
  {
    int ik = 0, ind = 1, ipos = 0;
    for (int m = 0; m <= N; m++) {  // the zig-zag of SETUP_TRANS (suwavedi_mod.F90:118-137)
      ik += ind;
      if (ik > NP) ik = NP, ind = -1;
      else if (ik < 1) ik = 1, ind = 1;
      if (ik - 1 != me) continue;
      const int ml = (int)mval.size();
      mval.push_back(m), nasm0.push_back(ipos), ebase.push_back((int)eps.size());
      for (int n = m; n <= N; n++) pairm.push_back(ml);
      ipos += 2 * (N - m + 1);
      for (int n = m; n <= N + 2; n++) eps.push_back(std::sqrt((double)(n * n - m * m) / (double)(4 * n * n - 1)));  // REPSNM (pre_suleg_mod.F90:55-63)
    }
  }
  for (int n = 1; n <= N + 2; n++) lapin[n + 1] = -(G.ra * G.ra / (double)(n * (n + 1)));  // RLAPIN (pre_suleg_mod.F90:64-69)
  const int nspec2 = 2 * (int)pairm.size();
  if (nspec2 == 0) return EMI_SUCCESS;
  if (nspec2_ext < nspec2)
    EMI_FAIL(EMI_ERR_ARG, "VORDIV_TO_UV:SECOND DIMENSION OF PSPVOR / PSPDIV / PSPU / PSPV TOO SMALL (%d < %d spectral coefficients at KSMAX = %d)", nspec2_ext, nspec2,
             ksmax);
  if (resolve_space("VORDIV_TO_UV", mem_space, {spvor, spdiv, spu, spv}, &mem_space)) return EMI_ERR_ARG;
  HostStage hs(precision);
  const bool host = mem_space == EMI_MEM_HOST;
  const void *d_vor = hs.in(spvor, (size_t)nspec2 * nfld, host, 0), *d_div = hs.in(spdiv, (size_t)nspec2 * nfld, host, 0);
  void *d_u = hs.out(spu, (size_t)nspec2 * nfld, host), *d_v = hs.out(spv, (size_t)nspec2 * nfld, host);
  int *d_pairm = nullptr, *d_mval = nullptr, *d_nasm0 = nullptr, *d_ebase = nullptr;
  double *d_eps = nullptr, *d_lapin = nullptr;
  int rc = EMI_SUCCESS;
  if (hs.failed || upload(pairm, &d_pairm) || upload(mval, &d_mval) || upload(nasm0, &d_nasm0) || upload(ebase, &d_ebase) || upload(eps, &d_eps) ||
      upload(lapin, &d_lapin)) {
    emi_set_error("VORDIV_TO_UV: no device memory");
    rc = EMI_ERR_RUNTIME;
  } else {
    Vd2uvDev d{d_pairm, d_mval, d_nasm0, d_ebase, d_eps, d_lapin, N, nspec2, nfld, 1.0 / G.ra};
    const long long nthr = (long long)(nspec2 / 2) * nfld;
    EMI_LAUNCH_P(precision, k_vd2uv, (nthr + 255) / 256, 256, 0, (emi_stream_t)0, d, (const RT *)d_vor, (const RT *)d_div, (RT *)d_u, (RT *)d_v);
    hs.flush(0);
    emi_stream_sync(0);
  }
  for (void *q : {(void *)d_pairm, (void *)d_mval, (void *)d_nasm0, (void *)d_ebase, (void *)d_eps, (void *)d_lapin}) emi_dev_free(q);
  return rc;
}

extern "C" int emi_work_model(int kresol, int nfields, double *leg, double *fft, double *fbytes) {
  Plan *Pp = get_plan(kresol);
  if (!Pp) EMI_FAIL(EMI_ERR_STATE, "emi_work_model: unknown resolution %d", kresol);
  Plan &P = *Pp;
  // LT flops/direction = KF * sum_m 2*NDGLU(m)*(N-m+2)*c_m, c_0=1, c_{m>0}=2 (this
  // task's wavenumbers); FFT ~ 2.5 n log2 n per row; Fourier bytes of this task's latitudes
  double s = 0.0;
  for (int ml = 0; ml < P.nump; ml++) {
    const int m = P.mval[ml];
    s += 2.0 * std::min(P.ndgnh, P.ndglu[m]) * (double)(P.nsmax - m + 2) * (m == 0 ? 1.0 : 2.0);
  }
  if (leg) *leg = s * nfields;
  double f = 0.0;
  for (int j = P.lat0; j < P.lat0 + P.nlat; j++) f += 2.5 * P.nloen[j] * std::log2((double)std::max(2, P.nloen[j]));
  if (fft) *fft = f * nfields;
  if (fbytes) *fbytes = (double)P.frows * 2.0 * P.esz * nfields;
  return EMI_SUCCESS;
}

extern "C" int emi_last_phase_ms(double *ms3) {
  int l[4];
  double ms[4];
  g_pt.resolve(ms, l);
  for (int i = 0; i < 3; i++) ms3[i] = ms[i];
  return EMI_SUCCESS;
}

extern "C" int emi_last_phase_launches(int *l3) {
  double ms[4];
  int l[4];
  g_pt.resolve(ms, l);
  for (int i = 0; i < 3; i++) l3[i] = l[i];
  return EMI_SUCCESS;
}

// TRMTOL / TRLTOM (and, with V-sets, TRLTOG / TRGTOL) of the calls since emi_set_profile: device time between an event in front of
// and one behind the all-to-all-v hook on the stream the hook was given, the number of exchanges, the bytes this task sent.
extern "C" int emi_last_exchange(double *ms, int *calls, double *bytes_sent) {
  double m4[4];
  int l[4];
  g_pt.resolve(m4, l);
  if (ms) *ms = m4[3];
  if (calls) *calls = l[3];
  if (bytes_sent) *bytes_sent = g_pt.xbytes;
  return EMI_SUCCESS;
}

// kernel launches of the FFT phases since emi_set_profile (one FTINV or FTDIR of one field batch launches one kernel per length class)
extern "C" int emi_last_fft_launches(long long *kernels) {
  if (kernels) *kernels = g_pt.fft_kernels;
  return EMI_SUCCESS;
}

extern "C" int emi_set_profile(int on) {
  G.profile = on;
  g_pt.n = 0;  // a new measurement starts here
  g_pt.xbytes = 0;
  g_pt.fft_kernels = 0;
  return EMI_SUCCESS;
}

extern "C" int emi_set_max_batch(int max_fields) {
  G.max_batch = max_fields;
  return EMI_SUCCESS;
}

// ------------------------------------------------------------------------------------------
// DIST_SPEC / GATH_SPEC / DIST_GRID / GATH_GRID: host-side re-layouts + two host collectives
// ------------------------------------------------------------------------------------------
extern "C" int emi_set_host_collectives(emi_bcast_fn bcast, emi_allgatherv_fn allgatherv, void *user) {
  G.hc_bcast = bcast;
  G.hc_gather = allgatherv;
  G.hc_user = user;
  return EMI_SUCCESS;
}
// broadcast of host bytes from task `root` (1-based) over the attached transport: what the drop-in layers above need where the
// reference sends a small array from its master task (GPNORM_TRANSAD: gpnorm_trans_ctlad_mod.F90)
extern "C" int emi_bcast_host(void *buf, long long bytes, int root) {
  if (!G.init) EMI_FAIL(EMI_ERR_STATE, "emi_bcast_host: SETUP_TRANS0 has not been called");
  if (root < 1 || root > G.nproc_all || bytes < 0 || (bytes > 0 && !buf)) EMI_FAIL(EMI_ERR_ARG, "emi_bcast_host: bad arguments");
  if (G.nproc_all == 1 || bytes == 0) return EMI_SUCCESS;
  if (!G.hc_bcast) EMI_FAIL(EMI_ERR_STATE, "emi_bcast_host: several tasks and no host collectives (emi_set_host_collectives)");
  if (G.hc_bcast(G.hc_user, buf, bytes, root - 1)) EMI_FAIL(EMI_ERR_RUNTIME, "emi_bcast_host: broadcast failed");
  return EMI_SUCCESS;
}
extern "C" int emi_inq_init(int *kmax_resol, double *prad) {
  if (!G.init) EMI_FAIL(EMI_ERR_STATE, "emi_inq_init: SETUP_TRANS0 has not been called");
  if (kmax_resol) *kmax_resol = G.max_resol;
  if (prad) *prad = G.ra;
  return EMI_SUCCESS;
}
extern "C" int emi_inq_tasks(int *nproc, int *myproc) {
  if (!G.init) EMI_FAIL(EMI_ERR_STATE, "emi_inq_tasks: SETUP_TRANS0 has not been called");
  if (nproc) *nproc = G.nproc_all;
  if (myproc) *myproc = G.myproc_all;
  return EMI_SUCCESS;
}
namespace {
struct TaskLayout {  // global <-> per-task positions of one resolution
  std::vector<long long> iasm0g;              // [N+1] global start of wavenumber m
  std::vector<std::vector<int>> ms;           // [task] its wavenumbers, ascending
  std::vector<std::vector<long long>> start;  // [task][i] local start of ms[task][i]
  std::vector<long long> nspec2, gp0, ngp;    // [task]
};
TaskLayout task_layout(const Plan &P) {
  TaskLayout L;
  const int N = P.nsmax, NP = P.nproc;
  L.iasm0g.assign(N + 2, 0);
  for (int m = 0; m <= N; m++) L.iasm0g[m + 1] = L.iasm0g[m] + 2LL * (N - m + 1);
  L.ms.assign(NP, {});
  L.start.assign(NP, {});
  L.nspec2.assign(NP, 0);
  for (int m = 0; m <= N; m++) {
    const int t = P.procm[m];
    L.ms[t].push_back(m);
    L.start[t].push_back(L.nspec2[t]);
    L.nspec2[t] += 2LL * (N - m + 1);
  }
  std::vector<long long> cum(P.ndgl + 1, 0);
  for (int j = 0; j < P.ndgl; j++) cum[j + 1] = cum[j] + P.nloen[j];
  L.gp0.assign(NP, 0);
  L.ngp.assign(NP, 0);
  for (int t = 0; t < NP; t++) {
    L.gp0[t] = cum[P.latlo[t]];
    L.ngp[t] = cum[P.latlo[t + 1]] - cum[P.latlo[t]];
  }
  return L;
}
int check_tasks(const Plan &P, const int *k, int nfld, const char *who) {
  // V-sets: a spectral field lives on the NPRTRW tasks of ONE V-set (KVSET of dist_spec.h / gath_spec.h), a grid field on the
  // sub-bands of all NPROC tasks; these re-layout helpers know the W-set decomposition only.  Hosts with NPRTRV > 1 move their
  // global fields themselves (TRANS_INQ gives every task's share: "latlo", "myms", "nasm0", "mysetv").
  if (P.nprv > 1) EMI_FAIL(EMI_ERR_UNSUPPORTED, "%s: not available with NPRTRV > 1", who);
  if (nfld < 0 || (nfld > 0 && !k)) EMI_FAIL(EMI_ERR_ARG, "%s: task list missing", who);
  for (int f = 0; f < nfld; f++)
    if (k[f] < 1 || k[f] > P.nproc) EMI_FAIL(EMI_ERR_ARG, "%s: task %d of field %d outside 1..%d", who, k[f], f + 1, P.nproc);
  if (P.nproc > 1 && (!G.hc_bcast || !G.hc_gather))
    EMI_FAIL(EMI_ERR_STATE, "%s: %d tasks but no host collectives registered (emi_set_host_collectives)", who, P.nproc);
  return 0;
}
int slot_of(const int *ksort, int f, int nfld, const char *who, int *slot) {
  *slot = ksort ? ksort[f] - 1 : f;
  if (*slot < 0 || *slot >= nfld) EMI_FAIL(EMI_ERR_ARG, "%s: KSORT(%d) = %d outside 1..%d", who, f + 1, *slot + 1, nfld);
  return 0;
}
}  // namespace

// global fields on their source tasks -> every task's share.  One broadcast per source task.
template <bool SPEC>
static int dist_impl(int kresol, const void *glob, int nfld, const int *kfrom, const int *ksort, int kproma, void *loc, const char *who) {
  Plan *Pp = get_plan(kresol);
  if (!Pp) EMI_FAIL(EMI_ERR_STATE, "%s: unknown resolution %d", who, kresol);
  Plan &P = *Pp;
  if (check_tasks(P, kfrom, nfld, who)) return EMI_ERR_ARG;
  if (!loc) EMI_FAIL(EMI_ERR_ARG, "%s: local array missing", who);
  const TaskLayout L = task_layout(P);
  const size_t esz = P.esz;
  const long long nglob = SPEC ? (long long)P.nspec2g : (long long)P.ngptotg;
  const int nproma = kproma > 0 ? kproma : P.ngptot;
  std::vector<char> tmp;
  long long mine_done = 0;  // global fields of this task consumed so far
  for (int root = 0; root < P.nproc; root++) {
    std::vector<int> fl;
    for (int f = 0; f < nfld; f++)
      if (kfrom[f] == root + 1) fl.push_back(f);
    if (fl.empty()) continue;
    const long long bytes = (long long)fl.size() * nglob * (long long)esz;
    const char *buf;
    if (root == P.me) {
      if (!glob) EMI_FAIL(EMI_ERR_ARG, "%s: this task is the source of %zu fields but passes no global array", who, fl.size());
      buf = (const char *)glob + (size_t)mine_done * nglob * esz;
      mine_done += (long long)fl.size();
      if (P.nproc > 1 && G.hc_bcast(G.hc_user, (void *)buf, bytes, root)) EMI_FAIL(EMI_ERR_RUNTIME, "%s: broadcast failed", who);
    } else {
      tmp.resize((size_t)bytes);
      if (G.hc_bcast(G.hc_user, tmp.data(), bytes, root)) EMI_FAIL(EMI_ERR_RUNTIME, "%s: broadcast failed", who);
      buf = tmp.data();
    }
    for (size_t i = 0; i < fl.size(); i++) {
      int slot;
      if (slot_of(ksort, fl[i], nfld, who, &slot)) return EMI_ERR_ARG;
      const char *g = buf + i * (size_t)nglob * esz;
      if (SPEC) {  // PSPEC(nfld, nspec2): element (slot, isp)
        const std::vector<int> &ms = L.ms[P.me];
        for (size_t k = 0; k < ms.size(); k++) {
          const long long cnt = 2LL * (P.nsmax - ms[k] + 1), gs = L.iasm0g[ms[k]], ls = L.start[P.me][k];
          for (long long e = 0; e < cnt; e++) memcpy((char *)loc + ((size_t)(ls + e) * nfld + slot) * esz, g + (size_t)(gs + e) * esz, esz);
        }
      } else {  // PGP(nproma, nfld, ngpblks): point p -> block p / nproma
        const long long g0 = L.gp0[P.me], ng = L.ngp[P.me];
        for (long long p0 = 0; p0 < ng; p0 += nproma) {
          const long long w = std::min<long long>(nproma, ng - p0), blk = p0 / nproma;
          memcpy((char *)loc + ((size_t)(blk * nfld + slot) * nproma) * esz, g + (size_t)(g0 + p0) * esz, (size_t)w * esz);
        }
      }
    }
  }
  return EMI_SUCCESS;
}

// every task's share -> global fields on their target tasks.  An all-gather-v of the packed local fields per chunk of fields.
template <bool SPEC>
static int gath_impl(int kresol, void *glob, int nfld, const int *kto, int kproma, const void *loc, const char *who) {
  Plan *Pp = get_plan(kresol);
  if (!Pp) EMI_FAIL(EMI_ERR_STATE, "%s: unknown resolution %d", who, kresol);
  Plan &P = *Pp;
  if (check_tasks(P, kto, nfld, who)) return EMI_ERR_ARG;
  if (!loc) EMI_FAIL(EMI_ERR_ARG, "%s: local array missing", who);
  const TaskLayout L = task_layout(P);
  const size_t esz = P.esz;
  const int NP = P.nproc, nproma = kproma > 0 ? kproma : P.ngptot;
  const long long nglob = SPEC ? (long long)P.nspec2g : (long long)P.ngptotg;
  auto nloc = [&](int t) { return SPEC ? L.nspec2[t] : L.ngp[t]; };
  // Fields in chunks whose gathered image stays below 256 MiB: every task receives every field of a chunk whatever KTO says (the
  // host collectives offer an all-gather-v only), so an unchunked GATH_GRID of 137 levels at TCo1279 would put 7 GB on every task.
  const long long per_field = nglob * (long long)esz;
  const long long budget = (getenv("EMI_GATH_CHUNK") && *getenv("EMI_GATH_CHUNK")) ? atoll(getenv("EMI_GATH_CHUNK")) : (256LL << 20);  // bytes (tests: a few fields per chunk)
  const int chunk = (int)std::max<long long>(1, std::min<long long>(nfld, budget / std::max<long long>(per_field, 1)));
  long long i_mine = 0;
  std::vector<char> mine, all;
  std::vector<long long> cnt(NP), dsp(NP);
  for (int fa = 0; fa < nfld; fa += chunk) {
    const int nf = std::min(chunk, nfld - fa);
    // pack [field][local element]
    mine.resize((size_t)nf * nloc(P.me) * esz);
    for (int fc = 0; fc < nf; fc++) {
      const int f = fa + fc;
      char *dst = mine.data() + (size_t)fc * nloc(P.me) * esz;
      if (SPEC) {
        for (long long e = 0; e < nloc(P.me); e++) memcpy(dst + (size_t)e * esz, (const char *)loc + ((size_t)e * nfld + f) * esz, esz);
      } else {
        for (long long p0 = 0; p0 < nloc(P.me); p0 += nproma) {
          const long long w = std::min<long long>(nproma, nloc(P.me) - p0), blk = p0 / nproma;
          memcpy(dst + (size_t)p0 * esz, (const char *)loc + ((size_t)(blk * nfld + f) * nproma) * esz, (size_t)w * esz);
        }
      }
    }
    const char *base = mine.data();
    if (NP > 1) {
      long long tot = 0;
      for (int t = 0; t < NP; t++) {
        cnt[t] = (long long)nf * nloc(t) * (long long)esz;
        dsp[t] = tot;
        tot += cnt[t];
      }
      all.resize((size_t)tot);
      if (G.hc_gather(G.hc_user, mine.data(), cnt[P.me], all.data(), cnt.data(), dsp.data(), NP)) EMI_FAIL(EMI_ERR_RUNTIME, "%s: all-gather-v failed", who);
      base = all.data();
    } else {
      dsp[0] = 0;
    }
    for (int fc = 0; fc < nf; fc++) {
      const int f = fa + fc;
      if (kto[f] != P.me + 1) continue;
      if (!glob) EMI_FAIL(EMI_ERR_ARG, "%s: this task is the target of field %d but passes no global array", who, f + 1);
      char *g = (char *)glob + (size_t)i_mine * nglob * esz;
      i_mine++;
      for (int t = 0; t < NP; t++) {
        const char *src = base + dsp[t] + (size_t)fc * nloc(t) * esz;
        if (SPEC) {
          for (size_t k = 0; k < L.ms[t].size(); k++) {
            const long long cnte = 2LL * (P.nsmax - L.ms[t][k] + 1);
            memcpy(g + (size_t)L.iasm0g[L.ms[t][k]] * esz, src + (size_t)L.start[t][k] * esz, (size_t)cnte * esz);
          }
        } else {
          memcpy(g + (size_t)L.gp0[t] * esz, src, (size_t)L.ngp[t] * esz);
        }
      }
    }
  }
  return EMI_SUCCESS;
}

extern "C" int emi_dist_spec(int kresol, const void *specg, int nfld, const int *kfrom, const int *ksort, void *spec) {
  return dist_impl<true>(kresol, specg, nfld, kfrom, ksort, 0, spec, "DIST_SPEC");
}
extern "C" int emi_gath_spec(int kresol, void *specg, int nfld, const int *kto, const void *spec) {
  return gath_impl<true>(kresol, specg, nfld, kto, 0, spec, "GATH_SPEC");
}
extern "C" int emi_dist_grid(int kresol, const void *gpg, int nfld, const int *kfrom, const int *ksort, int kproma, void *gp) {
  return dist_impl<false>(kresol, gpg, nfld, kfrom, ksort, kproma, gp, "DIST_GRID");
}
extern "C" int emi_gath_grid(int kresol, void *gpg, int nfld, const int *kto, int kproma, const void *gp) {
  return gath_impl<false>(kresol, gpg, nfld, kto, kproma, gp, "GATH_GRID");
}

// CRC-64/ECMA-182, table driven (ectrans-benchmark.F90:1455-1600 calls fiat's crc64, un-vendored)
extern "C" int emi_crc64(const void *data, size_t bytes, unsigned long long *crc) {
  if ((!data && bytes) || !crc) EMI_FAIL(EMI_ERR_ARG, "emi_crc64: null argument");
  static unsigned long long tab[256];
  static bool init = false;
  if (!init) {
    for (int i = 0; i < 256; i++) {
      unsigned long long c = (unsigned long long)i << 56;
      for (int k = 0; k < 8; k++) c = (c & 0x8000000000000000ULL) ? (c << 1) ^ 0x42F0E1EBA9EA3693ULL : (c << 1);
      tab[i] = c;
    }
    init = true;
  }
  unsigned long long c = *crc;
  const unsigned char *p = (const unsigned char *)data;
  for (size_t i = 0; i < bytes; i++) c = tab[((c >> 56) ^ p[i]) & 0xff] ^ (c << 8);
  *crc = c;
  return EMI_SUCCESS;
}

// ------------------------------------------------------------------------------------------
// V-sets (NPRTRV > 1): INV_TRANS / DIR_TRANS of task (MYSETW, MYSETV).
// Spectral and Fourier space hold the fields of ONE V-set on the wavenumbers / latitude band of the W-set; grid space holds ALL
// fields on the task's sub-band (inv_trans.F90:212-300, trltog_mod.F90, trgtol_mod.F90).  The transform of the local fields is
// the W-set transform above, run into / out of a band-sized device array; TRLTOG / TRGTOL between the NPRTRV tasks of the band
// is one all-to-all-v of dense [field][point] blocks, packed and unpacked by k_gridcopy.  Field order everywhere: the
// reference's ([vor] [div] u v scalars [N-S derivatives] [u, v E-W derivatives] [scalar E-W derivatives]); a V-set's fields
// are the global ones it owns, in global order, group by group.
// ------------------------------------------------------------------------------------------
struct StageBuf {  // device scratch from the staging pool
  void *p = nullptr;
  explicit StageBuf(size_t bytes) { p = bytes ? emi_stage::acquire(bytes) : nullptr; }
  ~StageBuf() {
    if (p) emi_stage::release(p);
  }
};
struct VGroups {
  int nuv_g = 0;
  std::vector<int> ouv;            // owner V-set (0-based) of every global u/v field
  std::vector<ScalarRef> sc_g;     // global scalars in the reference's order
  std::vector<int> osc;            // their owners
  int nsc_g[4] = {0, 0, 0, 0};     // global counts: PSPSCALAR fields, PSPSC2 fields, PSPSC3A levels, PSPSC3B levels
  int nvar3a = 0, nvar3b = 0;      // variables of PSPSC3A / PSPSC3B = IF_SC3A_G3 / IF_SC3B_G3 (inv_trans.F90:277, 310): the same on every task
};
template <class ARGS>
static int v_groups(const Plan &P, const ARGS &a, const char *who, VGroups &vg) {
  const emi_vsets_t *vs = a.vsets;
  if (!vs) EMI_FAIL(EMI_ERR_ARG, "%s: NPRTRV = %d but no KVSET arrays (emi_vsets_t)", who, P.nprv);
  auto owners = [&](const int *k, int n, const char *nm, std::vector<int> &out) {
    out.clear();
    for (int i = 0; i < n; i++) {
      if (!k || k[i] < 1 || k[i] > P.nprv) {
        emi_set_error("%s:%s TOO LONG OR CONTAINS VALUES OUTSIDE RANGE", who, nm);
        return -1;
      }
      out.push_back(k[i] - 1);
    }
    return 0;
  };
  vg.nuv_g = vs->kvsetuv ? vs->nuv_g : 0;
  if (owners(vs->kvsetuv, vg.nuv_g, "KVSETUV", vg.ouv)) return EMI_ERR_ARG;
  const int mine_uv = (int)std::count(vg.ouv.begin(), vg.ouv.end(), P.mev);
  if (mine_uv != ((a.spvor || a.spdiv) ? a.nf_uv : 0))
    EMI_FAIL(EMI_ERR_ARG, "%s: %d fields of KVSETUV belong to this V-set, the spectral arrays hold %d", who, mine_uv, (a.spvor || a.spdiv) ? a.nf_uv : 0);
  std::vector<int> o;
  vg.sc_g.clear(), vg.osc.clear();
  auto check_local = [&](const char *nm, int mine, int have) {
    if (mine != have) {
      emi_set_error("%s: %d fields of %s belong to this V-set, the spectral array holds %d", who, mine, nm, have);
      return -1;
    }
    return 0;
  };
  if (vs->kvsetsc) {  // PSPSCALAR form
    if (vs->kvsetsc2 || vs->kvsetsc3a || vs->kvsetsc3b) EMI_FAIL(EMI_ERR_ARG, "%s : PSPSCALAR AND PSPSC3A/PSPSC3B/PSPSC2 BOTH PRESENT", who);
    if (owners(vs->kvsetsc, vs->nsc_g, "KVSETSC", o)) return EMI_ERR_ARG;
    vg.nsc_g[0] = vs->nsc_g;
    for (int i = 0; i < vs->nsc_g; i++) vg.sc_g.push_back({0, i, 0}), vg.osc.push_back(o[i]);
    if (check_local("KVSETSC", (int)std::count(o.begin(), o.end(), P.mev), a.spscalar ? a.nf_scalar : 0)) return EMI_ERR_ARG;
  } else {
    if (vs->kvsetsc2) {
      if (owners(vs->kvsetsc2, vs->nsc2_g, "KVSETSC2", o)) return EMI_ERR_ARG;
      vg.nsc_g[1] = vs->nsc2_g;
      for (int i = 0; i < vs->nsc2_g; i++) vg.sc_g.push_back({1, i, 0}), vg.osc.push_back(o[i]);
      if (check_local("KVSETSC2", (int)std::count(o.begin(), o.end(), P.mev), a.spsc2 ? a.nf_sc2 : 0)) return EMI_ERR_ARG;
    }
    if (vs->kvsetsc3a) {
      if (owners(vs->kvsetsc3a, vs->nsc3a_g, "KVSETSC3A", o)) return EMI_ERR_ARG;
      vg.nsc_g[2] = vs->nsc3a_g;
      // The reference takes the variable count from UBOUND(PSPSC3A,3), which a task whose V-set owns no level still passes as a
      // zero-level array (inv_trans.F90:272-277 aborts without it).  Here such a task may have no spectral array at all, so the
      // count also travels in the KVSET block (from the grid array); without either the peers would disagree on the field list.
      // the count is UBOUND(PSPSC3A,3) wherever the task names one (also with zero levels), and the third extent of PGP3A (/ 3 with
      // LDSCDERS) must EQUAL it: inv_trans.F90:557, :587 and dir_trans.F90:451, :481 abort on `IUBOUND(3) /= IF_SC3A_G3`.  A grid array with
      // spare room would also let a task that names no PSPSC3A (count taken from the grid array) list other global fields than its peers.
      vg.nvar3a = a.sc3a_nvar > 0 ? a.sc3a_nvar : vs->nvar3a_g;
      if (vg.nvar3a <= 0) EMI_FAIL(EMI_ERR_ARG, "%s:KVSETSC3A BUT NOT PSPSC3A (number of variables unknown: pass sc3a_nvar or emi_vsets_t.nvar3a_g)", who);
      if (vs->nvar3a_g > 0 && vs->nvar3a_g != vg.nvar3a)
        EMI_FAIL(EMI_ERR_ARG, "%s:THIRD DIMENSION OF PGP3A INCONSISTENT (%d variables, IF_SC3A_G3 = %d)", who, vs->nvar3a_g, vg.nvar3a);
      for (int v = 0; v < vg.nvar3a; v++)
        for (int l = 0; l < vs->nsc3a_g; l++) vg.sc_g.push_back({2, l, v}), vg.osc.push_back(o[l]);
      if (check_local("KVSETSC3A", (int)std::count(o.begin(), o.end(), P.mev), a.spsc3a ? a.sc3a_nlev : 0)) return EMI_ERR_ARG;
    }
    if (vs->kvsetsc3b) {
      if (owners(vs->kvsetsc3b, vs->nsc3b_g, "KVSETSC3B", o)) return EMI_ERR_ARG;
      vg.nsc_g[3] = vs->nsc3b_g;
      // the count is UBOUND(PSPSC3B,3) wherever the task names one (also with zero levels), and the third extent of PGP3B (/ 3 with
      // LDSCDERS) must EQUAL it: inv_trans.F90:557, :587 and dir_trans.F90:451, :481 abort on `IUBOUND(3) /= IF_SC3B_G3`.  A grid array with
      // spare room would also let a task that names no PSPSC3B (count taken from the grid array) list other global fields than its peers.
      vg.nvar3b = a.sc3b_nvar > 0 ? a.sc3b_nvar : vs->nvar3b_g;
      if (vg.nvar3b <= 0) EMI_FAIL(EMI_ERR_ARG, "%s:KVSETSC3B BUT NOT PSPSC3B (number of variables unknown: pass sc3b_nvar or emi_vsets_t.nvar3b_g)", who);
      if (vs->nvar3b_g > 0 && vs->nvar3b_g != vg.nvar3b)
        EMI_FAIL(EMI_ERR_ARG, "%s:THIRD DIMENSION OF PGP3B INCONSISTENT (%d variables, IF_SC3B_G3 = %d)", who, vs->nvar3b_g, vg.nvar3b);
      for (int v = 0; v < vg.nvar3b; v++)
        for (int l = 0; l < vs->nsc3b_g; l++) vg.sc_g.push_back({3, l, v}), vg.osc.push_back(o[l]);
      if (check_local("KVSETSC3B", (int)std::count(o.begin(), o.end(), P.mev), a.spsc3b ? a.sc3b_nlev : 0)) return EMI_ERR_ARG;
    }
  }
  return 0;
}
// all grid fields of the call in the reference's order, over the caller's (global-count) arrays, with their owner V-sets
struct VGridList {
  std::vector<GridFld> g;
  std::vector<int> owner;
};
static void v_grid_fields(const VGroups &vg, bool lvorgp, bool ldivgp, bool lscders, bool luvder, void *gp, int gp_nfld, void *gpuv, int uv_dim3,
                          void *gp2, void *gp3a, void *gp3b, VGridList &out) {
  const int nvar3a = vg.nvar3a, nvar3b = vg.nvar3b;
  const int nuv = vg.nuv_g, nsc = (int)vg.sc_g.size(), dmul = lscders ? 3 : 1;
  int gcount = 0, uvvar = 0;
  auto uvf = [&](int lev) {
    GridFld g{};
    if (gp) { g.base = gp; g.nf_arr = gp_nfld; g.fidx = gcount; }
    else { g.base = gpuv; g.nf_arr = nuv * uv_dim3; g.fidx = uvvar * nuv + lev; }
    out.g.push_back(g), out.owner.push_back(vg.ouv[lev]);
    gcount++;
  };
  auto scf = [&](int isc, int kder) {
    GridFld g{};
    const ScalarRef &r = vg.sc_g[isc];
    if (gp) { g.base = gp; g.nf_arr = gp_nfld; g.fidx = gcount; }
    else if (r.arr == 1) { g.base = gp2; g.nf_arr = vg.nsc_g[1] * dmul; g.fidx = r.lev + kder * vg.nsc_g[1]; }
    else if (r.arr == 2) { g.base = gp3a; g.nf_arr = vg.nsc_g[2] * nvar3a * dmul; g.fidx = (r.var + kder * nvar3a) * vg.nsc_g[2] + r.lev; }
    else { g.base = gp3b; g.nf_arr = vg.nsc_g[3] * nvar3b * dmul; g.fidx = (r.var + kder * nvar3b) * vg.nsc_g[3] + r.lev; }
    out.g.push_back(g), out.owner.push_back(vg.osc[isc]);
    gcount++;
  };
  if (nuv) {
    if (lvorgp) { for (int i = 0; i < nuv; i++) uvf(i); uvvar++; }
    if (ldivgp) { for (int i = 0; i < nuv; i++) uvf(i); uvvar++; }
    for (int i = 0; i < nuv; i++) uvf(i);
    uvvar++;
    for (int i = 0; i < nuv; i++) uvf(i);
    uvvar++;
  }
  for (int i = 0; i < nsc; i++) scf(i, 0);
  if (lscders) for (int i = 0; i < nsc; i++) scf(i, 1);
  if (luvder && nuv) {
    for (int i = 0; i < nuv; i++) uvf(i);
    uvvar++;
    for (int i = 0; i < nuv; i++) uvf(i);
    uvvar++;
  }
  if (lscders) for (int i = 0; i < nsc; i++) scf(i, 2);
}
// pack / unpack one block of the V-exchange: nf fields x npts points between two lists of grid-field descriptors
static int v_copy(Plan &P, const std::vector<GridFld> &src, const std::vector<GridFld> &dst, long long sp0, long long dp0, long long npts, long long snp,
                  long long dnp, std::vector<StageBuf *> &keep, emi_stream_t st) {
  const int nf = (int)src.size();
  if (nf == 0 || npts == 0) return 0;
  StageBuf *d = new StageBuf(2 * (size_t)nf * sizeof(GridFld));
  keep.push_back(d);
  if (!d->p) EMI_FAIL(EMI_ERR_RUNTIME, "V-set exchange: no device memory for %d field descriptors", nf);
  std::vector<GridFld> both(src);
  both.insert(both.end(), dst.begin(), dst.end());
  if (emi_h2d(d->p, both.data(), both.size() * sizeof(GridFld), st)) return EMI_ERR_RUNTIME;
  const long long nblk = ((long long)nf * npts + 255) / 256;
  EMI_LAUNCH_P(P.esz, k_gridcopy, nblk, 256, 0, st, (const GridFld *)d->p, (const GridFld *)d->p + nf, nf, sp0, dp0, (int)npts, (int)snp, (int)dnp);
  return 0;
}
static GridFld dense_field(void *base, size_t field, long long npts, int esz) {
  GridFld g{};
  g.base = (char *)base + field * (size_t)npts * esz;
  g.nf_arr = 1;
  g.fidx = 0;
  return g;
}

static int inv_trans_vsets(int kresol, const emi_invtrans_t *ap, bool adj) {
  Plan *Pp = get_plan(kresol);
  const char *who = adj ? "DIR_TRANSAD" : "INV_TRANS";
  if (!Pp) EMI_FAIL(EMI_ERR_STATE, "%s: unknown resolution %d", who, kresol);
  if (!ap) EMI_FAIL(EMI_ERR_ARG, "%s: null argument block", who);
  Plan &P = *Pp;
  const emi_invtrans_t &a = *ap;
  emi_stream_t st = (emi_stream_t)a.stream;
  const bool host = a.mem_space == EMI_MEM_HOST;
  VGroups vg;
  if (v_groups(P, a, who, vg)) return EMI_ERR_ARG;
  const int nuvg = vg.nuv_g, nscg = (int)vg.sc_g.size();
  const bool lscders = a.ldscders && nscg > 0, lvorgp = a.ldvorgp != 0, ldivgp = a.lddivgp != 0 || lvorgp, luvder = a.lduvder && nuvg > 0;
  const int if_gp_g = 2 * nuvg + nscg + (lscders ? 2 * nscg : 0) + ((nuvg && lvorgp) ? nuvg : 0) + ((nuvg && ldivgp) ? nuvg : 0) + (luvder ? 2 * nuvg : 0);
  if (if_gp_g == 0) return EMI_SUCCESS;
  const int nvar_uv = ((nuvg && lvorgp) ? 1 : 0) + ((nuvg && ldivgp) ? 1 : 0) + 2 + (luvder ? 2 : 0), dmul = lscders ? 3 : 1;
  const long long myp = P.vpoints(P.me, P.mev), bandp = P.ngptot;
  const int nproma = a.kproma > 0 ? a.kproma : (int)myp;
  const int ngpblks = (int)((myp - 1) / nproma + 1);
  if (a.gp) {
    if (a.gpuv || a.gp3a || a.gp3b || a.gp2) EMI_FAIL(EMI_ERR_ARG, "%s:PGP AND PGPUV/PGP3A/PGP3B/PGP2 CAN NOT BOTH BE PRESENT", who);
    if (a.gp_nfld < if_gp_g) EMI_FAIL(EMI_ERR_ARG, "%s:SECOND DIMENSION OF PGP TOO SMALL (%d < %d)", who, a.gp_nfld, if_gp_g);
  } else {
    if (nuvg > 0 && !a.gpuv) EMI_FAIL(EMI_ERR_ARG, "%s:PGPUV MISSING", who);
    if (vg.nsc_g[0] > 0) EMI_FAIL(EMI_ERR_ARG, "%s:PGP MISSING (PSPSCALAR needs PGP)", who);
    if (vg.nsc_g[1] > 0 && !a.gp2) EMI_FAIL(EMI_ERR_ARG, "%s:PGP2 MISSING", who);
    if (vg.nsc_g[2] * vg.nvar3a > 0 && !a.gp3a) EMI_FAIL(EMI_ERR_ARG, "%s:PGP3A MISSING", who);
    if (vg.nsc_g[3] * vg.nvar3b > 0 && !a.gp3b) EMI_FAIL(EMI_ERR_ARG, "%s:PGP3B MISSING", who);
  }
  // ---- the caller's arrays on the device (spectral: local fields; grid: all fields on this task's points)
  HostStage hs(P.esz);
  const size_t ns2 = P.nspec2, gsz = (size_t)nproma * ngpblks;
  const bool gpad = gsz != (size_t)myp;
  emi_invtrans_t in = a;
  in.mem_space = EMI_MEM_DEVICE;
  in.ext = nullptr;
  in.vsets = nullptr;
  in.spvor = hs.in(a.spvor, ns2 * a.nf_uv, host, st), in.spdiv = hs.in(a.spdiv, ns2 * a.nf_uv, host, st);
  in.spscalar = hs.in(a.spscalar, ns2 * a.nf_scalar, host, st), in.spsc2 = hs.in(a.spsc2, ns2 * a.nf_sc2, host, st);
  in.spsc3a = hs.in(a.spsc3a, ns2 * a.sc3a_nlev * a.sc3a_nvar, host, st), in.spsc3b = hs.in(a.spsc3b, ns2 * a.sc3b_nlev * a.sc3b_nvar, host, st);
  void *d_gp = hs.out(a.gp, gsz * a.gp_nfld, host, gpad || a.gp_nfld > if_gp_g, st);
  void *d_gpuv = hs.out(a.gpuv, gsz * nuvg * nvar_uv, host && nuvg, gpad, st);
  void *d_gp2 = hs.out(a.gp2, gsz * vg.nsc_g[1] * dmul, host, gpad, st);
  void *d_gp3a = hs.out(a.gp3a, gsz * vg.nsc_g[2] * vg.nvar3a * dmul, host, gpad, st);
  void *d_gp3b = hs.out(a.gp3b, gsz * vg.nsc_g[3] * vg.nvar3b * dmul, host, gpad, st);
  if (hs.failed) EMI_FAIL(EMI_ERR_RUNTIME, "%s: cannot stage the host arrays through device memory (%s)", who, emi_last_error());
  VGridList gl;
  v_grid_fields(vg, lvorgp, ldivgp, lscders, luvder, d_gp, a.gp_nfld, d_gpuv, nvar_uv, d_gp2, d_gp3a, d_gp3b, gl);
  std::vector<int> nl(P.nprv, 0);  // grid fields every V-set computes
  for (int o : gl.owner) nl[o]++;
  const int nlm = nl[P.mev];
  // ---- the W-set transform of the local fields into a band-sized array [field][band point]
  StageBuf tband((size_t)nlm * bandp * P.esz), sbuf((size_t)nlm * bandp * P.esz), rbuf((size_t)if_gp_g * myp * P.esz);
  if ((nlm && (!tband.p || !sbuf.p)) || !rbuf.p) EMI_FAIL(EMI_ERR_RUNTIME, "%s: no device memory for the exchange between the V-sets", who);
  if (nlm) {
    in.gp = tband.p, in.gp_nfld = nlm, in.gpuv = in.gp3a = in.gp3b = in.gp2 = nullptr, in.kproma = (int)bandp;
    const int rc = inv_trans_impl(kresol, &in, adj);
    if (rc) return rc;
  }
  // ---- TRLTOG: block (me -> v') = my fields on the points of sub-band v'
  std::vector<StageBuf *> keep;
  std::vector<long long> sc(P.nprv), rc(P.nprv);
  std::vector<GridFld> tb(nlm);
  for (int j = 0; j < nlm; j++) tb[j].base = tband.p, tb[j].nf_arr = nlm, tb[j].fidx = j;
  int bad = 0;
  size_t soff = 0;
  for (int v = 0; v < P.nprv && !bad; v++) {
    const long long np = P.vpoints(P.me, v);
    std::vector<GridFld> ds(nlm);
    for (int j = 0; j < nlm; j++) ds[j] = dense_field((char *)sbuf.p + soff, j, np, P.esz);
    bad = v_copy(P, tb, ds, P.voffset(v), 0, np, bandp, np, keep, st);
    sc[v] = (long long)nlm * np * P.esz;
    rc[v] = (long long)nl[v] * myp * P.esz;
    soff += (size_t)sc[v];
  }
  if (!bad) bad = hook_alltoallv(sbuf.p, sc, rbuf.p, rc, P.nprv, 1, P.me * P.nprv, st);
  size_t roff = 0;
  for (int v = 0; v < P.nprv && !bad; v++) {
    std::vector<GridFld> sr, dd;
    for (size_t k = 0; k < gl.g.size(); k++)
      if (gl.owner[k] == v) {
        sr.push_back(dense_field((char *)rbuf.p + roff, sr.size(), myp, P.esz));
        dd.push_back(gl.g[k]);
      }
    bad = v_copy(P, sr, dd, 0, 0, myp, myp, nproma, keep, st);
    roff += (size_t)rc[v];
  }
  if (host && !bad) hs.flush(st);
  else emi_stream_sync(st);  // the scratch buffers go back to the pool below
  for (StageBuf *k : keep) delete k;
  return bad ? EMI_ERR_RUNTIME : EMI_SUCCESS;
}

static int dir_trans_vsets(int kresol, const emi_dirtrans_t *ap, bool adj, const AdjOpts *ao) {
  Plan *Pp = get_plan(kresol);
  const char *who = adj ? "INV_TRANSAD" : "DIR_TRANS";
  if (!Pp) EMI_FAIL(EMI_ERR_STATE, "%s: unknown resolution %d", who, kresol);
  if (!ap) EMI_FAIL(EMI_ERR_ARG, "%s: null argument block", who);
  Plan &P = *Pp;
  const emi_dirtrans_t &a = *ap;
  if (ao && (ao->scders || ao->vorgp || ao->divgp || ao->uvder))
    EMI_FAIL(EMI_ERR_UNSUPPORTED, "INV_TRANSAD: LDSCDERS / LDVORGP / LDDIVGP / LDUVDER are not supported with NPRTRV > 1");
  emi_stream_t st = (emi_stream_t)a.stream;
  const bool host = a.mem_space == EMI_MEM_HOST;
  VGroups vg;
  if (v_groups(P, a, who, vg)) return EMI_ERR_ARG;
  const int nuvg = vg.nuv_g, nscg = (int)vg.sc_g.size();
  const int if_gp_g = 2 * nuvg + nscg;
  if (if_gp_g == 0) return EMI_SUCCESS;
  const long long myp = P.vpoints(P.me, P.mev), bandp = P.ngptot;
  const int nproma = a.kproma > 0 ? a.kproma : (int)myp;
  const int ngpblks = (int)((myp - 1) / nproma + 1);
  if (a.gp) {
    if (a.gpuv || a.gp3a || a.gp3b || a.gp2) EMI_FAIL(EMI_ERR_ARG, "%s:PGP AND PGPUV/PGP3A/PGP3B/PGP2 CAN NOT BOTH BE PRESENT", who);
    if (a.gp_nfld < if_gp_g) EMI_FAIL(EMI_ERR_ARG, "%s:SECOND DIMENSION OF PGP TOO SMALL (%d < %d)", who, a.gp_nfld, if_gp_g);
  } else {
    if (nuvg > 0 && !a.gpuv) EMI_FAIL(EMI_ERR_ARG, "%s:PGPUV MISSING", who);
    if (vg.nsc_g[0] > 0) EMI_FAIL(EMI_ERR_ARG, "%s:PGP MISSING (PSPSCALAR needs PGP)", who);
    if (vg.nsc_g[1] > 0 && !a.gp2) EMI_FAIL(EMI_ERR_ARG, "%s:PGP2 MISSING", who);
    if (vg.nsc_g[2] * vg.nvar3a > 0 && !a.gp3a) EMI_FAIL(EMI_ERR_ARG, "%s:PGP3A MISSING", who);
    if (vg.nsc_g[3] * vg.nvar3b > 0 && !a.gp3b) EMI_FAIL(EMI_ERR_ARG, "%s:PGP3B MISSING", who);
  }
  HostStage hs(P.esz);
  const size_t ns2 = P.nspec2, gsz = (size_t)nproma * ngpblks;
  emi_dirtrans_t in = a;
  in.mem_space = EMI_MEM_DEVICE;
  in.ext = nullptr;
  in.vsets = nullptr;
  in.spvor = hs.out(a.spvor, ns2 * a.nf_uv, host), in.spdiv = hs.out(a.spdiv, ns2 * a.nf_uv, host);
  in.spscalar = hs.out(a.spscalar, ns2 * a.nf_scalar, host), in.spsc2 = hs.out(a.spsc2, ns2 * a.nf_sc2, host);
  in.spsc3a = hs.out(a.spsc3a, ns2 * a.sc3a_nlev * a.sc3a_nvar, host), in.spsc3b = hs.out(a.spsc3b, ns2 * a.sc3b_nlev * a.sc3b_nvar, host);
  void *d_gp = (void *)hs.in(a.gp, gsz * a.gp_nfld, host, st);
  void *d_gpuv = (void *)hs.in(a.gpuv, gsz * nuvg * 2, host && nuvg, st);
  void *d_gp2 = (void *)hs.in(a.gp2, gsz * vg.nsc_g[1], host, st);
  void *d_gp3a = (void *)hs.in(a.gp3a, gsz * vg.nsc_g[2] * vg.nvar3a, host, st);
  void *d_gp3b = (void *)hs.in(a.gp3b, gsz * vg.nsc_g[3] * vg.nvar3b, host, st);
  if (hs.failed) EMI_FAIL(EMI_ERR_RUNTIME, "%s: cannot stage the host arrays through device memory (%s)", who, emi_last_error());
  VGridList gl;  // u(nuv_g) v(nuv_g) scalars: dir_trans.F90:301
  v_grid_fields(vg, false, false, false, false, d_gp, a.gp_nfld, d_gpuv, 2, d_gp2, d_gp3a, d_gp3b, gl);
  std::vector<int> nl(P.nprv, 0);
  for (int o : gl.owner) nl[o]++;
  const int nlm = nl[P.mev];
  StageBuf tband((size_t)nlm * bandp * P.esz), rbuf((size_t)nlm * bandp * P.esz), sbuf((size_t)if_gp_g * myp * P.esz);
  if ((nlm && (!tband.p || !rbuf.p)) || !sbuf.p) EMI_FAIL(EMI_ERR_RUNTIME, "%s: no device memory for the exchange between the V-sets", who);
  // ---- TRGTOL: block (me -> v'') = the fields of V-set v'' on my points
  std::vector<StageBuf *> keep;
  std::vector<long long> sc(P.nprv), rc(P.nprv);
  int bad = 0;
  size_t soff = 0;
  for (int v = 0; v < P.nprv && !bad; v++) {
    std::vector<GridFld> sr, dd;
    for (size_t k = 0; k < gl.g.size(); k++)
      if (gl.owner[k] == v) {
        dd.push_back(dense_field((char *)sbuf.p + soff, dd.size(), myp, P.esz));
        sr.push_back(gl.g[k]);
      }
    bad = v_copy(P, sr, dd, 0, 0, myp, nproma, myp, keep, st);
    sc[v] = (long long)nl[v] * myp * P.esz;
    rc[v] = (long long)nlm * P.vpoints(P.me, v) * P.esz;
    soff += (size_t)sc[v];
  }
  if (!bad) bad = hook_alltoallv(sbuf.p, sc, rbuf.p, rc, P.nprv, 1, P.me * P.nprv, st);
  std::vector<GridFld> tb(nlm);
  for (int j = 0; j < nlm; j++) tb[j].base = tband.p, tb[j].nf_arr = nlm, tb[j].fidx = j;
  size_t roff = 0;
  for (int v = 0; v < P.nprv && !bad; v++) {
    const long long np = P.vpoints(P.me, v);
    std::vector<GridFld> sr(nlm);
    for (int j = 0; j < nlm; j++) sr[j] = dense_field((char *)rbuf.p + roff, j, np, P.esz);
    bad = v_copy(P, sr, tb, 0, P.voffset(v), np, np, bandp, keep, st);
    roff += (size_t)rc[v];
  }
  int rcode = bad ? EMI_ERR_RUNTIME : EMI_SUCCESS;
  if (!bad && nlm) {
    in.gp = tband.p, in.gp_nfld = nlm, in.gpuv = in.gp3a = in.gp3b = in.gp2 = nullptr, in.kproma = (int)bandp;
    rcode = dir_trans_impl(kresol, &in, adj, nullptr);
  }
  if (host && rcode == EMI_SUCCESS) hs.flush(st);
  else emi_stream_sync(st);
  for (StageBuf *k : keep) delete k;
  return rcode;
}

// EMI_MEM_AUTO of a transform call: every array of the argument block is classified (emi_ptr_space); the call then runs as
// EMI_MEM_DEVICE (arrays used in place) or EMI_MEM_HOST (staged) -- or is refused when the arrays are in both places
// EMI_MEM_AUTO with several tasks: every task learns whether ANY task found its arrays in both memories, and then all of them fail together
// with the text of the offending task (one 4-byte word per task over the host collectives; host and device callers may mix between tasks,
// staging is local).  Without registered collectives (a host that drives the exchange hook only) the outcome must be identical on every
// task by the caller's construction, as include/ectrans_mi.h says.
static int agree_on_mixed_arrays(const char *who, int ndev, int nhost) {
  const int mixed = (ndev && nhost) ? 1 : 0;
  if (G.nproc_all > 1 && G.hc_gather) {
    const int NA = G.nproc_all;
    std::vector<int> all(NA, 0);
    std::vector<long long> cnt(NA, 4), dsp(NA);
    for (int r = 0; r < NA; r++) dsp[r] = 4LL * r;
    if (G.hc_gather(G.hc_user, &mixed, 4, all.data(), cnt.data(), dsp.data(), NA)) EMI_FAIL(EMI_ERR_RUNTIME, "%s: all-gather-v failed", who);
    for (int r = 0; r < NA; r++)
      if (all[r] && !mixed)
        EMI_FAIL(EMI_ERR_ARG, "%s: ARRAYS OF THE CALL ARE IN DEVICE MEMORY AND IN HOST MEMORY ON TASK %d (all of them must live in one place)", who, r + 1);
  }
  if (mixed)
    EMI_FAIL(EMI_ERR_ARG, "%s: %d ARRAYS OF THE CALL ARE IN DEVICE MEMORY AND %d IN HOST MEMORY (all of them must live in one place)", who, ndev, nhost);
  return 0;
}
template <class A>
static int resolve_call(const char *who, const A *ap, A &a) {
  if (!ap) EMI_FAIL(EMI_ERR_ARG, "%s: null argument block", who);
  a = *ap;
  return resolve_space(who, ap->mem_space, {ap->spvor, ap->spdiv, ap->spscalar, ap->spsc3a, ap->spsc3b, ap->spsc2, ap->gp, ap->gpuv, ap->gp3a, ap->gp3b, ap->gp2},
                       &a.mem_space, true);
}
extern "C" int emi_inv_trans(int kresol, const emi_invtrans_t *args) {
  EmiRange rg(EMI_LBL_INV);  // GSTATS 4
  emi_invtrans_t a;
  if (resolve_call("INV_TRANS", args, a)) return EMI_ERR_ARG;
  if (G.nprtrv > 1) return inv_trans_vsets(kresol, &a, false);
  return inv_trans_impl(kresol, &a, false);
}
extern "C" int emi_dir_trans(int kresol, const emi_dirtrans_t *args) {
  EmiRange rg(EMI_LBL_DIR);  // GSTATS 5
  emi_dirtrans_t a;
  if (resolve_call("DIR_TRANS", args, a)) return EMI_ERR_ARG;
  if (G.nprtrv > 1) return dir_trans_vsets(kresol, &a, false, nullptr);
  return dir_trans_impl(kresol, &a, false);
}
extern "C" int emi_wait(int kresol) {
  if (!G.init) EMI_FAIL(EMI_ERR_STATE, "emi_wait: SETUP_TRANS0 has not been called");
  if (kresol > 0) {
    Plan *Pp = get_plan(kresol);
    if (!Pp) EMI_FAIL(EMI_ERR_STATE, "emi_wait: unknown resolution %d", kresol);
    if (const int rc = plan_quiesce(*Pp)) EMI_FAIL(EMI_ERR_RUNTIME, "emi_wait: the last call of resolution %d did not complete (%s)", kresol, emi_rt_errstr(rc));
    return EMI_SUCCESS;
  }
  for (size_t i = 0; i < G.plans.size(); i++) {
    Plan *Pp = G.plans[i];
    if (!Pp || !Pp->active) continue;
    if (const int rc = plan_quiesce(*Pp)) EMI_FAIL(EMI_ERR_RUNTIME, "emi_wait: the last call of resolution %d did not complete (%s)", (int)i, emi_rt_errstr(rc));
  }
  return EMI_SUCCESS;
}

// INV_TRANSAD (include/ectrans/inv_transad.h): arguments of INV_TRANS with the intents swapped
extern "C" int emi_inv_transad(int kresol, const emi_invtrans_t *ap) {
  emi_invtrans_t a;
  if (resolve_call("INV_TRANSAD", ap, a)) return EMI_ERR_ARG;
  emi_dirtrans_t d{};
  d.mem_space = a.mem_space;
  d.spvor = (void *)a.spvor, d.spdiv = (void *)a.spdiv, d.nf_uv = a.nf_uv;
  d.spscalar = (void *)a.spscalar, d.nf_scalar = a.nf_scalar;
  d.spsc3a = (void *)a.spsc3a, d.sc3a_nlev = a.sc3a_nlev, d.sc3a_nvar = a.sc3a_nvar;
  d.spsc3b = (void *)a.spsc3b, d.sc3b_nlev = a.sc3b_nlev, d.sc3b_nvar = a.sc3b_nvar;
  d.spsc2 = (void *)a.spsc2, d.nf_sc2 = a.nf_sc2;
  d.kproma = a.kproma;
  d.gp = a.gp, d.gp_nfld = a.gp_nfld, d.gpuv = a.gpuv, d.gp3a = a.gp3a, d.gp3b = a.gp3b, d.gp2 = a.gp2;
  d.stream = a.stream;
  d.ext = a.ext;
  d.vsets = a.vsets;
  // LDSCDERS / LDVORGP / LDDIVGP / LDUVDER: the grid arrays then carry the derivative / vorticity / divergence inputs in
  // INV_TRANS's layout (ltinvad_mod.F90:149-225, spnsdead_mod.F90, fscad_mod.F90)
  AdjOpts ao;
  ao.scders = a.ldscders != 0, ao.vorgp = a.ldvorgp != 0, ao.divgp = a.lddivgp != 0 || a.ldvorgp != 0, ao.uvder = a.lduvder != 0;
  if (G.nprtrv > 1) return dir_trans_vsets(kresol, &d, true, &ao);
  return dir_trans_impl(kresol, &d, true, &ao);
}
// DIR_TRANSAD (include/ectrans/dir_transad.h): arguments of DIR_TRANS with the intents swapped
extern "C" int emi_dir_transad(int kresol, const emi_dirtrans_t *ap) {
  emi_dirtrans_t d;
  if (resolve_call("DIR_TRANSAD", ap, d)) return EMI_ERR_ARG;
  emi_invtrans_t a{};
  a.mem_space = d.mem_space;
  a.spvor = d.spvor, a.spdiv = d.spdiv, a.nf_uv = d.nf_uv;
  a.spscalar = d.spscalar, a.nf_scalar = d.nf_scalar;
  a.spsc3a = d.spsc3a, a.sc3a_nlev = d.sc3a_nlev, a.sc3a_nvar = d.sc3a_nvar;
  a.spsc3b = d.spsc3b, a.sc3b_nlev = d.sc3b_nlev, a.sc3b_nvar = d.sc3b_nvar;
  a.spsc2 = d.spsc2, a.nf_sc2 = d.nf_sc2;
  a.kproma = d.kproma;
  a.gp = (void *)d.gp, a.gp_nfld = d.gp_nfld, a.gpuv = (void *)d.gpuv, a.gp3a = (void *)d.gp3a, a.gp3b = (void *)d.gp3b,
  a.gp2 = (void *)d.gp2;
  a.stream = d.stream;
  a.ext = d.ext;
  a.vsets = d.vsets;
  if (G.nprtrv > 1) return inv_trans_vsets(kresol, &a, true);
  return inv_trans_impl(kresol, &a, true);
}

#if defined(EMI_MR_STAMP) && !defined(EMI_CPU_EMU)
// experiments only (-DEMI_MR_STAMP): read and clear the stage clocks of k_fft_dir_mr
extern "C" int emi_debug_mr_stamps(unsigned long long *out) {
  hipDeviceSynchronize();
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(emi_mr_stamp), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
  unsigned long long z[16] = {0};
  return hipMemcpyToSymbol(HIP_SYMBOL(emi_mr_stamp), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif
