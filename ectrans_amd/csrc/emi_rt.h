// emi_rt.h -- thin device-runtime layer for the ectrans-mi kernels.
//
// Product build (hipcc, gfx950): everything maps 1:1 onto HIP.
//
// EMI_CPU_EMU build (g++ -fopenmp): a *functional emulator* used only by the CPU test-suite
// (tests/emu): one OS thread per lane, `#pragma omp barrier` for __syncthreads and an
// emulated v_mfma_f64_16x16x4_f64.  It exists so that kernel index logic can be debugged
// in a container without a GPU.  It is never loaded by the product API (ectrans_amd/*.py
// only opens libectrans_mi.so, the HIP build) and is not a fallback path.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>

#ifndef EMI_CPU_EMU
#include <hip/hip_runtime.h>

#define EMI_KERNEL __global__
#define EMI_KERNEL_LB(T) __global__ __launch_bounds__(T)
// FFT kernels: any workgroup size up to 1024, W waves per SIMD.  LDS limits the fp64 kernels to <= 5
// waves per SIMD, so W = 4 lets the scheduler spend 128 VGPRs on keeping loads in flight instead of
// chasing 8 waves per SIMD; the fp32 kernels need half the LDS and registers and run at W = 8.
#define EMI_KERNEL_FFT(W) __global__ __attribute__((amdgpu_flat_work_group_size(64, 1024), amdgpu_waves_per_eu(W, W)))
// direct mixed-radix FFT kernels: workgroups of 128 - 256 threads (mr_choose)
#define EMI_KERNEL_MR(W) __global__ __attribute__((amdgpu_flat_work_group_size(64, 256), amdgpu_waves_per_eu(W, W)))
#define EMI_KERNEL_LB2(T, W) __global__ __attribute__((amdgpu_flat_work_group_size(T, T), amdgpu_waves_per_eu(W, W)))
#define EMI_DEVFN __device__ __forceinline__
#define EMI_TID ((int)threadIdx.x)
#define EMI_BID ((int)blockIdx.x)
#define EMI_NTHREADS ((int)blockDim.x)
#define EMI_SYNC() __syncthreads()
// ordering of LDS accesses WITHIN one wave (its LDS instructions execute in order): a compiler fence is
// all that is needed between a pass that wrote and a pass that reads the same wave-private LDS block
#define EMI_WAVE_SYNC()                                   \
  do {                                                    \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
    __builtin_amdgcn_wave_barrier();                      \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
  } while (0)
// pointer known to address global memory (a pointer loaded from a descriptor is generic to the compiler: flat_load / flat_store, which
// also count against lgkmcnt): EMI_GLOBAL_AS real2 *p = (EMI_GLOBAL_AS real2 *)q; access the members, not the struct
#define EMI_GLOBAL_AS __attribute__((address_space(1)))
#define EMI_LDS_DECL extern __shared__ __attribute__((aligned(16))) char emi_lds_raw[]
// workgroup barrier that orders LDS accesses only: unlike __syncthreads() it does not wait for the wave's outstanding
// global loads (vmcnt), so table loads issued ahead of an LDS exchange stay in flight across it
#define EMI_LDS_SYNC() __asm__ volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
// compiler-only fence between LDS accesses of ONE wave (its LDS instructions execute in order)
#define EMI_WAVE_FENCE() __asm__ volatile("" ::: "memory")
// the instruction scheduler moves nothing across this point
#define EMI_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// first statement of a function body whose floating-point operations must round one by one (no fused multiply-adds): code that is
// instantiated in two places and has to give bit-identical results in both (fin_pair: decomposition invariance)
#define EMI_FP_STRICT() _Pragma("clang fp contract(off)")

// Buffer-descriptor access to tables and rows (k_fft_*_r16): one 32-bit lane offset, row / leg offsets in scalar registers,
// no 64-bit vector address arithmetic; a read whose lane offset + immediate is past `bytes` returns zero and such a write is
// dropped (raw buffer, stride 0: the range check of the hardware).  The scalar offset `soff` is NOT part of the check.
struct EmiBuf {
  __amdgpu_buffer_rsrc_t r;
};
typedef int emi_v4i __attribute__((ext_vector_type(4)));
typedef int emi_v2i __attribute__((ext_vector_type(2)));
EMI_DEVFN EmiBuf emi_buf(const void *p, unsigned bytes) {
  EmiBuf b;
  b.r = __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, bytes, 0x00020000);
  return b;
}
// a whole table: no range limit (the caller's lane offsets are inside it)
EMI_DEVFN EmiBuf emi_buf_all(const void *p) { return emi_buf(p, 0xFFFFFFFCu); }
template <class V>
EMI_DEVFN V emi_buf_ld(const EmiBuf &b, unsigned voff, unsigned soff) {
  static_assert(sizeof(V) == 16 || sizeof(V) == 8, "emi_buf_ld: 8- or 16-byte values");
  if constexpr (sizeof(V) == 16)
    return __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b128(b.r, voff, soff, 0));
  else
    return __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b64(b.r, voff, soff, 0));
}
template <class V>
EMI_DEVFN void emi_buf_st(const EmiBuf &b, unsigned voff, unsigned soff, V v) {
  static_assert(sizeof(V) == 16 || sizeof(V) == 8, "emi_buf_st: 8- or 16-byte values");
  if constexpr (sizeof(V) == 16)
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(emi_v4i, v), b.r, voff, soff, 0);
  else
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(emi_v2i, v), b.r, voff, soff, 0);
}
// Scalar addressing of loads whose base is uniform over the wave (Legendre stage loops): `emi_uniform*` move a value that IS the same in
// every lane into scalar registers (v_readfirstlane), and a load from uniform base + 32-bit lane offset is one
// `global_load_dwordx4 v, v_off, s[base:base+1]` -- the per-stage pointer arithmetic then runs on the scalar unit.  On gfx950 every
// vector instruction, 32-bit moves included, takes issue time from the pipe the fp64 matrix instructions run on
// (tools/dp_pipe_probe.hip: 2.5 - 5 cycles of a matrix instruction's 64 per 32-bit instruction, 5 - 8 per 64-bit one).
EMI_DEVFN int emi_uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }
EMI_DEVFN const char *emi_uniform_ptr(const void *p) {
  const unsigned long long a = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  return (const char *)(((unsigned long long)hi << 32) | lo);
}
template <class V>
EMI_DEVFN V emi_ld_sv(const char *ubase, unsigned voff) {
  return *(const EMI_GLOBAL_AS V *)((const EMI_GLOBAL_AS char *)ubase + voff);
}
// element `idx` (uniform) of a table that nothing writes while kernels run: read through the constant address space, which makes it a
// scalar load (s_load_dword) wherever it stands -- behind a workgroup barrier a plain load is a vector load + v_readfirstlane whose
// s_waitcnt vmcnt also waits for every operand prefetch issued before it
EMI_DEVFN int emi_ld_const(const int *tab, int idx) { return ((const __attribute__((address_space(4))) int *)tab)[idx]; }
// Wave priority around the MFMA block of the Legendre stage loops: the two waves of a SIMD are in
// different phases (one issues MFMAs, the other address arithmetic, LDS writes and loads for its next
// stage); with the MFMA wave at the higher priority its next MFMA never queues behind the other wave's
// vector instructions.  Measured +2.5 % on both Legendre kernels (priority 1 and 3 alike) in round 1; since the stage loops lost their
// vector address arithmetic (round 4) the kernels run the same with and without it (87.7 / 98.1 against 88.2 / 98.1 ms).
#define EMI_PRIO_HI() __builtin_amdgcn_s_setprio(1)
#define EMI_PRIO_LO() __builtin_amdgcn_s_setprio(0)
// the value of a per-lane integer becomes opaque to the optimiser at this point: index arithmetic that depends on it
// is recomputed where it is used (an add) instead of being hoisted out of the loop into registers that then spill
#define EMI_OPAQUE(x) __asm__ volatile("" : "+v"(x))
#define EMI_LDS_PTR (emi_lds_raw)

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));  // native 16-byte vector: whole-value copies stay in VGPRs
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

EMI_DEVFN v4d emi_mfma_f64(double a, double b, v4d c) {
  // v_mfma_f64_16x16x4_f64: A[row=l&15][k=l>>4], B[k=l>>4][col=l&15],
  // C/D: col=l&15, row=(l>>4)+4*i  (cdna_hip_programming.md §3)
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
EMI_DEVFN v4f emi_mfma_f32(float a, float b, v4f c) {
  // v_mfma_f32_16x16x4_f32: same A/B lane maps; C/D: col=l&15, row=4*(l>>4)+i (exact f32 fma chain)
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

typedef hipStream_t emi_stream_t;

#define EMI_CHECK(expr)                                                                 \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess) {                                                             \
      emi_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return -1;                                                                        \
    }                                                                                   \
  } while (0)

#define EMI_LAUNCH(kernel, grid, block, lds, stream, ...) \
  hipLaunchKernelGGL(kernel, dim3((unsigned)(grid)), dim3((unsigned)(block)), (size_t)(lds), (stream), __VA_ARGS__)

#else  // ------------------------------- CPU functional emulator ----------------------------
#include <omp.h>
#include <vector>

#define EMI_KERNEL
#define EMI_KERNEL_LB(T)
#define EMI_KERNEL_FFT(W)
#define EMI_KERNEL_LB2(T, W)
#define EMI_KERNEL_MR(W)
#define EMI_DEVFN inline
struct EmuCtx {
  int tid, bid, nthreads;
  char *lds;
  double *sa, *sb;  // mfma exchange scratch [nwaves][64]
};
extern thread_local EmuCtx *emu_ctx;
#define EMI_TID (emu_ctx->tid)
#define EMI_BID (emu_ctx->bid)
#define EMI_NTHREADS (emu_ctx->nthreads)
static inline void emu_barrier() {
#pragma omp barrier
}
#define EMI_SYNC() emu_barrier()
#define EMI_WAVE_SYNC() emu_barrier()  // lanes are threads here: a real barrier
#define EMI_LDS_SYNC() emu_barrier()
#define EMI_WAVE_FENCE() emu_barrier()  // lanes are threads here: a real barrier (every thread of the workgroup reaches it)
#define EMI_SCHED_FENCE() ((void)0)
#define EMI_FP_STRICT()
struct EmiBuf {
  const char *p;
  unsigned bytes;
};
inline EmiBuf emi_buf(const void *p, unsigned bytes) { return EmiBuf{(const char *)p, bytes}; }
inline EmiBuf emi_buf_all(const void *p) { return emi_buf(p, 0xFFFFFFFCu); }
template <class V>
inline V emi_buf_ld(const EmiBuf &b, unsigned voff, unsigned soff) {
  V v;
  memset(&v, 0, sizeof(V));
  if ((size_t)voff + sizeof(V) <= b.bytes) memcpy(&v, b.p + voff + soff, sizeof(V));
  return v;
}
template <class V>
inline void emi_buf_st(const EmiBuf &b, unsigned voff, unsigned soff, V v) {
  if ((size_t)voff + sizeof(V) <= b.bytes) memcpy((char *)b.p + voff + soff, &v, sizeof(V));
}
inline int emi_uniform(int x) { return x; }
inline const char *emi_uniform_ptr(const void *p) { return (const char *)p; }
inline int emi_ld_const(const int *tab, int idx) { return tab[idx]; }
template <class V>
inline V emi_ld_sv(const char *ubase, unsigned voff) {
  V v;
  memcpy(&v, ubase + voff, sizeof(V));
  return v;
}
#define EMI_GLOBAL_AS
#define EMI_LDS_DECL
#define EMI_OPAQUE(x) ((void)0)
#define EMI_PRIO_HI() ((void)0)
#define EMI_PRIO_LO() ((void)0)
#define EMI_LDS_PTR (emu_ctx->lds)

typedef double v4d __attribute__((vector_size(32)));
struct __attribute__((aligned(16))) d2 {
  double x, y;
};
struct int2 {
  int x, y;
};
typedef float v4f __attribute__((vector_size(16)));
struct __attribute__((aligned(8))) f2 {
  float x, y;
};

inline v4d emi_mfma_f64(double a, double b, v4d c) {
  EmuCtx *x = emu_ctx;
  int w = x->tid >> 6, l = x->tid & 63;
  x->sa[w * 64 + l] = a;
  x->sb[w * 64 + l] = b;
#pragma omp barrier
  for (int i = 0; i < 4; i++) {
    int row = (l >> 4) + 4 * i, col = l & 15;
    double s = c[i];
    for (int k = 0; k < 4; k++) s += x->sa[w * 64 + k * 16 + row] * x->sb[w * 64 + k * 16 + col];
    c[i] = s;
  }
#pragma omp barrier
  return c;
}
inline v4f emi_mfma_f32(float a, float b, v4f c) {
  EmuCtx *x = emu_ctx;
  int w = x->tid >> 6, l = x->tid & 63;
  x->sa[w * 64 + l] = a;
  x->sb[w * 64 + l] = b;
#pragma omp barrier
  for (int i = 0; i < 4; i++) {
    int row = 4 * (l >> 4) + i, col = l & 15;  // f32 accumulator row map
    float s = c[i];
    for (int k = 0; k < 4; k++) s = fmaf((float)x->sa[w * 64 + k * 16 + row], (float)x->sb[w * 64 + k * 16 + col], s);
    c[i] = s;
  }
#pragma omp barrier
  return c;
}

typedef void *emi_stream_t;
typedef int hipError_t;
#define hipSuccess 0

template <class F, class... A>
static void emu_launch(F f, long grid, int block, size_t lds, A... a) {
  std::vector<char> sh(lds + 64);
  std::vector<double> sa((block / 64 + 1) * 64), sb((block / 64 + 1) * 64);
  char *ldsp = (char *)(((uintptr_t)sh.data() + 15) & ~(uintptr_t)15);
  for (long b = 0; b < grid; b++) {
#pragma omp parallel num_threads(block)
    {
      EmuCtx c{omp_get_thread_num(), (int)b, block, ldsp, sa.data(), sb.data()};
      emu_ctx = &c;
      f(a...);
    }
  }
}
#define EMI_LAUNCH(kernel, grid, block, lds, stream, ...) emu_launch(kernel, (long)(grid), (int)(block), (size_t)(lds), __VA_ARGS__)
#define EMI_CHECK(expr) \
  do {                  \
    (void)(expr);       \
  } while (0)
#endif

void emi_set_error(const char *fmt, ...);

// ---- memory / stream helpers (same signatures in both builds) -------------------------
int emi_dev_malloc(void **p, size_t bytes);
int emi_dev_free(void *p);
int emi_dev_memset(void *p, int v, size_t bytes, emi_stream_t s);
int emi_h2d(void *dst, const void *src, size_t bytes, emi_stream_t s);
int emi_d2h(void *dst, const void *src, size_t bytes, emi_stream_t s);
int emi_d2d(void *dst, const void *src, size_t bytes, emi_stream_t s);
int emi_stream_sync(emi_stream_t s);
int emi_mem_info(size_t *free_b, size_t *total_b);
