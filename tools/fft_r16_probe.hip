// fft_r16_probe.hip -- prototype of the register-resident Bluestein convolution core (round 3).
//
//   hipcc -O3 --offload-arch=gfx950 tools/fft_r16_probe.hip -o tools/fft_r16_probe && tools/fft_r16_probe [rows] [reps]
//
// One workgroup = 256 threads = 4 waves = one row of S = 4096 = 16 * 16 * 16 complex points; every thread keeps 16 points
// in registers through the whole chain
//     A1 (DIF radix 16, stride 256)  X  A2 (stride 16)  L  A3 (stride 1) * filter * B3  L  B2  X  B1
// and the points change threads through ONE real plane of LDS at a time (real parts, then imaginary parts: 34 KiB per
// row instead of 64 KiB, so four independent 4-wave workgroups share a CU instead of two 8-wave ones).  X exchanges cross
// waves (3 workgroup barriers each); L exchanges stay inside a 16-lane row of a wave (no barrier at all).
// The probe measures rows per second on synthetic rows and checks the convolution against a host FFT.
#include <hip/hip_runtime.h>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <type_traits>

#define DEVFN static __device__ __forceinline__
#define CK(x)                                                                         \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      printf("%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);       \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

constexpr int S = 4096, ROWP = 272;  // LDS plane: 16 blocks of 256 (+16 pad) doubles
constexpr double PI = 3.14159265358979323846;

// ---------------------------------------------------------------------------------------------------------------------
// radix-16 butterflies on re[16], im[16].  Y[k] = sum_a x[a] W^(a k), W = exp(SGN 2 pi i / 16).
//   bf16_nr: natural input (register a holds x[a]) -> register j = 4 k1 + k0 holds Y[k1 + 4 k0]   ("reversed")
//   bf16_rn: reversed input (register 4 k1 + k0 holds x[k1 + 4 k0]) -> register c holds Y[c]       (natural)
// NZ: inputs a >= NZ are zero (NZ = 8: the first pass of a zero-padded convolution); NOUT: outputs k >= NOUT are not
// needed (NOUT = 8: the last pass, of which only the first half is read).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int rev16(int j) { return (j >> 2) + 4 * (j & 3); }

template <int SGN>
DEVFN void r4(double &ar, double &ai, double &br, double &bi, double &cr, double &ci, double &dr, double &di) {
  const double t0r = ar + cr, t0i = ai + ci, t1r = ar - cr, t1i = ai - ci;
  const double t2r = br + dr, t2i = bi + di, t3r = br - dr, t3i = bi - di;
  ar = t0r + t2r, ai = t0i + t2i;
  cr = t0r - t2r, ci = t0i - t2i;
  if (SGN < 0) {  // W4 = -i: W4 t3 = (t3i, -t3r)
    br = t1r + t3i, bi = t1i - t3r;
    dr = t1r - t3i, di = t1i + t3r;
  } else {
    br = t1r - t3i, bi = t1i + t3r;
    dr = t1r + t3i, di = t1i - t3r;
  }
}
// radix 4 with x2 = x3 = 0
template <int SGN>
DEVFN void r4_hz(double &ar, double &ai, double &br, double &bi, double &cr, double &ci, double &dr, double &di) {
  const double x0r = ar, x0i = ai, x1r = br, x1i = bi;
  ar = x0r + x1r, ai = x0i + x1i;
  cr = x0r - x1r, ci = x0i - x1i;
  if (SGN < 0) {
    br = x0r + x1i, bi = x0i - x1r;
    dr = x0r - x1i, di = x0i + x1r;
  } else {
    br = x0r - x1i, bi = x0i + x1r;
    dr = x0r + x1i, di = x0i - x1r;
  }
}
// radix 4, outputs 0 and 1 only
template <int SGN>
DEVFN void r4_h2(double &ar, double &ai, double &br, double &bi, double cr, double ci, double dr, double di) {
  const double t0r = ar + cr, t0i = ai + ci, t1r = ar - cr, t1i = ai - ci;
  const double t2r = br + dr, t2i = bi + di, t3r = br - dr, t3i = bi - di;
  ar = t0r + t2r, ai = t0i + t2i;
  if (SGN < 0)
    br = t1r + t3i, bi = t1i - t3r;
  else
    br = t1r - t3i, bi = t1i + t3r;
}
// x *= exp(SGN 2 pi i e / 16), e a compile-time constant
template <int SGN, int E>
DEVFN void w16(double &xr, double &xi) {
  constexpr int e = E & 15;
  if constexpr (e == 0) {
  } else if constexpr (e == 4) {
    const double t = xr;
    if (SGN < 0)
      xr = xi, xi = -t;
    else
      xr = -xi, xi = t;
  } else if constexpr (e == 8) {
    xr = -xr, xi = -xi;
  } else if constexpr (e == 12) {
    const double t = xr;
    if (SGN < 0)
      xr = -xi, xi = t;
    else
      xr = xi, xi = -t;
  } else {
    constexpr double cs[16] = {1.0, 0.92387953251128673848, 0.70710678118654752440, 0.38268343236508977173, 0.0, -0.38268343236508977173,
                               -0.70710678118654752440, -0.92387953251128673848, -1.0, -0.92387953251128673848, -0.70710678118654752440,
                               -0.38268343236508977173, 0.0, 0.38268343236508977173, 0.70710678118654752440, 0.92387953251128673848};
    constexpr double c = cs[e], s = (double)SGN * cs[(e + 12) & 15];  // sin(x) = cos(x - pi/2)
    const double tr = xr * c - xi * s, ti = xr * s + xi * c;
    xr = tr, xi = ti;
  }
}

template <int SGN, int NZ = 16, int NOUT = 16>
DEVFN void bf16_nr(double *re, double *im) {
  // step 1: over a1 (registers a0, a0 + 4, a0 + 8, a0 + 12) -> u[a0][k1] in register 4 k1 + a0
#pragma unroll
  for (int a0 = 0; a0 < 4; a0++) {
    if (NZ <= 8)
      r4_hz<SGN>(re[a0], im[a0], re[a0 + 4], im[a0 + 4], re[a0 + 8], im[a0 + 8], re[a0 + 12], im[a0 + 12]);
    else
      r4<SGN>(re[a0], im[a0], re[a0 + 4], im[a0 + 4], re[a0 + 8], im[a0 + 8], re[a0 + 12], im[a0 + 12]);
  }
  // step 2: u[a0][k1] *= W16^(a0 k1)
#define W16AT(k1, a0) w16<SGN, (k1) * (a0)>(re[4 * (k1) + (a0)], im[4 * (k1) + (a0)])
  W16AT(1, 1); W16AT(1, 2); W16AT(1, 3);
  W16AT(2, 1); W16AT(2, 2); W16AT(2, 3);
  W16AT(3, 1); W16AT(3, 2); W16AT(3, 3);
#undef W16AT
  // step 3: over a0 (registers 4 k1 .. 4 k1 + 3) -> Y[k1 + 4 k0] in register 4 k1 + k0
#pragma unroll
  for (int k1 = 0; k1 < 4; k1++) {
    if (NOUT <= 8)
      r4_h2<SGN>(re[4 * k1], im[4 * k1], re[4 * k1 + 1], im[4 * k1 + 1], re[4 * k1 + 2], im[4 * k1 + 2], re[4 * k1 + 3], im[4 * k1 + 3]);
    else
      r4<SGN>(re[4 * k1], im[4 * k1], re[4 * k1 + 1], im[4 * k1 + 1], re[4 * k1 + 2], im[4 * k1 + 2], re[4 * k1 + 3], im[4 * k1 + 3]);
  }
}
template <int SGN>
DEVFN void bf16_rn(double *re, double *im) {
  // step 1: over k0 (registers 4 k1 .. 4 k1 + 3) -> v[k1][c0] in register 4 k1 + c0
#pragma unroll
  for (int k1 = 0; k1 < 4; k1++) r4<SGN>(re[4 * k1], im[4 * k1], re[4 * k1 + 1], im[4 * k1 + 1], re[4 * k1 + 2], im[4 * k1 + 2], re[4 * k1 + 3], im[4 * k1 + 3]);
#define W16AT(k1, c0) w16<SGN, (k1) * (c0)>(re[4 * (k1) + (c0)], im[4 * (k1) + (c0)])
  W16AT(1, 1); W16AT(1, 2); W16AT(1, 3);
  W16AT(2, 1); W16AT(2, 2); W16AT(2, 3);
  W16AT(3, 1); W16AT(3, 2); W16AT(3, 3);
#undef W16AT
  // step 3: over k1 (registers c0, c0 + 4, ...) -> Y[4 c1 + c0] in register 4 c1 + c0
#pragma unroll
  for (int c0 = 0; c0 < 4; c0++) r4<SGN>(re[c0], im[c0], re[c0 + 4], im[c0 + 4], re[c0 + 8], im[c0 + 8], re[c0 + 12], im[c0 + 12]);
}

// ---------------------------------------------------------------------------------------------------------------------
DEVFN void lds_barrier() { __asm__ volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
DEVFN void wave_fence() { __asm__ volatile("" ::: "memory"); }

struct Tabs {
  const double2 *tw1;    // [k0 = 1..15][t]   exp(-2 pi i t k0 / 4096)
  const double2 *tw2;    // [k1 = 1..15][c]   exp(-2 pi i c k1 / 256)
  const double2 *bhat;   // [k2][v]           filter spectrum at k0 + 16 k1 + 256 k2, v = 16 k0 + k1
  const double2 *chirp;  // [i], i < 2048
  const double2 *tw6;    // [6][t]: w^t, w^2t, w^3t, w^4t, w^8t, w^12t, w = exp(-2 pi i / 4096)
};

// VARIANT 0: full chain.  1: no arithmetic (exchanges + loads only).  2: no exchanges (arithmetic + loads only).
// TAB 0: every twiddle from global tables (15 + 15 + 15 + 15 loads per thread).  1: no table loads at all (constants: timing
// only, wrong numbers).  2: tw2 from an LDS copy, tw1 as w^(t k1) w^(4 t k0') from 6 loads (9 more complex products per pass)
template <int VARIANT, int TAB>
__global__ __launch_bounds__(256, 4) void k_conv4096(Tabs T, const double2 *__restrict__ zin, double2 *__restrict__ zout, int nout) {
  __shared__ double pl[16 * ROWP];
  __shared__ double2 tw2s[(TAB >= 2) ? 15 * 16 : 1];
  const int t = threadIdx.x, row = blockIdx.x;
  const int hi = t >> 4, lo = t & 15;  // (k0, c) / (k0, k1) roles
  double re[16], im[16];
  const double2 *zi = zin + (size_t)row * (S / 2);
  // ---- input: u[256 a + t] = z[256 a + t] chirp[256 a + t], a < 8 (the rest of the work array is zero)
#pragma unroll
  for (int a = 0; a < 8; a++) {
    const double2 z = (VARIANT == 3) ? make_double2(1.0 + t, 2.0 + a) : zi[256 * a + t], c = (TAB == 1) ? make_double2(0.6, 0.8) : T.chirp[256 * a + t];
    re[a] = z.x * c.x - z.y * c.y;
    im[a] = z.x * c.y + z.y * c.x;
  }
#pragma unroll
  for (int a = 8; a < 16; a++) re[a] = im[a] = 0.0;
  if (TAB >= 2) {
    if (t < 240) tw2s[t] = T.tw2[t];
    lds_barrier();
  }
  // ---- A1
  if (VARIANT != 1) bf16_nr<-1, 8, 16>(re, im);
  if (TAB == 2) {
    // register j holds k0 = (j >> 2) + 4 (j & 3): w^(t k0) = w^(t (j >> 2)) w^(4 t (j & 3)) from tw6[0..2] = w^t, w^2t, w^3t, tw6[3..5] = w^4t, w^8t, w^12t
    double2 wa[4], wb[4];
#pragma unroll
    for (int q = 1; q < 4; q++) wa[q] = T.tw6[(q - 1) * 256 + t], wb[q] = T.tw6[(q + 2) * 256 + t];
#pragma unroll
    for (int j = 1; j < 16; j++) {
      const int qa = j >> 2, qb = j & 3;
      double2 w;
      if (qa == 0) w = wb[qb];
      else if (qb == 0) w = wa[qa];
      else w = make_double2(wa[qa].x * wb[qb].x - wa[qa].y * wb[qb].y, wa[qa].x * wb[qb].y + wa[qa].y * wb[qb].x);
      const double xr = re[j], xi = im[j];
      re[j] = xr * w.x - xi * w.y;
      im[j] = xr * w.y + xi * w.x;
    }
  } else {
#pragma unroll
  for (int j = 1; j < 16; j++) {  // register j holds k0 = rev16(j); j = 0 is k0 = 0
    const int k0 = rev16(j);
    const double2 w = (TAB == 1) ? make_double2(0.6, 0.8) : T.tw1[(k0 - 1) * 256 + t];
    const double xr = re[j], xi = im[j];
    re[j] = xr * w.x - xi * w.y;
    im[j] = xr * w.y + xi * w.x;
  }
  }
  // ---- X1: thread t, value k0 -> thread (k0, c = t & 15), slot b = t >> 4
  if (VARIANT >= 2) wave_fence();
  if (VARIANT < 2) {
#pragma unroll
    for (int j = 0; j < 16; j++) pl[rev16(j) * ROWP + t] = re[j];
    lds_barrier();
#pragma unroll
    for (int b = 0; b < 16; b++) re[b] = pl[hi * ROWP + 16 * b + lo];
    lds_barrier();
#pragma unroll
    for (int j = 0; j < 16; j++) pl[rev16(j) * ROWP + t] = im[j];
    lds_barrier();
#pragma unroll
    for (int b = 0; b < 16; b++) im[b] = pl[hi * ROWP + 16 * b + lo];
  }
  // ---- A2 in thread (k0 = hi, c = lo)
  if (VARIANT != 1) bf16_nr<-1>(re, im);
#pragma unroll
  for (int j = 1; j < 16; j++) {
    const int k1 = rev16(j);
    const double2 w = (TAB == 1) ? make_double2(0.6, 0.8) : (TAB >= 2) ? tw2s[(k1 - 1) * 16 + lo] : T.tw2[(k1 - 1) * 16 + lo];
    const double xr = re[j], xi = im[j];
    re[j] = xr * w.x - xi * w.y;
    im[j] = xr * w.y + xi * w.x;
  }
  // ---- L1: thread (k0, c), value k1 -> thread (k0, k1), slot c; [k0][k1][c] at k0 ROWP + 17 k1 + c
  if (VARIANT >= 2) wave_fence();
  if (VARIANT < 2) {
    wave_fence();
#pragma unroll
    for (int j = 0; j < 16; j++) pl[hi * ROWP + 17 * rev16(j) + lo] = re[j];
    wave_fence();
#pragma unroll
    for (int c = 0; c < 16; c++) re[c] = pl[hi * ROWP + 17 * lo + c];
    wave_fence();
#pragma unroll
    for (int j = 0; j < 16; j++) pl[hi * ROWP + 17 * rev16(j) + lo] = im[j];
    wave_fence();
#pragma unroll
    for (int c = 0; c < 16; c++) im[c] = pl[hi * ROWP + 17 * lo + c];
  }
  // ---- A3, filter, B3 in thread v = (k0, k1) = t
  if (VARIANT != 1) bf16_nr<-1>(re, im);
#pragma unroll
  for (int j = 0; j < 16; j++) {
    const double2 b = (TAB == 1) ? make_double2(0.6, 0.8) : T.bhat[rev16(j) * 256 + t];
    const double xr = re[j], xi = im[j];
    re[j] = xr * b.x - xi * b.y;
    im[j] = xr * b.y + xi * b.x;
  }
  if (VARIANT != 1) bf16_rn<+1>(re, im);  // register c: value for (k0, k1; c)
  // twiddle of B2 (DIT: on the inputs), conj(w_256^(c k1)), k1 = lo
#pragma unroll
  for (int c = 1; c < 16; c++) {
    const double2 w = (TAB == 1) ? make_double2(0.6, 0.8) : (TAB >= 2) ? tw2s[(c - 1) * 16 + lo] : T.tw2[(c - 1) * 16 + lo];  // symmetric in (c, k1)
    const double xr = re[c], xi = im[c];
    re[c] = xr * w.x + xi * w.y;
    im[c] = xi * w.x - xr * w.y;
  }
  // ---- L2: thread (k0, k1), value c -> thread (k0, c), slot k1
  if (VARIANT >= 2) wave_fence();
  if (VARIANT < 2) {
    wave_fence();
#pragma unroll
    for (int c = 0; c < 16; c++) pl[hi * ROWP + 17 * lo + c] = re[c];
    wave_fence();
#pragma unroll
    for (int k1 = 0; k1 < 16; k1++) re[k1] = pl[hi * ROWP + 17 * k1 + lo];
    wave_fence();
#pragma unroll
    for (int c = 0; c < 16; c++) pl[hi * ROWP + 17 * lo + c] = im[c];
    wave_fence();
#pragma unroll
    for (int k1 = 0; k1 < 16; k1++) im[k1] = pl[hi * ROWP + 17 * k1 + lo];
  }
  // ---- B2 in thread (k0, c): over k1 -> b (register j holds b = rev16(j))
  if (VARIANT != 1) bf16_nr<+1>(re, im);
  // ---- X2: thread (k0, c), value b -> thread t = 16 b + c, slot k0
  if (VARIANT >= 2) wave_fence();
  if (VARIANT < 2) {
    wave_fence();
#pragma unroll
    for (int j = 0; j < 16; j++) pl[hi * ROWP + 16 * rev16(j) + lo] = re[j];
    lds_barrier();
#pragma unroll
    for (int k0 = 0; k0 < 16; k0++) re[k0] = pl[k0 * ROWP + t];
    lds_barrier();
#pragma unroll
    for (int j = 0; j < 16; j++) pl[hi * ROWP + 16 * rev16(j) + lo] = im[j];
    lds_barrier();
#pragma unroll
    for (int k0 = 0; k0 < 16; k0++) im[k0] = pl[k0 * ROWP + t];
  }
  // ---- B1 in thread t: inputs k0 times conj(w_4096^(t k0)), outputs a < 8
  if (TAB == 2) {
    double2 wa[4], wb[4];  // k0 = qa + 4 qb
#pragma unroll
    for (int q = 1; q < 4; q++) wa[q] = T.tw6[(q - 1) * 256 + t], wb[q] = T.tw6[(q + 2) * 256 + t];
#pragma unroll
    for (int k0 = 1; k0 < 16; k0++) {
      const int qa = k0 & 3, qb = k0 >> 2;
      double2 w;
      if (qa == 0) w = wb[qb];
      else if (qb == 0) w = wa[qa];
      else w = make_double2(wa[qa].x * wb[qb].x - wa[qa].y * wb[qb].y, wa[qa].x * wb[qb].y + wa[qa].y * wb[qb].x);
      const double xr = re[k0], xi = im[k0];
      re[k0] = xr * w.x + xi * w.y;
      im[k0] = xi * w.x - xr * w.y;
    }
  } else {
#pragma unroll
  for (int k0 = 1; k0 < 16; k0++) {
    const double2 w = (TAB == 1) ? make_double2(0.6, 0.8) : T.tw1[(k0 - 1) * 256 + t];
    const double xr = re[k0], xi = im[k0];
    re[k0] = xr * w.x + xi * w.y;
    im[k0] = xi * w.x - xr * w.y;
  }
  }
  if (VARIANT != 1) bf16_nr<+1, 16, 8>(re, im);
  double2 *zo = zout + (size_t)row * (S / 2);
#pragma unroll
  for (int j = 0; j < 16; j++) {
    const int a = rev16(j);
    if (a < 8 && (VARIANT != 3 || re[j] == 1.2345) && 256 * a < nout) {
      const double2 c = (TAB == 1) ? make_double2(0.6, 0.8) : T.chirp[256 * a + t];
      zo[256 * a + t] = make_double2(re[j] * c.x + im[j] * c.y, im[j] * c.x - re[j] * c.y);
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// v2: the TAB 2 chain with every batch of table loads issued explicitly ahead of the phase that uses it (the compiler
// otherwise pairs each load with its use: one exposed L2 latency per twiddle).  SB = __builtin_amdgcn_sched_barrier(0).
// LDSV: LDS accesses through a volatile pointer (no ds_read2_b64 / ds_write2_b64 merging).
#define SB() __builtin_amdgcn_sched_barrier(0)
typedef int v4i_t __attribute__((ext_vector_type(4)));
// tables and rows through buffer descriptors: one VGPR offset per thread, row / leg offsets in scalar registers, no 64-bit
// vector address arithmetic; reads past num_records return zero and such writes are dropped (voffset + imm is checked)
DEVFN __amdgpu_buffer_rsrc_t mk_rsrc(const void *p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, bytes, 0x00020000); }
DEVFN double2 bld(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
DEVFN void bst(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, double2 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i_t, v), r, voff, soff, 0);
}
DEVFN void cmul_ip(double &xr, double &xi, const double2 w) {
  const double tr = xr * w.x - xi * w.y, ti = xr * w.y + xi * w.x;
  xr = tr, xi = ti;
}
DEVFN void cmulc_ip(double &xr, double &xi, const double2 w) {
  const double tr = xr * w.x + xi * w.y, ti = xi * w.x - xr * w.y;
  xr = tr, xi = ti;
}
DEVFN double2 cmul2(const double2 a, const double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

template <int LDSV, int PF>
__global__ __launch_bounds__(256, 4) void k_conv4096_v2(Tabs T, const double2 *__restrict__ zin, double2 *__restrict__ zout, int nout) {
  __shared__ double pl_[16 * ROWP];
  __shared__ double2 tw2s[15 * 16];
  typedef typename std::conditional<LDSV != 0, volatile double, double>::type lds_t;
  lds_t *pl = pl_;
  const unsigned t = threadIdx.x, row = blockIdx.x;
  const unsigned hi = t >> 4, lo = t & 15, t16 = t * 16;
  const __amdgpu_buffer_rsrc_t r_in = mk_rsrc(zin + (size_t)row * (S / 2), (S / 2) * 16), r_out = mk_rsrc(zout + (size_t)row * (S / 2), nout * 16);
  const __amdgpu_buffer_rsrc_t r_ch = mk_rsrc(T.chirp, (S / 2) * 16), r_bh = mk_rsrc(T.bhat, S * 16), r_tw = mk_rsrc(T.tw6, 6 * 256 * 16);
  double re[16], im[16];
  double2 wa[4], wb[4];
  {
    double2 z[8], c[8];
#pragma unroll
    for (int a = 0; a < 8; a++) z[a] = bld(r_in, t16, 4096 * a);
#pragma unroll
    for (int a = 0; a < 8; a++) c[a] = bld(r_ch, t16, 4096 * a);
    if (t < 240) tw2s[t] = T.tw2[t];
    if (PF) SB();
#pragma unroll
    for (int a = 0; a < 8; a++) re[a] = z[a].x * c[a].x - z[a].y * c[a].y, im[a] = z[a].x * c[a].y + z[a].y * c[a].x;
    if (PF) SB();
#pragma unroll
    for (int q = 1; q < 4; q++) wa[q] = bld(r_tw, t16, 4096 * (q - 1)), wb[q] = bld(r_tw, t16, 4096 * (q + 2));
    if (PF) SB();
  }
#pragma unroll
  for (int a = 8; a < 16; a++) re[a] = im[a] = 0.0;
  // ---- A1
  bf16_nr<-1, 8, 16>(re, im);
#pragma unroll
  for (int j = 1; j < 16; j++) {
    const int qa = j >> 2, qb = j & 3;
    cmul_ip(re[j], im[j], qa == 0 ? wb[qb] : (qb == 0 ? wa[qa] : cmul2(wa[qa], wb[qb])));
    if (PF && (j & 1)) SB();
  }
  // ---- X1
#pragma unroll
  for (int j = 0; j < 16; j++) pl[rev16(j) * ROWP + t] = re[j];
  lds_barrier();
#pragma unroll
  for (int b = 0; b < 16; b++) re[b] = pl[hi * ROWP + 16 * b + lo];
  lds_barrier();
#pragma unroll
  for (int j = 0; j < 16; j++) pl[rev16(j) * ROWP + t] = im[j];
  lds_barrier();
#pragma unroll
  for (int b = 0; b < 16; b++) im[b] = pl[hi * ROWP + 16 * b + lo];
  // ---- A2
  bf16_nr<-1>(re, im);
#pragma unroll
  for (int j0 = 1; j0 < 16; j0 += 5) {
    double2 w[5];
#pragma unroll
    for (int i = 0; i < 5; i++) w[i] = tw2s[(rev16(j0 + i) - 1) * 16 + lo];
    SB();
#pragma unroll
    for (int i = 0; i < 5; i++) cmul_ip(re[j0 + i], im[j0 + i], w[i]);
    SB();
  }
  // filter values of the registers j < 8, in flight during the L1 exchange and the A3 butterfly
  double2 bA[8];
  if (PF) {
#pragma unroll
    for (int j = 0; j < 8; j++) bA[j] = bld(r_bh, t16, 4096 * rev16(j));
    SB();
  }
  // ---- L1
  wave_fence();
#pragma unroll
  for (int j = 0; j < 16; j++) pl[hi * ROWP + 17 * rev16(j) + lo] = re[j];
  wave_fence();
#pragma unroll
  for (int c = 0; c < 16; c++) re[c] = pl[hi * ROWP + 17 * lo + c];
  wave_fence();
#pragma unroll
  for (int j = 0; j < 16; j++) pl[hi * ROWP + 17 * rev16(j) + lo] = im[j];
  wave_fence();
#pragma unroll
  for (int c = 0; c < 16; c++) im[c] = pl[hi * ROWP + 17 * lo + c];
  // ---- A3, filter, B3
  bf16_nr<-1>(re, im);
  if (PF) {
    SB();
#pragma unroll
    for (int j = 0; j < 8; j++) cmul_ip(re[j], im[j], bA[j]);
#pragma unroll
    for (int j = 0; j < 8; j++) bA[j] = bld(r_bh, t16, 4096 * rev16(j + 8));
    SB();
#pragma unroll
    for (int j = 0; j < 8; j++) cmul_ip(re[j + 8], im[j + 8], bA[j]);
  } else {
#pragma unroll
    for (int j = 0; j < 16; j++) cmul_ip(re[j], im[j], bld(r_bh, t16, 4096 * rev16(j)));
  }
  bf16_rn<+1>(re, im);
#pragma unroll
  for (int j0 = 1; j0 < 16; j0 += 5) {
    double2 w[5];
#pragma unroll
    for (int i = 0; i < 5; i++) w[i] = tw2s[(j0 + i - 1) * 16 + lo];
    SB();
#pragma unroll
    for (int i = 0; i < 5; i++) cmulc_ip(re[j0 + i], im[j0 + i], w[i]);
    SB();
  }
  // ---- L2
  wave_fence();
#pragma unroll
  for (int c = 0; c < 16; c++) pl[hi * ROWP + 17 * lo + c] = re[c];
  wave_fence();
#pragma unroll
  for (int k1 = 0; k1 < 16; k1++) re[k1] = pl[hi * ROWP + 17 * k1 + lo];
  wave_fence();
#pragma unroll
  for (int c = 0; c < 16; c++) pl[hi * ROWP + 17 * lo + c] = im[c];
  wave_fence();
#pragma unroll
  for (int k1 = 0; k1 < 16; k1++) im[k1] = pl[hi * ROWP + 17 * k1 + lo];
  // ---- B2
  bf16_nr<+1>(re, im);
  // twiddles of B1, in flight during the X2 exchange
  if (PF) {
#pragma unroll
    for (int q = 1; q < 4; q++) wa[q] = bld(r_tw, t16, 4096 * (q - 1)), wb[q] = bld(r_tw, t16, 4096 * (q + 2));
    SB();
  }
  // ---- X2
  wave_fence();
#pragma unroll
  for (int j = 0; j < 16; j++) pl[hi * ROWP + 16 * rev16(j) + lo] = re[j];
  lds_barrier();
#pragma unroll
  for (int k0 = 0; k0 < 16; k0++) re[k0] = pl[k0 * ROWP + t];
  lds_barrier();
#pragma unroll
  for (int j = 0; j < 16; j++) pl[hi * ROWP + 16 * rev16(j) + lo] = im[j];
  lds_barrier();
#pragma unroll
  for (int k0 = 0; k0 < 16; k0++) im[k0] = pl[k0 * ROWP + t];
  if (!PF) {
#pragma unroll
    for (int q = 1; q < 4; q++) wa[q] = bld(r_tw, t16, 4096 * (q - 1)), wb[q] = bld(r_tw, t16, 4096 * (q + 2));
  }
  // ---- B1
#pragma unroll
  for (int k0 = 1; k0 < 16; k0++) {
    const int qa = k0 & 3, qb = k0 >> 2;
    cmulc_ip(re[k0], im[k0], qa == 0 ? wb[qb] : (qb == 0 ? wa[qa] : cmul2(wa[qa], wb[qb])));
    if (PF && (k0 & 1)) SB();
  }
  // chirp of the outputs, in flight during the B1 butterfly
  double2 co[8];
  if (PF) SB();
#pragma unroll
  for (int a = 0; a < 8; a++) co[a] = bld(r_ch, t16, 4096 * a);
  if (PF) SB();
  bf16_nr<+1, 16, 8>(re, im);
  if (PF) SB();
#pragma unroll
  for (int j = 0; j < 16; j++) {
    const int a = rev16(j);
    if (a < 8) bst(r_out, t16 + 4096 * a, 0, make_double2(re[j] * co[a].x + im[j] * co[a].y, im[j] * co[a].x - re[j] * co[a].y));
  }
}
template <int LDSV, int PF>
static double run2(Tabs T, const double2 *zin, double2 *zout, int rows, int reps, int nout) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k_conv4096_v2<LDSV, PF>), dim3(rows), dim3(256), 0, 0, T, zin, zout, nout);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; r++) hipLaunchKernelGGL((k_conv4096_v2<LDSV, PF>), dim3(rows), dim3(256), 0, 0, T, zin, zout, nout);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

// ---------------------------------------------------------------------------------------------------------------------
typedef std::complex<double> cd;
static void host_fft(std::vector<cd> &a, int sgn) {  // in-place radix-2, natural order
  const int n = (int)a.size();
  for (int i = 1, j = 0; i < n; i++) {
    int bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) std::swap(a[i], a[j]);
  }
  for (int len = 2; len <= n; len <<= 1) {
    for (int i = 0; i < n; i += len)
      for (int k = 0; k < len / 2; k++) {
        const cd w = std::polar(1.0, sgn * 2.0 * PI * k / len);
        const cd u = a[i + k], v = a[i + k + len / 2] * w;
        a[i + k] = u + v, a[i + k + len / 2] = u - v;
      }
  }
}

template <int V, int TAB>
static double run(Tabs T, const double2 *zin, double2 *zout, int rows, int reps, int nout = 2048) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k_conv4096<V, TAB>), dim3(rows), dim3(256), 0, 0, T, zin, zout, nout);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; r++) hipLaunchKernelGGL((k_conv4096<V, TAB>), dim3(rows), dim3(256), 0, 0, T, zin, zout, nout);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main(int argc, char **argv) {
  const int rows = argc > 1 ? atoi(argv[1]) : 16384, reps = argc > 2 ? atoi(argv[2]) : 10;
  std::vector<cd> tw1(15 * 256), tw2(15 * 16), bh(S), bhat(S), chirp(S / 2);
  for (int k0 = 1; k0 < 16; k0++)
    for (int t = 0; t < 256; t++) tw1[(k0 - 1) * 256 + t] = std::polar(1.0, -2.0 * PI * t * k0 / 4096.0);
  for (int k1 = 1; k1 < 16; k1++)
    for (int c = 0; c < 16; c++) tw2[(k1 - 1) * 16 + c] = std::polar(1.0, -2.0 * PI * c * k1 / 256.0);
  srand(1);
  for (int i = 0; i < S; i++) bh[i] = cd(rand() / (double)RAND_MAX - 0.5, rand() / (double)RAND_MAX - 0.5);
  for (int k0 = 0; k0 < 16; k0++)
    for (int k1 = 0; k1 < 16; k1++)
      for (int k2 = 0; k2 < 16; k2++) bhat[k2 * 256 + 16 * k0 + k1] = bh[k0 + 16 * k1 + 256 * k2];
  for (int i = 0; i < S / 2; i++) chirp[i] = std::polar(1.0, -PI * (double)((long long)i * i % 5000) / 2500.0);
  std::vector<cd> zin((size_t)rows * (S / 2));
  for (size_t i = 0; i < zin.size(); i++) zin[i] = cd(rand() / (double)RAND_MAX - 0.5, rand() / (double)RAND_MAX - 0.5);
  Tabs T;
  double2 *d_tw1, *d_tw2, *d_bhat, *d_chirp, *d_zin, *d_zout;
  CK(hipMalloc(&d_tw1, tw1.size() * 16));
  CK(hipMalloc(&d_tw2, tw2.size() * 16));
  CK(hipMalloc(&d_bhat, bhat.size() * 16));
  CK(hipMalloc(&d_chirp, chirp.size() * 16));
  CK(hipMalloc(&d_zin, zin.size() * 16));
  CK(hipMalloc(&d_zout, zin.size() * 16));
  CK(hipMemcpy(d_tw1, tw1.data(), tw1.size() * 16, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_tw2, tw2.data(), tw2.size() * 16, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_bhat, bhat.data(), bhat.size() * 16, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_chirp, chirp.data(), chirp.size() * 16, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_zin, zin.data(), zin.size() * 16, hipMemcpyHostToDevice));
  T.tw1 = d_tw1, T.tw2 = d_tw2, T.bhat = d_bhat, T.chirp = d_chirp;
  std::vector<cd> tw6(6 * 256);
  for (int q = 1; q < 4; q++)
    for (int t = 0; t < 256; t++) tw6[(q - 1) * 256 + t] = std::polar(1.0, -2.0 * PI * t * q / 4096.0), tw6[(q + 2) * 256 + t] = std::polar(1.0, -2.0 * PI * t * 4 * q / 4096.0);
  double2 *d_tw6;
  CK(hipMalloc(&d_tw6, tw6.size() * 16));
  CK(hipMemcpy(d_tw6, tw6.data(), tw6.size() * 16, hipMemcpyHostToDevice));
  T.tw6 = d_tw6;

  // modes: name -> launcher; each timed `reps` launches, best of 3 rounds, rounds interleaved over the modes
  struct Mode { const char *name; double (*fn)(Tabs, const double2 *, double2 *, int, int, int); int nout; bool check; };
  std::vector<Mode> all = {
      {"v1_tab0", [](Tabs T, const double2 *a, double2 *b, int r, int n, int no) { return run<0, 0>(T, a, b, r, n, no); }, 1024, true},
      {"v1_tab2", [](Tabs T, const double2 *a, double2 *b, int r, int n, int no) { return run<0, 2>(T, a, b, r, n, no); }, 1024, true},
      {"v1_tab3", [](Tabs T, const double2 *a, double2 *b, int r, int n, int no) { return run<0, 3>(T, a, b, r, n, no); }, 1024, true},
      {"v1_notab", [](Tabs T, const double2 *a, double2 *b, int r, int n, int no) { return run<0, 1>(T, a, b, r, n, no); }, 1024, false},
      {"v1_notab_noarith", [](Tabs T, const double2 *a, double2 *b, int r, int n, int no) { return run<1, 1>(T, a, b, r, n, no); }, 1024, false},
      {"v1_notab_noxchg", [](Tabs T, const double2 *a, double2 *b, int r, int n, int no) { return run<2, 1>(T, a, b, r, n, no); }, 1024, false},
      {"arith_only", [](Tabs T, const double2 *a, double2 *b, int r, int n, int no) { return run<3, 1>(T, a, b, r, n, no); }, 1024, false},
      {"v2_plain", [](Tabs T, const double2 *a, double2 *b, int r, int n, int no) { return run2<0, 0>(T, a, b, r, n, no); }, 1024, true},
      {"v2_pf", [](Tabs T, const double2 *a, double2 *b, int r, int n, int no) { return run2<0, 1>(T, a, b, r, n, no); }, 1024, true},
  };
  std::vector<Mode> sel;
  for (int a = 3; a < argc; a++)
    for (auto &m : all)
      if (!strcmp(argv[a], m.name)) sel.push_back(m);
  if (sel.empty()) sel = all;
  double worst = 0.0;
  std::vector<double> best(sel.size(), 1e30);
  for (int round = 0; round < 3; round++)
    for (size_t m = 0; m < sel.size(); m++) {
      CK(hipMemset(d_zout, 0, zin.size() * 16));
      best[m] = fmin(best[m], sel[m].fn(T, d_zin, d_zout, rows, reps, sel[m].nout));
      if (round == 0 && sel[m].check) {
        std::vector<cd> zout(zin.size());
        CK(hipMemcpy(zout.data(), d_zout, zin.size() * 16, hipMemcpyDeviceToHost));
        double w = 0.0;
        for (int r : {0, rows / 2, rows - 1}) {
          std::vector<cd> a(S, cd(0, 0));
          for (int i = 0; i < S / 2; i++) a[i] = zin[(size_t)r * (S / 2) + i] * chirp[i];
          host_fft(a, -1);
          for (int i = 0; i < S; i++) a[i] *= bh[i];
          host_fft(a, +1);
          double mx = 0.0, err = 0.0;
          for (int i = 0; i < sel[m].nout; i++) {
            const cd ref = a[i] * std::conj(chirp[i]);
            mx = fmax(mx, std::abs(ref));
            err = fmax(err, std::abs(ref - zout[(size_t)r * (S / 2) + i]));
          }
          w = fmax(w, err / mx);
        }
        printf("%-18s check: max rel err %.3e %s\n", sel[m].name, w, w < 1e-12 ? "OK" : "WRONG");
        worst = fmax(worst, w);
      }
    }
  // reference point: k_fft_dir_hot<4> of round 2 takes 23.82 ms for 512 latitudes x 1645 fields = 28.3 ns per row
  for (size_t m = 0; m < sel.size(); m++) printf("%-18s %.3f ms = %6.2f ns per row (%d outputs)\n", sel[m].name, best[m], best[m] * 1e6 / rows, sel[m].nout);
  return worst < 1e-12 ? 0 : 1;
}
