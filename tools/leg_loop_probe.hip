// Where does k_leg_inv's main loop lose its MFMA throughput?  The loop of ectrans_amd/csrc/emi_kernels_body.h
// (tile 64 x 128, both parities, 8-row stages, 32 MFMAs per wave and stage) rebuilt piece by piece:
//   V0 MFMAs only | V1 + LDS fragment reads | V2 + the two barriers per stage | V3 + the LDS tile writes
//   V4 + global prefetch of the next stage (coalesced reads of a 1 GiB buffer) | V5 the same, two stages ahead
//   V8-V13 the operands by LDS-DMA into a ring of three stages (see probe_glds)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/probe tools/leg_loop_probe.hip && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double v4d __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
#define LDA 80
#define LDB 144

__device__ int g_random_operands = 0;

template <int V>
__global__ __attribute__((amdgpu_flat_work_group_size(256, 256), amdgpu_waves_per_eu(2, 2))) void probe(double *out, const double *src, int nst, long long stride) {
  extern __shared__ double lds[];
  double *As = lds, *Bs = lds + 2 * 8 * LDA;
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, wm = w & 1, wn = w >> 1;
  for (int i = tid; i < 2 * 8 * LDA + 2 * 8 * LDB; i += 256) {
    unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u;
    h ^= h >> 15, h *= 2246822519u, h ^= h >> 13;
    lds[i] = g_random_operands ? (double)(int)h * (1.0 / 2147483648.0) * (1.0 + 1e-13 * (h & 1023)) : 1.0 + 1e-9 * i;
  }
  __syncthreads();
  v4d acc[2][2][4];
  for (int p = 0; p < 2; p++)
    for (int i = 0; i < 2; i++)
      for (int j = 0; j < 4; j++) acc[p][i][j] = (v4d){0, 0, 0, 0};
  const int arow = tid >> 5, ac2 = tid & 31, brow = tid >> 6, bc2 = tid & 63;
  const double *g = src + (long long)blockIdx.x * 4096 + tid * 2;
  d2 ra0 = {1, 2}, ra1 = {3, 4}, rb0 = {5, 6}, rb1 = {7, 8}, rb2 = {9, 10}, rb3 = {11, 12};
  d2 qa0 = ra0, qa1 = ra1, qb0 = rb0, qb1 = rb1, qb2 = rb2, qb3 = rb3;
  double a0 = 1.0 + 1e-9 * l, b0 = 1.0 - 1e-9 * l;
  // touch address: line (k, w, i) of a stage = load k (4 KB apart), wave w (1 KB), 128-byte line i
  const int tt = tid < 192 ? tid : tid - 64;  // threads 192..255 repeat lines of the last loads
  const double *gt = src + (long long)blockIdx.x * 4096 + (tt >> 5) * 512 + ((tt >> 3) & 3) * 128 + (tt & 7) * 16;
  int tch = 0;
  for (int s = 0; s < nst; s++) {
    if (V >= 2 && s > 0) __syncthreads();
    if (V >= 3) {
      const bool alt = (V == 5) && !(s & 1);  // V5: store the set loaded two stages ago
      *(d2 *)(As + (0 * 8 + arow) * LDA + 2 * ac2) = alt ? qa0 : ra0;
      *(d2 *)(As + (1 * 8 + arow) * LDA + 2 * ac2) = alt ? qa1 : ra1;
      *(d2 *)(Bs + (((brow + 0) & 1) * 8 + ((brow + 0) >> 1)) * LDB + 2 * bc2) = alt ? qb0 : rb0;
      *(d2 *)(Bs + (((brow + 4) & 1) * 8 + ((brow + 4) >> 1)) * LDB + 2 * bc2) = alt ? qb1 : rb1;
      *(d2 *)(Bs + (((brow + 8) & 1) * 8 + ((brow + 8) >> 1)) * LDB + 2 * bc2) = alt ? qb2 : rb2;
      *(d2 *)(Bs + (((brow + 12) & 1) * 8 + ((brow + 12) >> 1)) * LDB + 2 * bc2) = alt ? qb3 : rb3;
    }
    if (V >= 2) __syncthreads();
    if (V == 23) {
      // V23: V4 + "touch": behind the six loads of stage s+1 every thread below 192 reads ONE dword of one of the 192 lines of
      // stage s+2 and drops it a stage later -- the line is then in L2 when the real loads of the next stage ask for it.  The
      // touch is younger than the loads, so the waits on them (vmcnt(1) ... ) leave it outstanding; loads return in order.
      asm volatile("" ::"v"(tch));  // the touch issued a stage ago has had a whole stage to arrive
      const double *q = g + (long long)s * stride;
      ra0 = *(const d2 *)q, ra1 = *(const d2 *)(q + 512), rb0 = *(const d2 *)(q + 1024), rb1 = *(const d2 *)(q + 1536);
      rb2 = *(const d2 *)(q + 2048), rb3 = *(const d2 *)(q + 2560);
      __builtin_amdgcn_sched_barrier(0);
      tch = *(const int *)(gt + (long long)(s + 1) * stride);  // every thread (no branch: the compiler must be able to count it)
      __builtin_amdgcn_sched_barrier(0);
    }
    if (V == 4 || V == 6 || V == 50 || V == 51) {
      const double *q = g + ((V == 6 || V == 51) ? 0 : (long long)s * stride);  // V6 / V51: the same lines every stage (cache hits)
      ra0 = *(const d2 *)q, ra1 = *(const d2 *)(q + 512), rb0 = *(const d2 *)(q + 1024), rb1 = *(const d2 *)(q + 1536);
      rb2 = *(const d2 *)(q + 2048), rb3 = *(const d2 *)(q + 2560);
      __builtin_amdgcn_sched_barrier(0);  // keep the loads here, ahead of the MFMAs (the compiler would sink them)
    }
    if (V == 5) {  // two stages ahead: two register sets used alternately (the store above takes the older one)
      const double *q = g + (long long)s * stride;
      if (s & 1) {
        qa0 = *(const d2 *)q, qa1 = *(const d2 *)(q + 512), qb0 = *(const d2 *)(q + 1024), qb1 = *(const d2 *)(q + 1536);
        qb2 = *(const d2 *)(q + 2048), qb3 = *(const d2 *)(q + 2560);
      } else {
        ra0 = *(const d2 *)q, ra1 = *(const d2 *)(q + 512), rb0 = *(const d2 *)(q + 1024), rb1 = *(const d2 *)(q + 1536);
        rb2 = *(const d2 *)(q + 2048), rb3 = *(const d2 *)(q + 2560);
      }
    }
    if (V >= 50) {
      // V50 / V51 (round 6): ONE parity per wave -- wave (parity wm, column half wn) owns 64 latitudes x 64 columns of its
      // parity as 4 x 4 fragments: 8 LDS fragment reads per 16 MFMAs instead of 12 (2 x (2 + 4)); same loads, LDS stores, barriers and
      // MFMA count per wave and stage as V4 / V6.  (The epilogue it would need -- the parities exchanged through LDS for north = S + A,
      // south = S - A -- is not part of the loop and not modelled.)
#pragma unroll
      for (int ks = 0; ks < 2; ks++) {
        const int kk = 4 * ks + (l >> 4);
        double a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; i++) a[i] = As[(wm * 8 + kk) * LDA + i * 16 + (l & 15)];
#pragma unroll
        for (int j = 0; j < 4; j++) b[j] = Bs[(wm * 8 + kk) * LDB + wn * 64 + j * 16 + (l & 15)];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
          for (int j = 0; j < 4; j++) acc[i >> 1][i & 1][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i >> 1][i & 1][j], 0, 0, 0);
      }
    } else
#pragma unroll
    for (int p = 0; p < 2; p++)
#pragma unroll
      for (int ks = 0; ks < 2; ks++) {
        const int kk = 4 * ks + (l >> 4);
        double a[2], b[4];
        if (V >= 1) {
#pragma unroll
          for (int i = 0; i < 2; i++) a[i] = As[(p * 8 + kk) * LDA + wm * 32 + i * 16 + (l & 15)];
#pragma unroll
          for (int j = 0; j < 4; j++) b[j] = Bs[(p * 8 + kk) * LDB + wn * 64 + j * 16 + (l & 15)];
        } else {
          a[0] = a[1] = a0;
          b[0] = b[1] = b[2] = b[3] = b0;
        }
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
          for (int j = 0; j < 4; j++) acc[p][i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[p][i][j], 0, 0, 0);
      }
  }
  double sum = ra0.x + rb3.y + qa0.x + qb3.y + tch;
  for (int p = 0; p < 2; p++)
    for (int i = 0; i < 2; i++)
      for (int j = 0; j < 4; j++) sum += acc[p][i][j][0] + acc[p][i][j][3];
  out[(long long)blockIdx.x * 256 + tid] = sum;
}


// two-stage-ahead prefetch done properly: the stage loop unrolled by two, register sets X and Y used
// alternately (stage s stores the set loaded during stage s-2 and reloads it for stage s+2)
#define PROBE_STAGE(SA0, SA1, SB0, SB1, SB2, SB3)                                                         \
  {                                                                                                       \
    if (s > 0) __syncthreads();                                                                           \
    *(d2 *)(As + (0 * 8 + arow) * LDA + 2 * ac2) = SA0;                                                   \
    *(d2 *)(As + (1 * 8 + arow) * LDA + 2 * ac2) = SA1;                                                   \
    *(d2 *)(Bs + (((brow + 0) & 1) * 8 + ((brow + 0) >> 1)) * LDB + 2 * bc2) = SB0;                       \
    *(d2 *)(Bs + (((brow + 4) & 1) * 8 + ((brow + 4) >> 1)) * LDB + 2 * bc2) = SB1;                       \
    *(d2 *)(Bs + (((brow + 8) & 1) * 8 + ((brow + 8) >> 1)) * LDB + 2 * bc2) = SB2;                       \
    *(d2 *)(Bs + (((brow + 12) & 1) * 8 + ((brow + 12) >> 1)) * LDB + 2 * bc2) = SB3;                     \
    __syncthreads();                                                                                      \
    {                                                                                                     \
      const double *q = g + (long long)s * stride;                                                        \
      SA0 = *(const d2 *)q, SA1 = *(const d2 *)(q + 512), SB0 = *(const d2 *)(q + 1024), SB1 = *(const d2 *)(q + 1536); \
      SB2 = *(const d2 *)(q + 2048), SB3 = *(const d2 *)(q + 2560);                                       \
      __builtin_amdgcn_sched_barrier(0);                                                                  \
    }                                                                                                     \
    _Pragma("unroll") for (int p = 0; p < 2; p++) _Pragma("unroll") for (int ks = 0; ks < 2; ks++) {      \
      const int kk = 4 * ks + (l >> 4);                                                                   \
      double a[2], b[4];                                                                                  \
      _Pragma("unroll") for (int i = 0; i < 2; i++) a[i] = As[(p * 8 + kk) * LDA + wm * 32 + i * 16 + (l & 15)];   \
      _Pragma("unroll") for (int j = 0; j < 4; j++) b[j] = Bs[(p * 8 + kk) * LDB + wn * 64 + j * 16 + (l & 15)];   \
      _Pragma("unroll") for (int i = 0; i < 2; i++) _Pragma("unroll") for (int j = 0; j < 4; j++)         \
        acc[p][i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[p][i][j], 0, 0, 0);           \
    }                                                                                                     \
  }
__global__ __attribute__((amdgpu_flat_work_group_size(256, 256), amdgpu_waves_per_eu(2, 2))) void probe2(double *out, const double *src, int nst, long long stride) {
  extern __shared__ double lds[];
  double *As = lds, *Bs = lds + 2 * 8 * LDA;
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, wm = w & 1, wn = w >> 1;
  for (int i = tid; i < 2 * 8 * LDA + 2 * 8 * LDB; i += 256) lds[i] = 1.0 + 1e-9 * i;
  __syncthreads();
  v4d acc[2][2][4];
  for (int p = 0; p < 2; p++)
    for (int i = 0; i < 2; i++)
      for (int j = 0; j < 4; j++) acc[p][i][j] = (v4d){0, 0, 0, 0};
  const int arow = tid >> 5, ac2 = tid & 31, brow = tid >> 6, bc2 = tid & 63;
  const double *g = src + (long long)blockIdx.x * 4096 + tid * 2;
  d2 xa0 = {1, 2}, xa1 = {3, 4}, xb0 = {5, 6}, xb1 = {7, 8}, xb2 = {9, 10}, xb3 = {11, 12};
  d2 ya0 = xa0, ya1 = xa1, yb0 = xb0, yb1 = xb1, yb2 = xb2, yb3 = xb3;
  for (int s = 0; s < nst; s += 2) {
    PROBE_STAGE(xa0, xa1, xb0, xb1, xb2, xb3)
    s++;
    PROBE_STAGE(ya0, ya1, yb0, yb1, yb2, yb3)
    s--;
  }
  double sum = xa0.x + xb3.y + ya0.x + yb3.y;
  for (int p = 0; p < 2; p++)
    for (int i = 0; i < 2; i++)
      for (int j = 0; j < 4; j++) sum += acc[p][i][j][0] + acc[p][i][j][3];
  out[(long long)blockIdx.x * 256 + tid] = sum;
}


// V8: the operands go global -> LDS directly (global_load_lds_dwordx4, no staging registers) into a ring of three
// stages, two stages in flight; ONE raw barrier per stage behind a counted vmcnt.  Unpadded LDS rows: the odd rows'
// 128-byte halves are swapped through the per-lane SOURCE address (the LDS image of an LDS-DMA is lane-linear).
typedef __attribute__((address_space(3))) void *lds_vp;
template <int HIT, int MODE = 0>
__global__ __attribute__((amdgpu_flat_work_group_size(256, 256), amdgpu_waves_per_eu(2, 2))) void probe_glds(double *out, const double *src, int nst, long long stride) {
  extern __shared__ double lds[];  // [3][A: 8 kk x (2 par x 64) | B: 16 rows x 128]
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, wm = w & 1, wn = w >> 1;
  v4d acc[2][2][4];
  for (int p = 0; p < 2; p++)
    for (int i = 0; i < 2; i++)
      for (int j = 0; j < 4; j++) acc[p][i][j] = (v4d){0, 0, 0, 0};
  const double *g = src + (long long)blockIdx.x * 4096;
#define GLDS_ISSUE(s_)                                                                                     \
  {                                                                                                        \
    double *st = lds + ((s_) % 3) * 3072;                                                                  \
    const double *q = g + (HIT ? 0 : (long long)(s_) * stride);                                            \
    _Pragma("unroll") for (int i = 0; i < 2; i++) {                                                        \
      const int kk = 2 * w + i;                                                                            \
      __builtin_amdgcn_global_load_lds(q + kk * 128 + ((l ^ ((kk & 1) << 3)) << 1), (lds_vp)(st + kk * 128), 16, 0, 0); \
    }                                                                                                      \
    _Pragma("unroll") for (int i = 0; i < 4; i++) {                                                        \
      const int r = 4 * w + i;                                                                             \
      __builtin_amdgcn_global_load_lds(q + 1024 + r * 128 + ((l ^ ((r & 1) << 3)) << 1), (lds_vp)(st + 1024 + r * 128), 16, 0, 0); \
    }                                                                                                      \
  }
  GLDS_ISSUE(0);
  if (nst > 1) GLDS_ISSUE(1);
  for (int s = 0; s < nst; s++) {
    if (MODE == 0 || MODE == 3) {
      if (s + 1 < nst)
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (MODE != 1 && MODE != 3 && s + 2 < nst) GLDS_ISSUE(s + 2);
    const double *As = lds + (s % 3) * 3072, *Bs = As + 1024;
#pragma unroll
    for (int p = 0; p < 2; p++)
#pragma unroll
      for (int ks = 0; ks < 2; ks++) {
        const int kk = 4 * ks + (l >> 4), sw = (kk & 1) << 4;
        double a[2], b[4];
#pragma unroll
        for (int i = 0; i < 2; i++) a[i] = As[kk * 128 + p * 64 + ((wm * 32 + i * 16 + (l & 15)) ^ sw)];
#pragma unroll
        for (int j = 0; j < 4; j++) b[j] = Bs[(p * 8 + kk) * 128 + ((wn * 64 + j * 16 + (l & 15)) ^ sw)];
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
          for (int j = 0; j < 4; j++) acc[p][i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[p][i][j], 0, 0, 0);
        if (MODE == 3 && p == 0 && ks == 1 && s + 2 < nst) GLDS_ISSUE(s + 2);  // DMA issued mid-stage
      }
  }
#undef GLDS_ISSUE
  double sum = 0;
  for (int p = 0; p < 2; p++)
    for (int i = 0; i < 2; i++)
      for (int j = 0; j < 4; j++) sum += acc[p][i][j][0] + acc[p][i][j][3];
  out[(long long)blockIdx.x * 256 + tid] = sum;
}

// V14-V17 "ping-pong": a 512-thread workgroup = two 256-thread halves, each with its own tile, LDS image and
// prefetch registers (one wave of each half per SIMD).  The halves alternate phases separated by ONE workgroup
// barrier: in phase p half (p & 1) runs the 32 MFMAs of its stage while the other half waits for its prefetched
// operands, writes them to its LDS image and requests the next ones -- matrix work beside memory work by
// construction, 2 barriers per pair of stages instead of 4.  HEAVY: the non-MFMA phase of k_leg_dir (12 loads,
// 8 add/sub pairs, 12 LDS writes instead of 6 loads and 6 writes).
template <int HIT, int HEAVY>
__global__ __attribute__((amdgpu_flat_work_group_size(512, 512), amdgpu_waves_per_eu(2, 2))) void probe_pp(double *out, const double *src, int nst, long long stride) {
  extern __shared__ double lds[];
  const int half = threadIdx.x >> 8, tid = threadIdx.x & 255, w = tid >> 6, l = tid & 63, wm = w & 1, wn = w >> 1;
  const int area = 2 * 8 * LDA + (HEAVY ? 4 : 2) * 8 * LDB;
  double *As = lds + half * area, *Bs = As + 2 * 8 * LDA, *Bx = Bs + 2 * 8 * LDB;
  for (int i = tid; i < area; i += 256) As[i] = 1.0 + 1e-9 * i;
  __syncthreads();
  v4d acc[2][2][4];
  for (int p = 0; p < 2; p++)
    for (int i = 0; i < 2; i++)
      for (int j = 0; j < 4; j++) acc[p][i][j] = (v4d){0, 0, 0, 0};
  const int arow = tid >> 5, ac2 = tid & 31, brow = tid >> 6, bc2 = tid & 63;
  const double *g = src + (long long)(blockIdx.x * 2 + half) * 4096 + tid * 2;
  d2 ra0 = {1, 2}, ra1 = {3, 4}, rb0 = {5, 6}, rb1 = {7, 8}, rb2 = {9, 10}, rb3 = {11, 12};
  d2 rc0 = ra0, rc1 = ra1, rc2 = rb0, rc3 = rb1, rc4 = rb2, rc5 = rb3;
#define PP_LOAD(s_)                                                                                         \
  {                                                                                                         \
    const double *q = g + (HIT ? 0 : (long long)(s_) * stride);                                             \
    ra0 = *(const d2 *)q, ra1 = *(const d2 *)(q + 512), rb0 = *(const d2 *)(q + 1024), rb1 = *(const d2 *)(q + 1536); \
    rb2 = *(const d2 *)(q + 2048), rb3 = *(const d2 *)(q + 2560);                                           \
    if (HEAVY) {                                                                                            \
      const double *q2 = q + stride / 2 + 3072;                                                             \
      rc0 = *(const d2 *)q2, rc1 = *(const d2 *)(q2 + 512), rc2 = *(const d2 *)(q2 + 1024), rc3 = *(const d2 *)(q2 + 1536); \
      rc4 = *(const d2 *)(q2 + 2048), rc5 = *(const d2 *)(q2 + 2560);                                       \
    }                                                                                                       \
  }
#define PP_STORE()                                                                                          \
  {                                                                                                         \
    *(d2 *)(As + (0 * 8 + arow) * LDA + 2 * ac2) = ra0;                                                     \
    *(d2 *)(As + (1 * 8 + arow) * LDA + 2 * ac2) = ra1;                                                     \
    if (!HEAVY) {                                                                                           \
      *(d2 *)(Bs + (((brow + 0) & 1) * 8 + ((brow + 0) >> 1)) * LDB + 2 * bc2) = rb0;                       \
      *(d2 *)(Bs + (((brow + 4) & 1) * 8 + ((brow + 4) >> 1)) * LDB + 2 * bc2) = rb1;                       \
      *(d2 *)(Bs + (((brow + 8) & 1) * 8 + ((brow + 8) >> 1)) * LDB + 2 * bc2) = rb2;                       \
      *(d2 *)(Bs + (((brow + 12) & 1) * 8 + ((brow + 12) >> 1)) * LDB + 2 * bc2) = rb3;                     \
    } else {                                                                                                \
      *(d2 *)(Bx + (0 * 8 + arow) * LDA + 2 * ac2) = rc0;                                                   \
      *(d2 *)(Bx + (1 * 8 + arow) * LDA + 2 * ac2) = rc1;                                                   \
      *(d2 *)(Bs + (0 * 8 + brow) * LDB + 2 * bc2) = rb0 + rc2;                                             \
      *(d2 *)(Bs + (1 * 8 + brow) * LDB + 2 * bc2) = rb0 - rc2;                                             \
      *(d2 *)(Bs + (0 * 8 + brow + 4) * LDB + 2 * bc2) = rb1 + rc3;                                         \
      *(d2 *)(Bs + (1 * 8 + brow + 4) * LDB + 2 * bc2) = rb1 - rc3;                                         \
      *(d2 *)(Bx + 2 * 8 * LDA + (0 * 8 + brow) * LDB + 2 * bc2) = rb2 + rc4;                               \
      *(d2 *)(Bx + 2 * 8 * LDA + (1 * 8 + brow) * LDB + 2 * bc2) = rb2 - rc4;                               \
      *(d2 *)(Bx + 2 * 8 * LDA + (0 * 8 + brow + 4) * LDB + 2 * bc2) = rb3 + rc5;                           \
      *(d2 *)(Bx + 2 * 8 * LDA + (1 * 8 + brow + 4) * LDB + 2 * bc2) = rb3 - rc5;                           \
    }                                                                                                       \
  }
  PP_LOAD(0);
  if (half == 0) {
    PP_STORE();
    PP_LOAD(1);
  }
  __syncthreads();
  for (int p = 0; p < 2 * nst; p++) {
    const int s = p >> 1;
    if ((p & 1) == half) {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int pr = 0; pr < 2; pr++)
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
          const int kk = 4 * ks + (l >> 4);
          double a[2], b[4];
#pragma unroll
          for (int i = 0; i < 2; i++) a[i] = As[(pr * 8 + kk) * LDA + wm * 32 + i * 16 + (l & 15)];
#pragma unroll
          for (int j = 0; j < 4; j++) b[j] = Bs[(pr * 8 + kk) * LDB + wn * 64 + j * 16 + (l & 15)];
#pragma unroll
          for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[pr][i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[pr][i][j], 0, 0, 0);
        }
      __builtin_amdgcn_s_setprio(0);
    } else {
      PP_STORE();
      PP_LOAD(s + 2);
    }
    __syncthreads();
  }
#undef PP_LOAD
#undef PP_STORE
  double sum = ra0.x + rb3.y + rc0.x + rc5.y;
  for (int p = 0; p < 2; p++)
    for (int i = 0; i < 2; i++)
      for (int j = 0; j < 4; j++) sum += acc[p][i][j][0] + acc[p][i][j][3];
  out[(long long)blockIdx.x * 512 + threadIdx.x] = sum;
}

// V20-V22 "four waves per SIMD": the same 64 x 128 x two-parity tile and 8-row stages on a 512-thread workgroup (8 waves, wave
// (wm, wn) owns 32 latitudes x 32 columns of both parities: 8 accumulator fragments = 64 registers), two workgroups per CU =
// 16 waves, 128 registers each.  More waves to fill the matrix pipe while others sit in barriers / LDS phases; per MFMA one
// ds_read_b64 instead of 0.75.  MODE 0: MFMAs only, 1: full loop with HBM-jumping loads, 2: full loop with cache hits.
template <int MODE>
__global__ __attribute__((amdgpu_flat_work_group_size(512, 512), amdgpu_waves_per_eu(4, 4))) void probe_w4(double *out, const double *src, int nst, long long stride) {
  extern __shared__ double lds[];
  double *As = lds, *Bs = lds + 2 * 8 * LDA;
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, wm = w & 1, wn = w >> 1;
  for (int i = tid; i < 2 * 8 * LDA + 2 * 8 * LDB; i += 512) lds[i] = 1.0 + 1e-9 * i;
  __syncthreads();
  v4d acc[2][2][2];
  for (int p = 0; p < 2; p++)
    for (int i = 0; i < 2; i++)
      for (int j = 0; j < 2; j++) acc[p][i][j] = (v4d){0, 0, 0, 0};
  // loader: A = 2 parities x 8 k x 64 latitudes = 512 x 16 B (one per thread); B = 16 rows x 128 columns = 1024 x 16 B (two per thread)
  const int apar = tid >> 8, arow = (tid >> 5) & 7, ac2 = tid & 31, brow = tid >> 6, bc2 = tid & 63;
  const double *g = src + (long long)blockIdx.x * 4096 + tid * 2;
  d2 ra0 = {1, 2}, rb0 = {5, 6}, rb1 = {7, 8};
  double a0 = 1.0 + 1e-9 * l, b0 = 1.0 - 1e-9 * l;
  for (int s = 0; s < nst; s++) {
    if (MODE >= 1) {
      if (s > 0) __syncthreads();
      *(d2 *)(As + (apar * 8 + arow) * LDA + 2 * ac2) = ra0;
      *(d2 *)(Bs + (((brow + 0) & 1) * 8 + ((brow + 0) >> 1)) * LDB + 2 * bc2) = rb0;
      *(d2 *)(Bs + (((brow + 8) & 1) * 8 + ((brow + 8) >> 1)) * LDB + 2 * bc2) = rb1;
      __syncthreads();
      const double *q = g + (MODE == 2 ? 0 : (long long)s * stride);
      ra0 = *(const d2 *)q, rb0 = *(const d2 *)(q + 1024), rb1 = *(const d2 *)(q + 2048);
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int p = 0; p < 2; p++)
#pragma unroll
      for (int ks = 0; ks < 2; ks++) {
        const int kk = 4 * ks + (l >> 4);
        double a[2], b[2];
        if (MODE >= 1) {
#pragma unroll
          for (int i = 0; i < 2; i++) a[i] = As[(p * 8 + kk) * LDA + wm * 32 + i * 16 + (l & 15)];
#pragma unroll
          for (int j = 0; j < 2; j++) b[j] = Bs[(p * 8 + kk) * LDB + wn * 32 + j * 16 + (l & 15)];
        } else {
          a[0] = a[1] = a0;
          b[0] = b[1] = b0;
        }
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
          for (int j = 0; j < 2; j++) acc[p][i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[p][i][j], 0, 0, 0);
      }
    __builtin_amdgcn_s_setprio(0);
  }
  double sum = ra0.x + rb0.y + rb1.x;
  for (int p = 0; p < 2; p++)
    for (int i = 0; i < 2; i++)
      for (int j = 0; j < 2; j++) sum += acc[p][i][j][0] + acc[p][i][j][3];
  out[(long long)blockIdx.x * 512 + tid] = sum;
}

// V30-V35 "one big workgroup per CU": 256 threads = one wave per SIMD with the whole register file (256 arch + 256 accumulator
// registers), tile 128 latitudes x 128 columns x two parities (wave (wm, wn): 64 x 64 of both parities = 32 accumulator fragments),
// stages of 8 KS rows per parity = 64 KS MFMAs per wave between barriers.  16 flop per byte fetched instead of 10.9; nothing else
// on the SIMD to cover a gap, so the loop itself has to be tight.  MODE 0: MFMAs only, 1: HBM-jumping loads, 2: cache hits.
#define LDB2 144
template <int MODE, int KS>
__global__ __attribute__((amdgpu_flat_work_group_size(256, 256), amdgpu_waves_per_eu(1, 1))) void probe_big(double *out, const double *src, int nst, long long stride) {
  extern __shared__ double lds[];
  constexpr int KR = 8 * KS;  // rows per parity and stage
  double *As = lds, *Bs = lds + 2 * KR * LDB2;
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, wm = w & 1, wn = w >> 1;
  for (int i = tid; i < 4 * KR * LDB2; i += 256) lds[i] = 1.0 + 1e-9 * i;
  __syncthreads();
  v4d acc[2][4][4];
  for (int p = 0; p < 2; p++)
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 4; j++) acc[p][i][j] = (v4d){0, 0, 0, 0};
  const int row = tid >> 6, c2 = tid & 63;  // loader: rows row + 4 i of A (2 KR rows of 128) and of B (2 KR rows of 128)
  const double *g = src + (long long)blockIdx.x * 8192 + tid * 2;
  d2 ra[4 * KS], rb[4 * KS];
  for (int i = 0; i < 4 * KS; i++) ra[i] = (d2){1.0, 2.0}, rb[i] = (d2){3.0, 4.0};
  double a0 = 1.0 + 1e-9 * l, b0 = 1.0 - 1e-9 * l;
  for (int s = 0; s < nst; s++) {
    if (MODE >= 1) {
      if (s > 0) __syncthreads();
#pragma unroll
      for (int i = 0; i < 4 * KS; i++) {
        *(d2 *)(As + (row + 4 * i) * LDB2 + 2 * c2) = ra[i];
        *(d2 *)(Bs + (row + 4 * i) * LDB2 + 2 * c2) = rb[i];
      }
      __syncthreads();
      const double *q = g + (MODE == 2 ? 0 : (long long)s * stride);
#pragma unroll
      for (int i = 0; i < 4 * KS; i++) ra[i] = *(const d2 *)(q + i * 512), rb[i] = *(const d2 *)(q + (4 * KS + i) * 512);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int p = 0; p < 2; p++)
#pragma unroll
      for (int ks = 0; ks < 2 * KS; ks++) {
        const int kk = 4 * ks + (l >> 4);
        double a[4], b[4];
        if (MODE >= 1) {
#pragma unroll
          for (int i = 0; i < 4; i++) a[i] = As[(p * KR + kk) * LDB2 + wm * 64 + i * 16 + (l & 15)];
#pragma unroll
          for (int j = 0; j < 4; j++) b[j] = Bs[(p * KR + kk) * LDB2 + wn * 64 + j * 16 + (l & 15)];
        } else {
          a[0] = a[1] = a[2] = a[3] = a0;
          b[0] = b[1] = b[2] = b[3] = b0;
        }
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
          for (int j = 0; j < 4; j++) acc[p][i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[p][i][j], 0, 0, 0);
      }
  }
  double sum = ra[0].x + rb[0].y;
  for (int p = 0; p < 2; p++)
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 4; j++) sum += acc[p][i][j][0] + acc[p][i][j][3];
  out[(long long)blockIdx.x * 256 + tid] = sum;
}

// V40-V43 fp32: the fp32 library's loop on v_mfma_f32_16x16x4_f32 (half the cycles per instruction) with 16-row stages (round 4), and
// the same flops on v_mfma_f32_32x32x2_f32 (half the instructions).  MODE 0: MFMAs only (16x16x4), 1: full loop with cache hits
// (16x16x4), 2: MFMAs only (32x32x2), 3: MFMAs only 16x16x4 with four waves per SIMD is V20's business -- not repeated here.
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ __attribute__((amdgpu_flat_work_group_size(256, 256), amdgpu_waves_per_eu(2, 2))) void probe_f32(float *out, const float *src, int nst, long long stride) {
  extern __shared__ float ldsf[];
  constexpr int KR = 16;
  float *As = ldsf, *Bs = ldsf + 2 * KR * LDA;
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, wm = w & 1, wn = w >> 1;
  for (int i = tid; i < 2 * KR * LDA + 2 * KR * LDB; i += 256) ldsf[i] = 1.0f + 1e-5f * (i & 1023);
  __syncthreads();
  float sum = 0.f;
  if (MODE == 4) {  // full loop (cache hits) on v_mfma_f32_32x32x2_f32: 32 instead of 64 MFMAs per stage, 64 cycles each
    v16f acc[2][2];
    for (int p = 0; p < 2; p++)
      for (int j = 0; j < 2; j++)
        for (int e = 0; e < 16; e++) acc[p][j][e] = 0.f;
    const int arow = tid >> 4, ac = (tid & 15) * 4, brow = tid >> 5, bc = (tid & 31) * 4;
    const float *g = src + (long long)blockIdx.x * 8192 + tid * 4;
    v4f ra0 = {1, 2, 3, 4}, ra1 = ra0, rb0 = ra0, rb1 = ra0, rb2 = ra0, rb3 = ra0;
    for (int s = 0; s < nst; s++) {
      if (s > 0) __syncthreads();
      *(v4f *)(As + (0 * KR + arow) * LDA + ac) = ra0;
      *(v4f *)(As + (1 * KR + arow) * LDA + ac) = ra1;
      *(v4f *)(Bs + (((brow + 0) & 1) * KR + ((brow + 0) >> 1)) * LDB + bc) = rb0;
      *(v4f *)(Bs + (((brow + 8) & 1) * KR + ((brow + 8) >> 1)) * LDB + bc) = rb1;
      *(v4f *)(Bs + (((brow + 16) & 1) * KR + ((brow + 16) >> 1)) * LDB + bc) = rb2;
      *(v4f *)(Bs + (((brow + 24) & 1) * KR + ((brow + 24) >> 1)) * LDB + bc) = rb3;
      __syncthreads();
      ra0 = *(const v4f *)g, ra1 = *(const v4f *)(g + 1024), rb0 = *(const v4f *)(g + 2048), rb1 = *(const v4f *)(g + 3072);
      rb2 = *(const v4f *)(g + 4096), rb3 = *(const v4f *)(g + 5120);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int p = 0; p < 2; p++)
#pragma unroll
        for (int ks = 0; ks < KR / 2; ks++) {
          const int kk = 2 * ks + (l >> 5);
          const float a = As[(p * KR + kk) * LDA + wm * 32 + (l & 31)];
          float b[2];
#pragma unroll
          for (int j = 0; j < 2; j++) b[j] = Bs[(p * KR + kk) * LDB + wn * 64 + j * 32 + (l & 31)];
#pragma unroll
          for (int j = 0; j < 2; j++) acc[p][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[j], acc[p][j], 0, 0, 0);
        }
      __builtin_amdgcn_s_setprio(0);
    }
    sum = ra0[0] + rb3[1];
    for (int p = 0; p < 2; p++)
      for (int j = 0; j < 2; j++) sum += acc[p][j][0] + acc[p][j][15];
  } else if (MODE == 2) {
    v16f acc[2][2];
    for (int p = 0; p < 2; p++)
      for (int j = 0; j < 2; j++)
        for (int e = 0; e < 16; e++) acc[p][j][e] = 0.f;
    float a0 = 1.0f + 1e-5f * l, b0 = 1.0f - 1e-5f * l;
    for (int s = 0; s < nst; s++)
#pragma unroll
      for (int p = 0; p < 2; p++)
#pragma unroll
        for (int ks = 0; ks < KR / 2; ks++)  // k = 2 per instruction
#pragma unroll
          for (int j = 0; j < 2; j++) acc[p][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[p][j], 0, 0, 0);
    for (int p = 0; p < 2; p++)
      for (int j = 0; j < 2; j++) sum += acc[p][j][0] + acc[p][j][15];
  } else {
    v4f acc[2][2][4];
    for (int p = 0; p < 2; p++)
      for (int i = 0; i < 2; i++)
        for (int j = 0; j < 4; j++) acc[p][i][j] = (v4f){0, 0, 0, 0};
    const int arow = tid >> 4, ac = (tid & 15) * 4, brow = tid >> 5, bc = (tid & 31) * 4;
    const float *g = src + (long long)blockIdx.x * 8192 + tid * 4;
    v4f ra0 = {1, 2, 3, 4}, ra1 = ra0, rb0 = ra0, rb1 = ra0, rb2 = ra0, rb3 = ra0;
    float a0 = 1.0f + 1e-5f * l, b0 = 1.0f - 1e-5f * l;
    for (int s = 0; s < nst; s++) {
      if (MODE == 1 || MODE == 3) {
        if (s > 0) __syncthreads();
        *(v4f *)(As + (0 * KR + arow) * LDA + ac) = ra0;
        *(v4f *)(As + (1 * KR + arow) * LDA + ac) = ra1;
        *(v4f *)(Bs + (((brow + 0) & 1) * KR + ((brow + 0) >> 1)) * LDB + bc) = rb0;
        *(v4f *)(Bs + (((brow + 8) & 1) * KR + ((brow + 8) >> 1)) * LDB + bc) = rb1;
        *(v4f *)(Bs + (((brow + 16) & 1) * KR + ((brow + 16) >> 1)) * LDB + bc) = rb2;
        *(v4f *)(Bs + (((brow + 24) & 1) * KR + ((brow + 24) >> 1)) * LDB + bc) = rb3;
        __syncthreads();
        ra0 = *(const v4f *)g, ra1 = *(const v4f *)(g + 1024), rb0 = *(const v4f *)(g + 2048), rb1 = *(const v4f *)(g + 3072);
        rb2 = *(const v4f *)(g + 4096), rb3 = *(const v4f *)(g + 5120);
        __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int p = 0; p < 2; p++)
#pragma unroll
        for (int ks = 0; ks < KR / 4; ks++) {
          const int kk = 4 * ks + (l >> 4);
          float a[2], b[4];
          if (MODE == 3) {
            // two k-steps per LDS read: the operands of k-steps ks (even) and ks + 1 sit side by side (float2), so a stage needs half the
            // ds_read instructions (the probe only measures the instruction mix: the values are whatever the image holds)
            typedef float f2v __attribute__((ext_vector_type(2)));
            if (ks & 1) continue;
            f2v a2[2], b2[4];
            const int kq = 2 * ks + (l >> 4);  // row of the pair image: KR / 2 rows of 2 x the width
#pragma unroll
            for (int i = 0; i < 2; i++) a2[i] = *(const f2v *)(As + (p * KR + kq) * LDA + 2 * (wm * 16 + i * 8) + 2 * (l & 7) + 32 * ((l >> 3) & 1));
#pragma unroll
            for (int j = 0; j < 4; j++) b2[j] = *(const f2v *)(Bs + (p * KR + kq) * LDB + 2 * (wn * 32 + j * 8) + 2 * (l & 7) + 64 * ((l >> 3) & 1));
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
              for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[p][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[i][h], b2[j][h], acc[p][i][j], 0, 0, 0);
            continue;
          }
          if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 2; i++) a[i] = As[(p * KR + kk) * LDA + wm * 32 + i * 16 + (l & 15)];
#pragma unroll
            for (int j = 0; j < 4; j++) b[j] = Bs[(p * KR + kk) * LDB + wn * 64 + j * 16 + (l & 15)];
          } else {
            a[0] = a[1] = a0;
            b[0] = b[1] = b[2] = b[3] = b0;
          }
#pragma unroll
          for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[p][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[p][i][j], 0, 0, 0);
        }
      __builtin_amdgcn_s_setprio(0);
    }
    sum = ra0[0] + rb3[1];
    for (int p = 0; p < 2; p++)
      for (int i = 0; i < 2; i++)
        for (int j = 0; j < 4; j++) sum += acc[p][i][j][0] + acc[p][i][j][3];
  }
  out[(long long)blockIdx.x * 256 + tid] = sum;
}

template <int V>
static void run(double *out, const double *src, const char *what, int nst = 80) {
  const int nblk = 256 * 2 * 8;  // 16 tiles per CU-slot; nst = 80: K = 640 n-pairs
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 4; rep++) {
    hipEventRecord(e0, 0);
    if (V >= 40 && V <= 44) {
      // fp32: nst counts 16-row stages of 64 MFMAs (16x16x4) = the flops of two fp64 stages: run nst / 2 of them, the flop count below holds
      hipLaunchKernelGGL((probe_f32<V - 40>), dim3(nblk), dim3(256), (2 * 16 * LDA + 2 * 16 * LDB) * 4, 0, (float *)out, (const float *)src, nst / 2, (long long)0);
    } else if (V >= 30 && V <= 35) {
      // nblk / 2 workgroups of twice the tile; nst counts 8-row stages, so KS = 2 runs nst / 2 of them: the flop count below holds
      constexpr int KS = V >= 33 ? 2 : 1;
      hipLaunchKernelGGL((probe_big<(V - 30) % 3, KS>), dim3(nblk / 2), dim3(256), 4 * 8 * KS * LDB2 * 8, 0, out, src, nst / KS, (long long)8192 * (nblk / 2) / 64 * KS);
    } else if (V >= 20 && V <= 22)
      hipLaunchKernelGGL((probe_w4<V - 20>), dim3(nblk), dim3(512), (2 * 8 * LDA + 2 * 8 * LDB) * 8, 0, out, src, nst, (long long)4096 * nblk / 64);
    else if (V >= 14 && V <= 17)
      hipLaunchKernelGGL((probe_pp<(V & 1), (V >= 16)>), dim3(nblk / 2), dim3(512), 2 * (2 * 8 * LDA + (V >= 16 ? 4 : 2) * 8 * LDB) * 8, 0, out, src, nst,
                         (long long)4096 * nblk / 64);
    else if (V == 8 || V == 9)
      hipLaunchKernelGGL(probe_glds<(V == 9)>, dim3(nblk), dim3(256), 3 * 3072 * 8, 0, out, src, nst, (long long)4096 * nblk / 64);
    else if (V >= 10 && V <= 12)
      hipLaunchKernelGGL((probe_glds<1, V - 9>), dim3(nblk), dim3(256), 3 * 3072 * 8, 0, out, src, nst, (long long)4096 * nblk / 64);
    else if (V == 13)
      hipLaunchKernelGGL((probe_glds<0, 3>), dim3(nblk), dim3(256), 3 * 3072 * 8, 0, out, src, nst, (long long)4096 * nblk / 64);
    else if (V == 7)
      hipLaunchKernelGGL(probe2, dim3(nblk), dim3(256), (2 * 8 * LDA + 2 * 8 * LDB) * 8, 0, out, src, nst, (long long)4096 * nblk / 64);
    else
      hipLaunchKernelGGL(probe<V>, dim3(nblk), dim3(256), (2 * 8 * LDA + 2 * 8 * LDB) * 8, 0, out, src, nst, (long long)4096 * nblk / 64);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  double flops = (double)nblk * 4 * nst * 32 * 2048.0;
  if (V >= 40 && V <= 44)
    printf("V%d nst %2d %-44s %6.1f TFLOP/s  (%.1f %% of 157.3, fp32)\n", V, nst, what, flops / best / 1e9, 100 * flops / best / 1e9 / 157.3);
  else
    printf("V%d nst %2d %-44s %6.1f TFLOP/s  (%.1f %% of 78.6)\n", V, nst, what, flops / best / 1e9, 100 * flops / best / 1e9 / 78.6);
}

int main() {
  double *out, *src;
  hipMalloc((void **)&out, (size_t)4096 * 512 * 8);
  hipMalloc((void **)&src, (size_t)1 << 30);
  hipMemset(src, 0, (size_t)1 << 30);
  if (getenv("PROBE_RANDOM")) {  // operand tiles with random mantissas: the matrix cores' power draw is data dependent
    int one = 1;
    hipMemcpyToSymbol(HIP_SYMBOL(g_random_operands), &one, sizeof(int));
    printf("random operand values in LDS (V1-V3 read them; V4+ overwrite them with the zero-filled source)\n");
  }
  if (getenv("PROBE_ONE_PARITY")) {  // round 6: one parity per wave (V50 / V51) against the kernels' two-parity fragments (V4 / V6), twice each
    for (int rep = 0; rep < 2; rep++) {
      run<4>(out, src, "two parities per wave, HBM-jumping loads");
      run<50>(out, src, "ONE parity per wave, HBM-jumping loads");
      run<6>(out, src, "two parities per wave, cache hits");
      run<51>(out, src, "ONE parity per wave, cache hits");
      run<4>(out, src, "two parities per wave, short tiles", 20);
      run<50>(out, src, "ONE parity per wave, short tiles", 20);
    }
    return 0;
  }
  if (getenv("PROBE_ONLY_W4")) {
    run<0>(out, src, "MFMAs only");
    run<4>(out, src, "+ global prefetch of the next stage");
    run<6>(out, src, "V4 with the same lines every stage (hits)");
    run<23>(out, src, "V4 + touch of stage s+2 behind the loads of s+1");
    run<30>(out, src, "big tile, 1 wave/SIMD: MFMAs only");
    run<31>(out, src, "big tile, 1 wave/SIMD: full loop, HBM-jumping loads");
    run<32>(out, src, "big tile, 1 wave/SIMD: full loop, cache hits");
    run<34>(out, src, "big tile, 16-row stages: HBM-jumping loads");
    run<35>(out, src, "big tile, 16-row stages: cache hits");
    run<40>(out, src, "fp32 16x16x4: MFMAs only");
    run<41>(out, src, "fp32 16x16x4: full loop, 16-row stages, cache hits");
    run<42>(out, src, "fp32 32x32x2: MFMAs only");
    run<43>(out, src, "fp32 16x16x4: full loop, two k-steps per LDS read");
    run<44>(out, src, "fp32 32x32x2: full loop, 16-row stages, cache hits");
    run<20>(out, src, "four waves per SIMD: MFMAs only");
    run<21>(out, src, "four waves per SIMD: full loop, HBM-jumping loads");
    run<22>(out, src, "four waves per SIMD: full loop, cache hits");
    run<4>(out, src, "V4 short tiles", 20);
    run<21>(out, src, "V21 short tiles", 20);
    return 0;
  }
  run<0>(out, src, "MFMAs only");
  run<1>(out, src, "+ LDS fragment reads");
  run<2>(out, src, "+ two barriers per stage");
  run<3>(out, src, "+ LDS tile writes");
  run<4>(out, src, "+ global prefetch of the next stage");
  run<5>(out, src, "+ global prefetch two stages ahead");
  run<6>(out, src, "V4 with the same lines every stage (hits)");
  run<7>(out, src, "V4 two stages ahead, loop unrolled by two");
  run<8>(out, src, "LDS-DMA ring of 3, one barrier per stage");
  run<9>(out, src, "V8 with the same lines every stage (hits)");
  run<10>(out, src, "V9 without the DMA (barrier + MFMA loop only)");
  run<11>(out, src, "V9 with DMA but no vmcnt wait");
  run<12>(out, src, "V9 with the DMA issued mid-stage");
  run<13>(out, src, "V8 with the DMA issued mid-stage");
  run<14>(out, src, "ping-pong halves, one barrier per phase");
  run<15>(out, src, "V14 with the same lines every stage (hits)");
  run<16>(out, src, "V14 with k_leg_dir's non-MFMA phase");
  run<17>(out, src, "V16 with the same lines every stage (hits)");
  run<4>(out, src, "V4 short tiles", 20);
  run<14>(out, src, "V14 short tiles", 20);
  run<16>(out, src, "V16 short tiles", 20);
  return 0;
}
