// emi_kernels_body.h -- kernel bodies, included once per precision inside namespace emi_f64 /
// emi_f32 by emi_kernels.h with the traits macros EMI_REAL, EMI_REAL2, EMI_ACC4, EMI_MFMA, EMI_ACC_ROW set.
// (No include guard on purpose.)
typedef EMI_REAL real_t;
typedef EMI_REAL2 real2;
typedef EMI_ACC4 acc4;
#define emi_mfma_f64_16x16x4 EMI_MFMA

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------
EMI_DEVFN real2 mk2(real_t x, real_t y) {
  real2 r;
  r.x = x;
  r.y = y;
  return r;
}
EMI_DEVFN real2 cmul(real2 a, real2 b) { return mk2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
EMI_DEVFN real2 cmulc(real2 a, real2 b) { return mk2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }  // a*conj(b)
EMI_DEVFN real2 cadd(real2 a, real2 b) { return mk2(a.x + b.x, a.y + b.y); }
EMI_DEVFN real2 csub(real2 a, real2 b) { return mk2(a.x - b.x, a.y - b.y); }
EMI_DEVFN real2 cconj(real2 a) { return mk2(a.x, -a.y); }
EMI_DEVFN real2 cscale(real2 a, real_t s) { return mk2(a.x * s, a.y * s); }
EMI_DEVFN real2 cmuli(real2 a) { return mk2(-a.y, a.x); }  // i*a

EMI_DEVFN int upper_m(const int *pref, int nump, int t) {
  // largest ml in [0,nump) with pref[ml] <= t  (pref non-decreasing, pref[nump] > t)
  int lo = 0, hi = nump - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (pref[mid] <= t)
      lo = mid;
    else
      hi = mid - 1;
  }
  return lo;
}

// XCD-aware remap.  Workgroups are dealt round-robin over the 8 XCDs (bid % 8 labels the XCD).
// Logical tiles are cut into groups of G consecutive tiles (same Legendre panel / operand rows,
// different column tiles); group g*8+x goes to XCD x, so (i) the tiles of a group share one L2 and
// (ii) all XCDs walk through the m-ascending (longest-K-first) tile list at the same pace --
// chunking the list contiguously per XCD would hand XCD 0 all the expensive low-m tiles.
// Bijective for any grid size (the tail that does not fill 8 groups is mapped 1:1).
EMI_DEVFN long long xcd_swizzle(long long bid, long long nwg, int G) {
  const long long super = 8LL * G, nfull = (nwg / super) * super;
  if (bid >= nfull) return bid;
  const long long x = bid & 7, k = bid >> 3;
  return ((k / G) * 8 + x) * G + (k % G);
}

// ==========================================================================================
// k_prepack_inv: PRFI1B (prfi1b_mod.F90:81-115) + VDTUV (vdtuv_mod.F90:97-143) + SPNSDE
// (spnsde_mod.F90:95-114).  One thread per (packed row, field); fields fastest.
//   W[(wbase[m]+r)*ldw + 2f + c],  r = n-m in [0, wrows[m])  (rows n > N+1 are zero)
// ==========================================================================================
EMI_DEVFN real2 spec_get(const void *av, int sa, int ia, long long isp, int m) {
  const real_t *a = (const real_t *)av;
  real2 v;
  v.x = a[isp * sa + ia];
  v.y = (m == 0) ? 0.0 : a[(isp + 1) * sa + ia];
  return v;
}

EMI_KERNEL_LB(256) void k_prepack_inv(EmiGeomDev g, const SpecSrc *flds, int nfld, int nfld_pad, real_t *W, int ldw,
                              long long nrows) {
  const int N = g.nsmax;
  {
    // block -> (packed row, chunk of 256 fields): 32-bit arithmetic only (blocks of 4 rows x 64 fields, as in
    // k_postpack_dir, are 18 % slower here: the reads of the caller's arrays want the longer runs)
    const int nchunk = (nfld_pad + EMI_NTHREADS - 1) / EMI_NTHREADS;
    const long long row = EMI_BID / nchunk;
    const int f = (int)(EMI_BID - row * nchunk) * EMI_NTHREADS + EMI_TID;
    if (row >= nrows || f >= nfld_pad) return;
    real2 out = mk2(0.0, 0.0);
    if (f < nfld) {
      const int ml = g.rowm[row];
      const int m = g.mval[ml];
      int r = (int)(row - g.wbase[ml]);
      int n = m + r;
      if (n <= N + 1) {
        SpecSrc s = flds[f];
        long long isp = g.nasm0[ml] + 2LL * r;  // Re(m,n)
        if (s.kind == SPK_COPY) {
          if (n <= N) out = spec_get(s.a, s.sa, s.ia, isp, m);
        } else {
          const double *eps = g.eps + g.ebase[ml] - m;  // eps[n], n=m..N+2
          real_t zn_m1 = (real_t)(n - 1), zn_p2 = (real_t)(n + 2);
          real_t e_n = (real_t)eps[n], e_np1 = (real_t)eps[n + 1];
          if (s.kind == SPK_NSD) {
            real2 fm = (n - 1 >= m) ? spec_get(s.a, s.sa, s.ia, isp - 2, m) : mk2(0, 0);
            real2 fp = (n + 1 <= N) ? spec_get(s.a, s.sa, s.ia, isp + 2, m) : mk2(0, 0);
            out.x = -zn_m1 * e_n * fm.x + zn_p2 * e_np1 * fp.x;
            out.y = -zn_m1 * e_n * fm.y + zn_p2 * e_np1 * fp.y;
          } else {
            // a = vorticity, b = divergence.  SPK_U_AD / SPK_V_AD: the adjoint of UVTVD (uvtvdad_mod.F90) is
            // VDTUV with the inverse Laplacian replaced by -1 and the (0,0) coefficients, which UVTVD
            // never produces, read as zero
            const bool ad = s.kind >= SPK_U_AD, isu = (s.kind == SPK_U || s.kind == SPK_U_AD);
            const void *pa = isu ? s.a : s.b;  // the field entering the +-(n-1),(n+2) terms
            const void *pb = isu ? s.b : s.a;  // the field entering the i*m term
            int sa = isu ? s.sa : s.sb, ia = isu ? s.ia : s.ib;
            int sb = isu ? s.sb : s.sa, ib = isu ? s.ib : s.ia;
            real_t l_n = ad ? (real_t)-1.0 : (real_t)g.lapin[n + 1], l_nm1 = ad ? (real_t)-1.0 : (real_t)g.lapin[n],
                   l_np1 = ad ? (real_t)-1.0 : (real_t)g.lapin[n + 2];
            real2 xm = (n - 1 >= m && !(ad && n - 1 == 0)) ? spec_get(pa, sa, ia, isp - 2, m) : mk2(0, 0);
            real2 xp = (n + 1 <= N) ? spec_get(pa, sa, ia, isp + 2, m) : mk2(0, 0);
            real2 y0 = (n <= N && !(ad && n == 0)) ? spec_get(pb, sb, ib, isp, m) : mk2(0, 0);
            real_t zkm = (real_t)m;
            real_t c1 = zn_m1 * e_n * l_nm1, c2 = zn_p2 * e_np1 * l_np1;
            real_t sg = isu ? 1.0 : -1.0;
            // U = i m L_n D_n + c1 vor_{n-1} - c2 vor_{n+1};  V = i m L_n vor_n - c1 D_{n-1} + c2 D_{n+1}
            out.x = -zkm * l_n * y0.y + sg * (c1 * xm.x - c2 * xp.x);
            out.y = zkm * l_n * y0.x + sg * (c1 * xm.y - c2 * xp.y);
            if (m == 0) out.y = 0.0;
          }
        }
      }
    }
    *(real2 *)(W + row * ldw + 2 * f) = out;
  }
}

// ==========================================================================================
// k_postpack_dir: UVTVD (uvtvd_mod.F90:91-139) + UPDSP/UPDSPB (updsp_mod.F90:100-161,
// updspb_mod.F90:92-149).  One thread per (packed row with n<=N, output field).
// ==========================================================================================
EMI_KERNEL_LB(256) void k_postpack_dir(EmiGeomDev g, const SpecDst *flds, int nfld, const real_t *W, int ldw,
                               long long nrows) {
  const int N = g.nsmax;
  {
    // block -> (4 packed rows, chunk of 64 output fields)
    const int nchunk = (nfld + 63) / 64;
    const long long rb = EMI_BID / nchunk;
    const long long row = rb * (EMI_NTHREADS / 64) + (EMI_TID >> 6);
    const int f = (int)(EMI_BID - rb * nchunk) * 64 + (EMI_TID & 63);
    if (row >= nrows || f >= nfld) return;
    const int ml = g.rowm[row];
    const int m = g.mval[ml];
    int r = (int)(row - g.wbase[ml]);
    int n = m + r;
    if (n > N) return;
    SpecDst s = flds[f];
    real2 out;
    const real_t zkm = (real_t)m;
    // W[row][field], with the adjoint of an east-west derivative added: the Fourier-space factor i m / (a cos) of FSC
    // (fsc_mod.F90:163-187) has the adjoint -i m / (a cos); 1 / (a cos) was applied by k_fft_dir, -i m is applied here
    auto wget = [&](long long rw, int fld, int fld_ew) {
      real2 x = *(const real2 *)(W + rw * ldw + 2 * fld);
      if (fld_ew >= 0) {
        const real2 e = *(const real2 *)(W + rw * ldw + 2 * fld_ew);
        x.x += zkm * e.y;  // -i m (e.x + i e.y) = m e.y - i m e.x
        x.y -= zkm * e.x;
      }
      return x;
    };
    if (s.kind == SPO_COPY) {
      out = *(const real2 *)(W + row * ldw + 2 * s.src0);
    } else if (s.kind == SPO_SC_AD) {
      out = wget(row, s.src0, s.src2);
      if (s.src1 >= 0) {
        // adjoint of SPNSDE (spnsde_mod.F90:95-114: d_n = -(n-1) e_n f_{n-1} + (n+2) e_{n+1} f_{n+1}, n = m .. N+1):
        // f*_n = -n e_{n+1} d_{n+1} + (n+1) e_n d_{n-1}   (spnsdead_mod.F90:97-103)
        const double *eps = g.eps + g.ebase[ml] - m;
        const real2 dp = *(const real2 *)(W + (row + 1) * ldw + 2 * s.src1);
        const real2 dm = (n - 1 >= m) ? *(const real2 *)(W + (row - 1) * ldw + 2 * s.src1) : mk2(0, 0);
        const real_t a = -(real_t)n * (real_t)eps[n + 1], b = (real_t)(n + 1) * (real_t)eps[n];
        out.x += a * dp.x + b * dm.x;
        out.y += a * dp.y + b * dm.y;
      }
    } else {
      const double *eps = g.eps + g.ebase[ml] - m;
      // vor: x=V (i m term), y=U ; div: x=U, y=V with opposite sign on the n-terms
      const bool isvor = (s.kind == SPO_VOR || s.kind == SPO_VOR_AD);
      int fx = isvor ? s.src1 : s.src0, fxe = isvor ? s.src3 : s.src2;
      int fy = isvor ? s.src0 : s.src1, fye = isvor ? s.src2 : s.src3;
      real_t sg = isvor ? 1.0 : -1.0;
      real2 x0 = wget(row, fx, fxe);
      real2 yp = wget(row + 1, fy, fye);                                  // n+1 (<= N+1 stored)
      real2 ym = (n - 1 >= m) ? wget(row - 1, fy, fye) : mk2(0, 0);       // n-1
      real_t c1 = (real_t)n * (real_t)eps[n + 1], c2 = (real_t)(n + 1) * (real_t)eps[n];
      // vor_n = i m V_n - n e_{n+1} U_{n+1} + (n+1) e_n U_{n-1}
      // div_n = i m U_n + n e_{n+1} V_{n+1} - (n+1) e_n V_{n-1}
      out.x = -zkm * x0.y + sg * (-c1 * yp.x + c2 * ym.x);
      out.y = zkm * x0.x + sg * (-c1 * yp.y + c2 * ym.y);
      if (m == 0 && n == 0) out = mk2(0, 0);  // updsp_mod.F90:113-126
      // adjoint of VDTUV (vdtuvad_mod.F90) = -RLAPIN(n) x the UVTVD stencil
      if (s.kind >= SPO_VOR_AD) out = cscale(out, -(real_t)g.lapin[n + 1]);
      if (s.kind >= SPO_VOR_AD && s.src4 >= 0) {  // + the adjoint of the grid vorticity / divergence output (a plain copy)
        const real2 c = *(const real2 *)(W + row * ldw + 2 * s.src4);
        out.x += c.x;
        out.y += c.y;
      }
    }
    if (m == 0) out.y = 0.0;  // updspb_mod.F90:106,117
    long long isp = g.nasm0[ml] + 2LL * r;
    real_t *dst = (real_t *)s.dst;
    dst[isp * s.stride + s.idx] = out.x;
    dst[(isp + 1) * s.stride + s.idx] = out.y;
  }
}

// ==========================================================================================
// Legendre transforms on the fp64 matrix cores.
//   workgroup = 256 threads = 4 waves; MFMA v_mfma_f64_16x16x4_f64.
//   LDS tiles: As[2 parities][8 k][LG_LDA], Bs[2][8 k][LG_LDB]; the paddings (16 doubles) put
//   the two k-rows a half-wave reads with one ds_read_b64 on disjoint bank halves.
// ==========================================================================================

// Accumulator of a Legendre tile.  WIDE = false: the library precision (v_mfma_f64_16x16x4_f64 in the fp64 library,
// v_mfma_f32_16x16x4_f32 in the fp32 one).  WIDE = true (fp32 library, zonal wavenumber 0 only): the float operands
// are promoted and the products accumulated in double on the fp64 matrix cores, one rounding to float in the
// epilogue -- the reference's single-precision library computes the mean wavenumber this way ("DGEM for the mean to
// improve mass conservation", cpu/internal/ledir_mod.F90:133-171; its GPU back-end also in the inverse transform,
// gpu/internal/leinv_mod.F90:273).  The two instructions share the A / B lane maps; the accumulator rows differ.
// -DEMI_MR_STAMP -DEMI_LEG_STAMP=1|2 (experiments only; tools/leg_stamp.py): wave 0 of every k_leg_inv (1) / k_leg_dir (2) workgroup adds the
// clock ticks between the LEG_STAMP points of its stage loop to emi_mr_stamp[]: 0 end of the matrix phase -> first barrier passed, 1 operand
// waits (vmcnt) + LDS writes issued, 2 next stage's loads issued, 3 LDS writes done + second barrier passed, 4 matrix phase issued (k_leg_dir:
// + the sums and differences of the next stage's rows); [6] counts stages, [7] tiles.  s_memtime itself returns through lgkmcnt, so the points
// sit outside the matrix phase.
#if defined(EMI_MR_STAMP) && defined(EMI_LEG_STAMP) && !defined(EMI_CPU_EMU)
#define LEG_STAMP_BEGIN(k_) unsigned long long lst_acc[5] = {0}; unsigned long long lst_prev = __builtin_readcyclecounter(); const bool lst_on = (EMI_LEG_STAMP == (k_))
#define LEG_STAMP(i_) do { if (lst_on) { const unsigned long long n_ = __builtin_readcyclecounter(); lst_acc[i_] += n_ - lst_prev; lst_prev = n_; } } while (0)
#define LEG_STAMP_END(nst_) do { if (lst_on && EMI_TID == 0) { for (int i_ = 0; i_ < 5; i_++) atomicAdd(&emi_mr_stamp[i_], lst_acc[i_]); atomicAdd(&emi_mr_stamp[6], (unsigned long long)(nst_)); atomicAdd(&emi_mr_stamp[7], 1ull); } } while (0)
// tile start-up (kernel entry -> stage loop, [5]) and drain (stage loop -> last store issued [8] -> stores acknowledged [9])
#define LEG_STAMP_T0() const unsigned long long lst_t0 = __builtin_readcyclecounter()
#define LEG_STAMP_PRO(k_) do { if (EMI_LEG_STAMP == (k_) && EMI_TID == 0) atomicAdd(&emi_mr_stamp[5], __builtin_readcyclecounter() - lst_t0); } while (0)
#define LEG_STAMP_EPI0() const unsigned long long lst_e0 = __builtin_readcyclecounter()
#define LEG_STAMP_EPI(k_) do { if (EMI_LEG_STAMP == (k_)) { const unsigned long long e1_ = __builtin_readcyclecounter(); __builtin_amdgcn_s_waitcnt(0); \
    const unsigned long long e2_ = __builtin_readcyclecounter(); if (EMI_TID == 0) { atomicAdd(&emi_mr_stamp[8], e1_ - lst_e0); atomicAdd(&emi_mr_stamp[9], e2_ - lst_e0); } } } while (0)
#else
#define LEG_STAMP_BEGIN(k_) ((void)0)
#define LEG_STAMP(i_) ((void)0)
#define LEG_STAMP_END(nst_) ((void)0)
#define LEG_STAMP_T0() ((void)0)
#define LEG_STAMP_PRO(k_) ((void)0)
#define LEG_STAMP_EPI0() ((void)0)
#define LEG_STAMP_EPI(k_) ((void)0)
#endif

template <bool WIDE>
struct LegAcc {
  typedef acc4 type;
  static EMI_DEVFN type mma(real_t a, real_t b, type c) { return emi_mfma_f64_16x16x4(a, b, c); }
  static EMI_DEVFN int row(int l, int i) { return EMI_ACC_ROW(l, i); }
};
template <>
struct LegAcc<true> {
  typedef v4d type;
  static EMI_DEVFN type mma(real_t a, real_t b, type c) { return emi_mfma_f64((double)a, (double)b, c); }
  static EMI_DEVFN int row(int l, int i) { return (l >> 4) + 4 * i; }
};

// Loader vectors are 16 bytes in both libraries: a real2 in the fp64 one, FOUR floats in the fp32 one.  With the same six loads and
// six LDS stores per thread a stage of the fp32 kernels therefore carries twice the rows (inverse: 16 per parity, direct: 32
// latitudes) and twice the MFMAs, whose v_mfma_f32_16x16x4_f32 take half the cycles: the matrix phase between two barriers lasts as
// long as in the fp64 kernels instead of half as long against the same barrier / LDS-write / first-fragment gap (round 4; the 8 / 16
// row stages of the fp64 kernels ran the fp32 library at 0.66 - 0.69 of its matrix peak against 0.75 for fp64).
typedef typename std::conditional<sizeof(real_t) == 8, real2, v4f>::type lgvec;
#define LGV (16 / (int)sizeof(real_t))  // elements of a loader vector: 2 | 4
#define LG_KR (4 * LGV)                 // k_leg_inv: rows per parity and stage: 8 | 16
// k_leg_dir: loader vectors of TWO values in both precisions (16 | 8 bytes) and stages of 16 latitudes.  Until round 5 the fp32 kernel loaded 16 bytes
// as k_leg_inv does (32-latitude stages, 57 KB of LDS + row-number tables: two workgroups per CU); with 8-byte vectors a wave's load
// instruction covers ONE Fourier row, so the row numbers are scalar as in fp64 (no tables), the stage image is 29 KB and -- the
// double-precision tiles of wavenumber 0 being in a kernel of their own -- three workgroups fit a CU: 50.9 -> 46.4 ms at TCo1279 fp32
// (8-byte vectors at two waves per SIMD 49.9, at four 47.6; profiles/r5_fft_experiments.txt section 6)
typedef real2 lgdvec;
#define LGDV 2
#define LG_LS 16
EMI_DEVFN real2 lg_add(real2 a, real2 b) { return cadd(a, b); }
EMI_DEVFN real2 lg_sub(real2 a, real2 b) { return csub(a, b); }
EMI_DEVFN v4f lg_add(v4f a, v4f b) { return a + b; }
EMI_DEVFN v4f lg_sub(v4f a, v4f b) { return a - b; }

// ---- inverse: FB[lat][m][col] = sum_n P[lat,n] W[m][n][col]; north = S+A, south = S-A
// (leinv_mod.F90:92-186 DGEMM('N','N') x2, asre1b_mod.F90:83-102)
// tile: 64 latitudes x 128 columns, both parities; wave (wm, wn) owns 32 lat x 64 col.
template <bool WIDE>
EMI_DEVFN void leg_inv_tile(const EmiGeomDev &g, const int m, const int lt, const int ct, const real_t *W, int ldw, real_t *FB, int ldf) {
  typedef typename LegAcc<WIDE>::type acc_t;
  LEG_STAMP_T0();
  EMI_LDS_DECL;
  real_t *As = (real_t *)EMI_LDS_PTR;
  real_t *Bs = As + 2 * LG_KR * LG_LDA;
  const int tid = EMI_TID, w = tid >> 6, l = tid & 63;
  const int wm = w & 1, wn = w >> 1;
  const int ld = g.ldp[m];
  const int lat0 = lt * 64, col0 = ct * LG_BN;
  const int nst = g.wrows[m] / (2 * LG_KR);

  acc_t acc[2][2][4];
#pragma unroll
  for (int p = 0; p < 2; p++)
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) acc[p][i][j] = (acc_t){0.0, 0.0, 0.0, 0.0};

  // global -> register prefetch (one stage ahead): a panel row of the tile is 64 / LGV lanes wide, a packed spectral row 128 / LGV
  // lanes.  Every address is (base uniform over the wave) + (32-bit lane offset that never changes): the bases live in scalar
  // registers and advance on the scalar unit, the loads are `global_load_dwordx4 v, v_off, s[base]` -- the stage loop has no vector
  // address arithmetic (it had six 64-bit adds, which cost the matrix pipe what 0.6 matrix instructions do; emi_rt.h)
  constexpr int LA = 64 / LGV, LB = LG_BN / LGV, RPP = LG_THREADS / LB;  // RPP: W rows per pass of the workgroup (4 | 8)
  constexpr int WRA = 64 / LA, WRB = 64 / LB;                            // panel rows | W rows that one wave loads per instruction
  const int wv = emi_uniform(w);
  const int arow = tid / LA, ac = (tid % LA) * LGV;
  unsigned voA = (unsigned)(((l / LA) * ld + ac) * (int)sizeof(real_t));
  const char *uS = emi_uniform_ptr((const real_t *)g.P + g.offS[m] + (long long)(wv * WRA) * ld + lat0);
  const char *uA = emi_uniform_ptr((const real_t *)g.P + g.offA[m] + (long long)(wv * WRA) * ld + lat0);
  const int brow = tid / LB, bc = (tid % LB) * LGV;  // rows brow + RPP i, i = 0..3, of the stage
  unsigned voB = (unsigned)(((l / LB) * ldw + bc) * (int)sizeof(real_t));
  const char *uW = emi_uniform_ptr(W + ((long long)g.wbase[m] + wv * WRB) * ldw + col0);
  const long long stepA = (long long)LG_KR * ld * (long long)sizeof(real_t), stepW = 2LL * LG_KR * ldw * (long long)sizeof(real_t);
  const long long rowW4 = (long long)RPP * ldw * (long long)sizeof(real_t);
  real_t *sA0 = As + (0 * LG_KR + arow) * LG_LDA + ac;
  real_t *sA1 = As + (1 * LG_KR + arow) * LG_LDA + ac;
  // W row r of the stage -> parity r&1, k = r>>1 ; r = brow + RPP*i
  real_t *sB0 = Bs + (((brow + 0 * RPP) & 1) * LG_KR + ((brow + 0 * RPP) >> 1)) * LG_LDB + bc;
  real_t *sB1 = Bs + (((brow + 1 * RPP) & 1) * LG_KR + ((brow + 1 * RPP) >> 1)) * LG_LDB + bc;
  real_t *sB2 = Bs + (((brow + 2 * RPP) & 1) * LG_KR + ((brow + 2 * RPP) >> 1)) * LG_LDB + bc;
  real_t *sB3 = Bs + (((brow + 3 * RPP) & 1) * LG_KR + ((brow + 3 * RPP) >> 1)) * LG_LDB + bc;
  // fragment positions, one register per (parity, k step): the paired LDS reads of a step then reach their operands through the
  // 8-bit offsets of the instruction and the stage loop has no address adds (the compiler, given one base, re-derives the others
  // by nine vector adds per stage)
  int fa[2][LG_KR / 4], fb[2][LG_KR / 4];
#pragma unroll
  for (int p = 0; p < 2; p++)
#pragma unroll
    for (int ks = 0; ks < LG_KR / 4; ks++) {
      fa[p][ks] = (p * LG_KR + 4 * ks + (l >> 4)) * LG_LDA + wm * 32 + (l & 15);
      fb[p][ks] = (p * LG_KR + 4 * ks + (l >> 4)) * LG_LDB + wn * 64 + (l & 15);
      EMI_OPAQUE(fa[p][ks]);
      EMI_OPAQUE(fb[p][ks]);
    }
  lgvec ra0 = emi_ld_sv<lgvec>(uS, voA), ra1 = emi_ld_sv<lgvec>(uA, voA);
  lgvec rb0 = emi_ld_sv<lgvec>(uW, voB), rb1 = emi_ld_sv<lgvec>(uW + rowW4, voB), rb2 = emi_ld_sv<lgvec>(uW + 2 * rowW4, voB), rb3 = emi_ld_sv<lgvec>(uW + 3 * rowW4, voB);
  // Fourier rows of the tile's 64 latitudes (north | south; -1 past the last latitude), staged in LDS now and read back by the epilogue:
  // looked up there, each of its eight row groups waited for two dependent index loads behind a divergent branch -- 17.5 k of a
  // tile's 211 k clocks (wave-0 stamps, tools/leg_stamp.py)
  int *erow = (int *)((char *)As + LG_LDS_BYTES);
  const int lb = g.lbase[m], ndglu = g.lbase[m + 1] - lb;
  if (tid < 128) {
    const int jj = lat0 + (tid & 63);
    int r_ = -1;
    if (jj < ndglu) r_ = (tid < 64 ? g.legN : g.legS)[lb + jj];
    erow[tid] = r_;
  }
  LEG_STAMP_PRO(1);
  LEG_STAMP_BEGIN(1);
  for (int s = 0; s < nst; s++) {
    if (s > 0) EMI_SYNC();
    LEG_STAMP(0);
    *(lgvec *)sA0 = ra0;
    *(lgvec *)sA1 = ra1;
    *(lgvec *)sB0 = rb0;
    *(lgvec *)sB1 = rb1;
    *(lgvec *)sB2 = rb2;
    *(lgvec *)sB3 = rb3;
    LEG_STAMP(1);
    // the next stage is requested BEFORE the second barrier (the LDS writes have read their registers): issuing six loads took 320 clocks
    // of every stage between that barrier and the first matrix instruction
    if (s + 1 < nst) {
      uS += stepA;
      uA += stepA;
      uW += stepW;
      EMI_OPAQUE(voA);  // the zero-extension of the lane offset has to sit in this block for the scalar-base form to be selected
      EMI_OPAQUE(voB);
      ra0 = emi_ld_sv<lgvec>(uS, voA);
      ra1 = emi_ld_sv<lgvec>(uA, voA);
      rb0 = emi_ld_sv<lgvec>(uW, voB);
      rb1 = emi_ld_sv<lgvec>(uW + rowW4, voB);
      rb2 = emi_ld_sv<lgvec>(uW + 2 * rowW4, voB);
      rb3 = emi_ld_sv<lgvec>(uW + 3 * rowW4, voB);
    }
    LEG_STAMP(2);
    EMI_SYNC();
    LEG_STAMP(3);
    EMI_PRIO_HI();
#pragma unroll
    for (int p = 0; p < 2; p++)
#pragma unroll
      for (int ks = 0; ks < LG_KR / 4; ks++) {
        real_t a[2], b[4];
#pragma unroll
        for (int i = 0; i < 2; i++) a[i] = As[fa[p][ks] + i * 16];
#pragma unroll
        for (int j = 0; j < 4; j++) b[j] = Bs[fb[p][ks] + j * 16];
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
          for (int j = 0; j < 4; j++) acc[p][i][j] = LegAcc<WIDE>::mma(a[i], b[j], acc[p][i][j]);
      }
    EMI_PRIO_LO();
    LEG_STAMP(4);
  }
  LEG_STAMP_END(nst);
  LEG_STAMP_EPI0();
  // epilogue (ASRE1B): rows = latitudes
  int ern[2][4], ers[2][4];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int jl = wm * 32 + i * 16 + LegAcc<WIDE>::row(l, q);
      ern[i][q] = erow[jl];
      ers[i][q] = erow[64 + jl];
    }
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int q = 0; q < 4; q++) {
      if (ern[i][q] >= 0) {
        real_t *pn = FB + (long long)ern[i][q] * ldf + col0 + wn * 64 + (l & 15);
        real_t *ps = FB + (long long)ers[i][q] * ldf + col0 + wn * 64 + (l & 15);
#pragma unroll
        for (int jn = 0; jn < 4; jn++) {
          const auto sv = acc[0][i][jn][q], av = acc[1][i][jn][q];  // WIDE: the sum and difference in double too
          pn[jn * 16] = (real_t)(sv + av);
          ps[jn * 16] = (real_t)(sv - av);
        }
      }
    }
  LEG_STAMP_EPI(1);
}
EMI_KERNEL_LB2(256, EMI_LEG_INV_WAVES) void k_leg_inv(EmiGeomDev g, const int2 *tilemap, const real_t *W, int ldw, real_t *FB, int ldf) {
  // host-built tile map (leg_tilemap): block -> (local wavenumber, latitude tile, column tile),
  // 2-D blocked per XCD for L2 reuse; padding entries have x < 0
  const int2 tm = tilemap[EMI_BID];
  if (tm.x < 0) return;
  leg_inv_tile<false>(g, tm.x, tm.y >> 16, tm.y & 0xffff, W, ldw, FB, ldf);
}
#if EMI_LEG_WIDE_KERNEL
// fp32 library: the tiles of zonal wavenumber 0, which accumulate in double (LegAcc<true>), from a tile map of their own (leg_tilemaps)
EMI_KERNEL_LB2(256, 2) void k_leg_inv_wide(EmiGeomDev g, const int2 *tilemap, const real_t *W, int ldw, real_t *FB, int ldf) {
  const int2 tm = tilemap[EMI_BID];
  if (tm.x < 0) return;
  leg_inv_tile<true>(g, tm.x, tm.y >> 16, tm.y & 0xffff, W, ldw, FB, ldf);
}
#endif

// ---- direct: W[m][n][col] = sum_lat P[lat,n] * (FB_north +- FB_south)[lat][col]
// (prfi2b_mod.F90:82-94, ledir_mod.F90:100-267 DGEMM('T','N') x2; Gaussian weights and
//  1/(a cos) were folded into FB by k_fft_dir)
// fp64 tile (round 5): 128 k (n-pairs) of ONE parity x 128 columns; wave (wk, wn) owns the 16-row groups wk, wk + 2, wk + 4, wk + 6 of the
// tile x 64 col (interleaved, so that a partial tile occupies both k waves alike).  The tile of rounds 1 - 4 (leg_dir_tile2 below: 64 k of
// BOTH parities) has the same sixteen accumulator fragments and 64 | 128 matrix instructions per wave and stage, but stores 12 instead
// of 8 vectors per thread to LDS (panel rows of both parities, sums AND differences of the Fourier rows) and forms 16 instead of 8 sums
// per stage -- the two largest items of round 4's piece-removal table; loads per thread and stage are twelve in both (four panel
// vectors, four north and four south rows).  The last <= 64 n-pairs of a wavenumber still run on ONE two-parity tile (two half-empty
// one-parity tiles would each pay a full stage's loads and barriers), and the fp32 library uses the two-parity tile throughout.
// 90.1 - 91.5 against 95.5 ms at TCo1279 (profiles/r5_fft_experiments.txt section 5).
// FULL: all eight 16-row groups of the tile are live: the stage loop is then one
// straight-line block, which lets the compiler interleave the LDS fragment reads with the MFMAs.
// EMI_LD_COMB_BACK: the k step of a stage (counted from its end) in front of whose matrix instructions north +- south of the NEXT stage's
// rows is formed (measured 1 | 2 | 3 | wherever the scheduler puts them: 95.7 | 92.6 | 94.3 | 92.5 - 93.1 ms, without the two-parity tails).
EMI_DEVFN real2 lg_fma(real2 b, real_t sg, real2 a) { return mk2(a.x + sg * b.x, a.y + sg * b.y); }  // a + sg b
EMI_DEVFN v4f lg_fma(v4f b, real_t sg, v4f a) { return a + b * (float)sg; }
#ifndef EMI_LD_COMB_BACK
#define EMI_LD_COMB_BACK 2
#endif
#define LG_LDK LG_LDB  // row stride of both LDS images of leg_dir_tile (128 values + padding, as Bs of the other kernels)
template <bool FULL, bool WIDE>
EMI_DEVFN void leg_dir_tile(const EmiGeomDev &g, const int m, const int par, const int kt, const int ct, const int nlive, const real_t *FB, const int zrow, int ldf,
                            real_t *W, int ldw, const FuseDst *fd) {
  typedef typename LegAcc<WIDE>::type acc_t;
  LEG_STAMP_T0();
  EMI_LDS_DECL;
  real_t *As = (real_t *)EMI_LDS_PTR;
  real_t *Bs = As + LG_LS * LG_LDK;
  const int tid = EMI_TID, w = tid >> 6, l = tid & 63;
  const int wk = w & 1, wn = w >> 1;
  const int k0 = kt * 128, col0 = ct * LG_BN;
  const int nkpad = g.wrows[m] >> 1;
  const int lb = g.lbase[m], ndglu = g.lbase[m + 1] - lb;
  // at least one stage: a wavenumber above every latitude's NMEN (truncation finer than the grid) has no latitudes but still has coefficients
  // to zero -- its one stage multiplies the zero row; with no stage at all the prologue's loads would use row numbers nobody staged
  const int nst = ndglu > 0 ? (ndglu + LG_LS - 1) / LG_LS : 1;
  const long long wb = g.wbase[m];
#ifdef EMI_CPU_EMU  // lanes are threads there and a matrix instruction is a rendezvous of the whole workgroup: every wave takes the same number
  const int ni = FULL ? 4 : (nlive > 4 ? 4 : nlive);
#else
  const int ni = FULL ? 4 : (nlive - wk + 1) >> 1;  // live 16-row groups of this wave: groups wk, wk + 2, wk + 4, wk + 6 of the tile (interleaved, so that a partial tile occupies both k waves alike)
#endif
  const real_t sg = par ? (real_t)-1.0 : (real_t)1.0;

  acc_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = (acc_t){0.0, 0.0, 0.0, 0.0};

  constexpr int LA = 128 / LGDV, RA = LG_THREADS / LA;  // lanes per row of 128 values; rows per pass of the workgroup: 4 | 8, four passes per stage
  constexpr int WRA = 64 / LA;                         // rows that one wave loads per instruction: 1 | 2
  static_assert(LA == 64, "a wave loads ONE row of 128 values per instruction: row addresses are scalar");
  const int wv = emi_uniform(w);
  const int arow = tid / LA, ac = (tid % LA) * LGDV;
  const int ldk = g.ldk[m];
  unsigned voA = (unsigned)(((l / LA) * ldk + ac) * (int)sizeof(real_t));
  const char *uP = emi_uniform_ptr((const real_t *)g.PT + (par ? g.offTA[m] : g.offTS[m]) + (long long)(wv * WRA) * ldk + k0);
  const long long stepA = (long long)LG_LS * ldk * (long long)sizeof(real_t), rowA = (long long)RA * ldk * (long long)sizeof(real_t);
  unsigned voB = (unsigned)(ac * (int)sizeof(real_t));
  const char *uFB = emi_uniform_ptr(FB + col0);
  lgdvec ra0, ra1, ra2, ra3;                      // P^T rows of stage s+1
  lgdvec rn0, rn1, rn2, rn3, rs0, rs1, rs2, rs3;  // FB rows (north, south) of stage s+1; rn* then hold the combination
  FuseDst *efd = (FuseDst *)(Bs + LG_LS * LG_LDK);
  if (tid < 64) {
    FuseDst d_;
    d_.dst = nullptr;
    d_.stride = 0;
    d_.idx = 0;
    if (fd) d_ = fd[(col0 >> 1) + tid];
    efd[tid] = d_;
  }
  const int nasm0_m = emi_ld_const(g.nasm0, m), mval_m = emi_ld_const(g.mval, m);
  int qn[4], qs[4];
  const int lb_last = lb + ndglu > 0 ? lb + ndglu - 1 : 0;  // a row number that exists, for the look-ups past the last latitude (their rows are then replaced by the zero row)
#define LEGDIR_ROWS(s_)                                       \
  {                                                        \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; i_++) {     \
      const int j_ = LG_LS * (s_) + wv + RA * i_;          \
      const int jc_ = j_ < ndglu ? lb + j_ : lb_last;      \
      qn[i_] = emi_ld_const(g.legN, jc_);                  \
      qs[i_] = emi_ld_const(g.legS, jc_);                  \
    }                                                      \
  }
  unsigned qrn[4], qrs[4];
#define LEGDIR_SEL(s_)                                        \
  {                                                        \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; i_++) {     \
      const bool live_ = LG_LS * (s_) + wv + RA * i_ < ndglu; \
      qrn[i_] = live_ ? qn[i_] : zrow;                     \
      qrs[i_] = live_ ? qs[i_] : zrow;                     \
    }                                                      \
  }
#define LEGDIR_LOADB(s_)                                                               \
  {                                                                                 \
    const unsigned long long ldfb_ = (unsigned long long)(unsigned)ldf * sizeof(real_t); \
    EMI_OPAQUE(voB);                                                                \
    rn0 = emi_ld_sv<lgdvec>(uFB + qrn[0] * ldfb_, voB);                                 \
    rs0 = emi_ld_sv<lgdvec>(uFB + qrs[0] * ldfb_, voB);                                 \
    rn1 = emi_ld_sv<lgdvec>(uFB + qrn[1] * ldfb_, voB);                                 \
    rs1 = emi_ld_sv<lgdvec>(uFB + qrs[1] * ldfb_, voB);                                 \
    rn2 = emi_ld_sv<lgdvec>(uFB + qrn[2] * ldfb_, voB);                                 \
    rs2 = emi_ld_sv<lgdvec>(uFB + qrs[2] * ldfb_, voB);                                 \
    rn3 = emi_ld_sv<lgdvec>(uFB + qrn[3] * ldfb_, voB);                                 \
    rs3 = emi_ld_sv<lgdvec>(uFB + qrs[3] * ldfb_, voB);                                 \
  }
#define LEGDIR_LOADA(s_)                                             \
  {                                                               \
    const char *up_ = uP + (s_) * stepA;                          \
    EMI_OPAQUE(voA);                                              \
    ra0 = emi_ld_sv<lgdvec>(up_, voA);                             \
    ra1 = emi_ld_sv<lgdvec>(up_ + rowA, voA);                      \
    ra2 = emi_ld_sv<lgdvec>(up_ + 2 * rowA, voA);                  \
    ra3 = emi_ld_sv<lgdvec>(up_ + 3 * rowA, voA);                  \
  }
  int fa[LG_LS / 4], fb[LG_LS / 4];
#pragma unroll
  for (int ks = 0; ks < LG_LS / 4; ks++) {
    fa[ks] = (4 * ks + (l >> 4)) * LG_LDK + wk * 16 + (l & 15);
    fb[ks] = (4 * ks + (l >> 4)) * LG_LDK + wn * 64 + (l & 15);
    EMI_OPAQUE(fa[ks]);
    EMI_OPAQUE(fb[ks]);
  }
  // PRFI2B for this parity: north + south | north - south, in place in the north registers
#define LEGDIR_COMBINE()              \
  {                                \
    rn0 = lg_fma(rs0, sg, rn0);    \
    rn1 = lg_fma(rs1, sg, rn1);    \
    rn2 = lg_fma(rs2, sg, rn2);    \
    rn3 = lg_fma(rs3, sg, rn3);    \
  }
  LEGDIR_ROWS(0);
  LEGDIR_SEL(0);
  LEGDIR_LOADB(0);
  LEGDIR_LOADA(0);
  LEGDIR_COMBINE();
  LEGDIR_ROWS(nst > 1 ? 1 : 0);
  LEG_STAMP_PRO(2);
  LEG_STAMP_BEGIN(2);
  // Order of a stage: barrier, LDS writes (no arithmetic), loads of the next stage, barrier, matrix phase, north +- south of the rows
  // that arrived meanwhile.  The fp64 sums used to sit in front of the LDS writes, where the wave runs at low priority beside the
  // other workgroup's matrix phase and every one waited for a gap between matrix instructions (2.3 k of a stage's 9 k clocks); at the end
  // of the wave's own matrix phase they issue back to back.
  // LAST_: the last stage of the tile requests nothing (the loop is peeled rather than guarded: with the loads inside an
  // `if (s + 1 < nst)` the compiler copies all twelve prefetch registers at the loop back edge, 44 moves per stage; and a last stage that
  // re-requested its own rows, as it did until round 4, made the epilogue wait a memory round trip for data nobody reads)
#define LEGDIR_STAGE(s, LAST_)                                                                                            \
  {                                                                                                                       \
    if ((s) > 0) EMI_SYNC();                                                                                              \
    LEG_STAMP(0);                                                                                                         \
    if constexpr (!(LAST_)) {                                                                                             \
      /* row numbers: those of stage s+1 (requested a stage ago) are consumed, those of stage s+2 requested -- here, ahead of the LDS */ \
      /* writes and the second barrier, so that the matrix phase's first fragment reads never wait on a scalar load */     \
      LEGDIR_SEL((s) + 1);                                                                                                \
      const int sn2 = ((s) + 2 < nst) ? (s) + 2 : (s) + 1;                                                                \
      LEGDIR_ROWS(sn2);                                                                                                   \
      EMI_SCHED_FENCE(); /* left to itself the scheduler sinks the scalar loads to the wait in front of the second barrier */ \
    }                                                                                                                     \
    /* As[latitude in stage][k index], Bs[latitude in stage][column] */                                                   \
    *(lgdvec *)(As + (arow) * LG_LDK + ac) = ra0;                                                                          \
    *(lgdvec *)(As + (arow + RA) * LG_LDK + ac) = ra1;                                                                     \
    *(lgdvec *)(As + (arow + 2 * RA) * LG_LDK + ac) = ra2;                                                                 \
    *(lgdvec *)(As + (arow + 3 * RA) * LG_LDK + ac) = ra3;                                                                 \
    *(lgdvec *)(Bs + (arow) * LG_LDK + ac) = rn0;                                                                          \
    *(lgdvec *)(Bs + (arow + RA) * LG_LDK + ac) = rn1;                                                                     \
    *(lgdvec *)(Bs + (arow + 2 * RA) * LG_LDK + ac) = rn2;                                                                 \
    *(lgdvec *)(Bs + (arow + 3 * RA) * LG_LDK + ac) = rn3;                                                                 \
    LEG_STAMP(1);                                                                                                         \
    if constexpr (!(LAST_)) {                                                                                             \
      LEGDIR_LOADB((s) + 1);                                                                                              \
      LEGDIR_LOADA((s) + 1);                                                                                              \
    }                                                                                                                     \
    LEG_STAMP(2);                                                                                                         \
    EMI_SYNC();                                                                                                           \
    LEG_STAMP(3);                                                                                                         \
    EMI_PRIO_HI();                                                                                                        \
    _Pragma("unroll") for (int ks = 0; ks < LG_LS / 4; ks++) {                                                            \
      real_t a[4], b[4];                                                                                                  \
      _Pragma("unroll") for (int i = 0; i < 4; i++) a[i] = As[fa[ks] + i * 32];                                           \
      _Pragma("unroll") for (int j = 0; j < 4; j++) b[j] = Bs[fb[ks] + j * 16];                                           \
      if constexpr (!(LAST_)) {                                                                                           \
        /* north +- south of the rows requested in this stage, ahead of the stage's last sixteen matrix instructions: in front of the matrix */ \
        /* phase (where the scheduler puts them by itself) the wave waits for the rows just requested, behind the last matrix instruction */ \
        /* the LDS writes of the next stage wait for the sums (profiles/r5_fft_experiments.txt section 5) */              \
        if (ks == LG_LS / 4 - EMI_LD_COMB_BACK) {                                                                         \
          EMI_SCHED_FENCE();                                                                                              \
          LEGDIR_COMBINE();                                                                                               \
          EMI_SCHED_FENCE();                                                                                              \
        }                                                                                                                 \
      }                                                                                                                   \
      _Pragma("unroll") for (int i = 0; i < 4; i++)                                                                       \
        if (FULL || i < ni) { /* the last k tile of a wavenumber: 16-row groups past the end are skipped */               \
          _Pragma("unroll") for (int j = 0; j < 4; j++) acc[i][j] = LegAcc<WIDE>::mma(a[i], b[j], acc[i][j]);             \
        }                                                                                                                 \
    }                                                                                                                     \
    EMI_PRIO_LO();                                                                                                        \
    LEG_STAMP(4);                                                                                                         \
  }
  for (int s = 0; s < nst - 1; s++) LEGDIR_STAGE(s, false);
  LEGDIR_STAGE(nst - 1, true);
#undef LEGDIR_STAGE
  LEG_STAMP_END(nst);
  LEG_STAMP_EPI0();
#undef LEGDIR_ROWS
#undef LEGDIR_SEL
#undef LEGDIR_COMBINE
#undef LEGDIR_LOADA
#undef LEGDIR_LOADB
  // Epilogue.  Fields whose spectral output is a plain copy (UPDSP, updsp_mod.F90:100-161: every scalar) go
  // straight to the caller's array -- element (NASM0(m) + 2 (n-m) + c, field), n <= N, imaginary parts of m = 0
  // zero (updspb_mod.F90:106,117) -- instead of through W and k_postpack_dir; the wind fields (U, V), which
  // UVTVD combines over n-1, n, n+1, and the padding columns still go to W.  A lane holds one component
  // (c = l & 1) of four fields.
  real_t *ud[4];
  long long us[4];
  const int cpar = l & 1;
#pragma unroll
  for (int jn = 0; jn < 4; jn++) {
    ud[jn] = nullptr;
    us[jn] = 0;
    const FuseDst d = efd[wn * 32 + jn * 8 + ((l & 15) >> 1)];
    if (d.dst) {
      ud[jn] = (real_t *)d.dst + d.idx + (long long)(nasm0_m + cpar) * d.stride;
      us[jn] = 2LL * d.stride;
    }
  }
  const int rmax = g.nsmax - mval_m;
  const bool zero_im = (mval_m == 0) && cpar;
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int q = 0; q < 4; q++) {
      int k = k0 + (wk + 2 * i) * 16 + LegAcc<WIDE>::row(l, q);
      if (k < nkpad) {
        const int r = 2 * k + par;
        real_t *pw = W + (wb + r) * ldw + col0 + wn * 64 + (l & 15);
#pragma unroll
        for (int jn = 0; jn < 4; jn++) {
          if (ud[jn]) {
            if (r <= rmax) ud[jn][(long long)r * us[jn]] = zero_im ? (real_t)0.0 : (real_t)acc[i][jn][q];
          } else {
            pw[jn * 16] = (real_t)acc[i][jn][q];
          }
        }
      }
    }
  LEG_STAMP_EPI(2);
}
// ---- direct, the tile of rounds 1 - 4: 64 k (n-pairs) x 2 parities x 128 columns; wave (par, wn) owns 64 k x 64 col.  Kept for the
// LAST tile of a wavenumber when at most 64 n-pairs are left: one workgroup then does what two half-empty one-parity tiles would, each
// with a full stage's loads and barriers (k_leg_dir below).
template <bool FULL, bool WIDE>
EMI_DEVFN void leg_dir_tile2(const EmiGeomDev &g, const int m, const int kt, const int ct, const int ni, const real_t *FB, const int zrow, int ldf,
                            real_t *W, int ldw, const FuseDst *fd) {
  typedef typename LegAcc<WIDE>::type acc_t;
  LEG_STAMP_T0();
  EMI_LDS_DECL;
  real_t *As = (real_t *)EMI_LDS_PTR;
  real_t *Bs = As + 2 * LG_LS * LG_LDA;
  const int tid = EMI_TID, w = tid >> 6, l = tid & 63;
  const int par = w & 1, wn = w >> 1;
  const int k0 = kt * 64, col0 = ct * LG_BN;
  const int nkpad = g.wrows[m] >> 1;
  const int lb = g.lbase[m], ndglu = g.lbase[m + 1] - lb;
  // at least one stage: a wavenumber above every latitude's NMEN (truncation finer than the grid) has no latitudes but still has coefficients
  // to zero -- its one stage multiplies the zero row; with no stage at all the prologue's loads would use row numbers nobody staged
  const int nst = ndglu > 0 ? (ndglu + LG_LS - 1) / LG_LS : 1;  // stages of 16 | 32 latitudes: 64 | 128 MFMAs per wave between barriers
  const long long wb = g.wbase[m];

  acc_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = (acc_t){0.0, 0.0, 0.0, 0.0};

  // P^T tile: LG_LS latitudes x 64 k per parity, k contiguous in HBM (coalesced rows) and in LDS; a row is 64 / LGDV lanes wide, so the
  // workgroup covers RA = 8 | 16 latitude rows per pass and the stage in two passes
  constexpr int LA = 64 / LGDV, RA = LG_THREADS / LA, LB = LG_BN / LGDV, RB = LG_THREADS / LB;  // RB = 4 | 8 Fourier rows per pass, four passes
  constexpr int WRA = 64 / LA;                       // panel rows that one wave loads per instruction
  static_assert(LB == 64, "a wave loads ONE Fourier row per instruction: row addresses are scalar");
  const int wv = emi_uniform(w);
  const int arow = tid / LA, ac = (tid % LA) * LGDV;  // latitude rows arow and arow + RA of the stage
  const int ldk = g.ldk[m];
  // as in leg_inv_tile: uniform bases in scalar registers + constant 32-bit lane offsets, no vector address arithmetic per stage
  unsigned voA = (unsigned)(((l / LA) * ldk + ac) * (int)sizeof(real_t));
  const char *uS = emi_uniform_ptr((const real_t *)g.PT + g.offTS[m] + (long long)(wv * WRA) * ldk + k0);
  const char *uA = emi_uniform_ptr((const real_t *)g.PT + g.offTA[m] + (long long)(wv * WRA) * ldk + k0);
  const long long stepA = (long long)LG_LS * ldk * (long long)sizeof(real_t), rowA8 = (long long)RA * ldk * (long long)sizeof(real_t);
  const int brow = tid / LB, bc = (tid % LB) * LGDV;  // latitude rows brow + RB i, i = 0..3, of each stage
  unsigned voB = (unsigned)(bc * (int)sizeof(real_t));
  const char *uFB = emi_uniform_ptr(FB + col0);
  lgdvec ra0, ra1, ra2, ra3;                      // P^T of stage s+1
  lgdvec rn0, rn1, rn2, rn3, rs0, rs1, rs2, rs3;  // FB rows (north, south) of stage s+1
  // The FB rows of one zonal wavenumber are ~26 MB apart (FB is latitude-major for the FFT kernels), so each stage touches 32
  // far-apart rows, prefetched one stage (~2 x 4096 MFMA cycles per SIMD) ahead.  A wave loads one whole row piece per
  // instruction (two values per lane in both precisions), so its row numbers (fbase[lat]+m) are SCALAR loads from the latitude tables, fetched a further stage ahead, and the
  // row address is scalar arithmetic (before: eight LDS look-ups, eight 32 x 32 -> 64-bit vector multiply-adds and eight 64-bit vector
  // adds per stage, which together took the matrix pipe for as long as two matrix instructions).  Latitudes past the last one read
  // row `zrow` of the buffer, a row of zeros behind the Fourier rows: no branch around the loads.
  // (Until round 5 the fp32 kernel loaded four values per lane, two rows per instruction, and staged the row numbers of the whole
  // wavenumber in LDS -- 10 - 20 KB per workgroup on top of a 57 KB stage image, which held it at two workgroups per CU.)
  // destinations of the tile's 64 fields (fused epilogue), staged in LDS now: fetched in the epilogue they cost eight serialised memory
  // round trips (descriptor, then index / stride / NASM0 behind a divergent branch, per 16-column group) -- about 20 k of the 35 k clocks
  // a tile spent between its last matrix instruction and its last store (tools/leg_stamp.py)
  FuseDst *efd = (FuseDst *)(Bs + 2 * LG_LS * LG_LDB);
  if (tid < 64) {
    FuseDst d_;
    d_.dst = nullptr;
    d_.stride = 0;
    d_.idx = 0;
    if (fd) d_ = fd[(col0 >> 1) + tid];
    efd[tid] = d_;
  }
  const int nasm0_m = emi_ld_const(g.nasm0, m), mval_m = emi_ld_const(g.mval, m);
  int qn[4], qs[4];  // row numbers of the stage that is requested next
  const int lb_last = lb + ndglu > 0 ? lb + ndglu - 1 : 0;  // a row number that exists, for the look-ups past the last latitude (their rows are then replaced by the zero row)
#define LEGDIR_ROWS(s_)                                    \
  {                                                        \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; i_++) {     \
      const int j_ = LG_LS * (s_) + wv + RB * i_;          \
      const int jc_ = j_ < ndglu ? lb + j_ : lb_last;      \
      qn[i_] = emi_ld_const(g.legN, jc_);                  \
      qs[i_] = emi_ld_const(g.legS, jc_);                  \
    }                                                      \
  }
  unsigned qrn[4], qrs[4];  // the rows that LEGDIR_LOADB requests
  // the selects sit a stage after the scalar loads were issued (nothing waits for them) and BEFORE the next ones overwrite qn / qs
#define LEGDIR_SEL(s_)                                     \
  {                                                        \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; i_++) {     \
      const bool live_ = LG_LS * (s_) + wv + RB * i_ < ndglu; \
      qrn[i_] = live_ ? qn[i_] : zrow;                     \
      qrs[i_] = live_ ? qs[i_] : zrow;                     \
    }                                                      \
  }
#define LEGDIR_LOADB(s_)                                                            \
  {                                                                                 \
    const unsigned long long ldfb_ = (unsigned long long)(unsigned)ldf * sizeof(real_t); \
    EMI_OPAQUE(voB);                                                                \
    rn0 = emi_ld_sv<lgdvec>(uFB + qrn[0] * ldfb_, voB);                                 \
    rs0 = emi_ld_sv<lgdvec>(uFB + qrs[0] * ldfb_, voB);                                 \
    rn1 = emi_ld_sv<lgdvec>(uFB + qrn[1] * ldfb_, voB);                                 \
    rs1 = emi_ld_sv<lgdvec>(uFB + qrs[1] * ldfb_, voB);                                 \
    rn2 = emi_ld_sv<lgdvec>(uFB + qrn[2] * ldfb_, voB);                                 \
    rs2 = emi_ld_sv<lgdvec>(uFB + qrs[2] * ldfb_, voB);                                 \
    rn3 = emi_ld_sv<lgdvec>(uFB + qrn[3] * ldfb_, voB);                                 \
    rs3 = emi_ld_sv<lgdvec>(uFB + qrs[3] * ldfb_, voB);                                 \
  }
#define LEGDIR_LOADA(s_)                                          \
  {                                                               \
    const char *us_ = uS + (s_) * stepA, *ua_ = uA + (s_) * stepA; \
    EMI_OPAQUE(voA);                                              \
    ra0 = emi_ld_sv<lgdvec>(us_, voA);                             \
    ra1 = emi_ld_sv<lgdvec>(ua_, voA);                             \
    ra2 = emi_ld_sv<lgdvec>(us_ + rowA8, voA);                     \
    ra3 = emi_ld_sv<lgdvec>(ua_ + rowA8, voA);                     \
  }
  // fragment positions, one register per k step (leg_inv_tile)
  int fa[LG_LS / 4], fb[LG_LS / 4];
#pragma unroll
  for (int ks = 0; ks < LG_LS / 4; ks++) {
    fa[ks] = (par * LG_LS + 4 * ks + (l >> 4)) * LG_LDA + (l & 15);
    fb[ks] = (par * LG_LS + 4 * ks + (l >> 4)) * LG_LDB + wn * 64 + (l & 15);
    EMI_OPAQUE(fa[ks]);
    EMI_OPAQUE(fb[ks]);
  }
  // PRFI2B: (north, south) -> (symmetric, antisymmetric), in place, as soon as the rows of the next stage have arrived
#define LEGDIR_SUMDIFF()                                   \
  {                                                        \
    lgdvec t_;                                              \
    t_ = lg_sub(rn0, rs0), rn0 = lg_add(rn0, rs0), rs0 = t_; \
    t_ = lg_sub(rn1, rs1), rn1 = lg_add(rn1, rs1), rs1 = t_; \
    t_ = lg_sub(rn2, rs2), rn2 = lg_add(rn2, rs2), rs2 = t_; \
    t_ = lg_sub(rn3, rs3), rn3 = lg_add(rn3, rs3), rs3 = t_; \
  }
  LEGDIR_ROWS(0);
  LEGDIR_SEL(0);
  LEGDIR_LOADB(0);
  LEGDIR_LOADA(0);
  LEGDIR_SUMDIFF();
  // Order of a stage: barrier, LDS writes (no arithmetic), loads of the next stage, barrier, matrix phase, sums and differences of the rows
  // that arrived meanwhile.  The sixteen fp64 adds used to sit in front of the LDS writes, where the wave runs at low priority beside the
  // other workgroup's matrix phase and every add waited for a gap between matrix instructions (2.3 k of a stage's 9 k clocks); at the end
  // of the wave's own matrix phase they issue back to back.
  LEGDIR_ROWS(nst > 1 ? 1 : 0);
  LEG_STAMP_PRO(2);
  LEG_STAMP_BEGIN(2);
  // One stage; LAST_: the last stage of the tile requests nothing (the loop is peeled rather than guarded: with the loads inside an
  // `if (s + 1 < nst)` the compiler copies all twelve prefetch registers at the loop back edge, 44 moves per stage; and a last stage that
  // re-requested its own rows, as it did until round 4, made the epilogue wait a memory round trip for data nobody reads)
#define LEGDIR_STAGE(s, LAST_)                                                                                            \
  {                                                                                                                       \
    if ((s) > 0) EMI_SYNC();                                                                                              \
    LEG_STAMP(0);                                                                                                         \
    if constexpr (!(LAST_)) {                                                                                             \
      /* row numbers: those of stage s+1 (requested a stage ago) are consumed, those of stage s+2 requested -- here, ahead of the LDS */ \
      /* writes and the second barrier, so that the matrix phase's first fragment reads never wait on a scalar load */     \
      LEGDIR_SEL((s) + 1);                                                                                                \
      const int sn2 = ((s) + 2 < nst) ? (s) + 2 : (s) + 1;                                                                \
      LEGDIR_ROWS(sn2);                                                                                                   \
      EMI_SCHED_FENCE(); /* left to itself the scheduler sinks the scalar loads to the wait in front of the second barrier */ \
    }                                                                                                                     \
    /* As[par][latitude in stage][k index], Bs[par][latitude in stage][column] */                                         \
    *(lgdvec *)(As + (0 * LG_LS + arow) * LG_LDA + ac) = ra0;                                                              \
    *(lgdvec *)(As + (1 * LG_LS + arow) * LG_LDA + ac) = ra1;                                                              \
    *(lgdvec *)(As + (0 * LG_LS + arow + RA) * LG_LDA + ac) = ra2;                                                         \
    *(lgdvec *)(As + (1 * LG_LS + arow + RA) * LG_LDA + ac) = ra3;                                                         \
    *(lgdvec *)(Bs + (0 * LG_LS + brow) * LG_LDB + bc) = rn0; /* symmetric part */                                         \
    *(lgdvec *)(Bs + (1 * LG_LS + brow) * LG_LDB + bc) = rs0; /* antisymmetric part */                                     \
    *(lgdvec *)(Bs + (0 * LG_LS + brow + RB) * LG_LDB + bc) = rn1;                                                         \
    *(lgdvec *)(Bs + (1 * LG_LS + brow + RB) * LG_LDB + bc) = rs1;                                                         \
    *(lgdvec *)(Bs + (0 * LG_LS + brow + 2 * RB) * LG_LDB + bc) = rn2;                                                     \
    *(lgdvec *)(Bs + (1 * LG_LS + brow + 2 * RB) * LG_LDB + bc) = rs2;                                                     \
    *(lgdvec *)(Bs + (0 * LG_LS + brow + 3 * RB) * LG_LDB + bc) = rn3;                                                     \
    *(lgdvec *)(Bs + (1 * LG_LS + brow + 3 * RB) * LG_LDB + bc) = rs3;                                                     \
    LEG_STAMP(1);                                                                                                         \
    if constexpr (!(LAST_)) {                                                                                             \
      LEGDIR_LOADB((s) + 1);                                                                                              \
      LEGDIR_LOADA((s) + 1);                                                                                              \
    }                                                                                                                     \
    LEG_STAMP(2);                                                                                                         \
    EMI_SYNC();                                                                                                           \
    LEG_STAMP(3);                                                                                                         \
    EMI_PRIO_HI();                                                                                                        \
    _Pragma("unroll") for (int ks = 0; ks < LG_LS / 4; ks++) {                                                            \
      real_t a[4], b[4];                                                                                                  \
      _Pragma("unroll") for (int i = 0; i < 4; i++) a[i] = As[fa[ks] + i * 16];                                           \
      _Pragma("unroll") for (int j = 0; j < 4; j++) b[j] = Bs[fb[ks] + j * 16];                                           \
      _Pragma("unroll") for (int i = 0; i < 4; i++)                                                                       \
        if (FULL || i < ni) { /* the last k tile of a wavenumber: 16-row groups past the end are skipped */               \
          _Pragma("unroll") for (int j = 0; j < 4; j++) acc[i][j] = LegAcc<WIDE>::mma(a[i], b[j], acc[i][j]);             \
        }                                                                                                                 \
    }                                                                                                                     \
    if constexpr (!(LAST_)) LEGDIR_SUMDIFF();                                                                             \
    EMI_PRIO_LO();                                                                                                        \
    LEG_STAMP(4);                                                                                                         \
  }
  for (int s = 0; s < nst - 1; s++) LEGDIR_STAGE(s, false);
  LEGDIR_STAGE(nst - 1, true);
#undef LEGDIR_STAGE
  LEG_STAMP_END(nst);
  LEG_STAMP_EPI0();
#undef LEGDIR_ROWS
#undef LEGDIR_SEL
#undef LEGDIR_SUMDIFF
#undef LEGDIR_LOADA
#undef LEGDIR_LOADB
  // Epilogue.  Fields whose spectral output is a plain copy (UPDSP, updsp_mod.F90:100-161: every scalar) go
  // straight to the caller's array -- element (NASM0(m) + 2 (n-m) + c, field), n <= N, imaginary parts of m = 0
  // zero (updspb_mod.F90:106,117) -- instead of through W and k_postpack_dir; the wind fields (U, V), which
  // UVTVD combines over n-1, n, n+1, and the padding columns still go to W.  A lane holds one component
  // (c = l & 1) of four fields.
  real_t *ud[4];
  long long us[4];
  const int cpar = l & 1;
#pragma unroll
  for (int jn = 0; jn < 4; jn++) {
    ud[jn] = nullptr;
    us[jn] = 0;
    const FuseDst d = efd[wn * 32 + jn * 8 + ((l & 15) >> 1)];
    if (d.dst) {
      ud[jn] = (real_t *)d.dst + d.idx + (long long)(nasm0_m + cpar) * d.stride;
      us[jn] = 2LL * d.stride;
    }
  }
  const int rmax = g.nsmax - mval_m;  // rows r = n - m <= rmax carry a coefficient
  const bool zero_im = (mval_m == 0) && cpar;
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int q = 0; q < 4; q++) {
      int k = k0 + i * 16 + LegAcc<WIDE>::row(l, q);
      if (k < nkpad) {
        const int r = 2 * k + par;
        real_t *pw = W + (wb + r) * ldw + col0 + wn * 64 + (l & 15);
#pragma unroll
        for (int jn = 0; jn < 4; jn++) {
          if (ud[jn]) {
            if (r <= rmax) ud[jn][(long long)r * us[jn]] = zero_im ? (real_t)0.0 : (real_t)acc[i][jn][q];
          } else {
            pw[jn * 16] = (real_t)acc[i][jn][q];
          }
        }
      }
    }
  LEG_STAMP_EPI(2);
}
EMI_KERNEL_LB2(256, EMI_LEG_DIR_WAVES) void k_leg_dir(EmiGeomDev g, const int2 *tilemap, const real_t *FB, const int zrow, int ldf, real_t *W, int ldw,
                                      const FuseDst *fd) {
  const int2 tm = tilemap[EMI_BID];
  if (tm.x < 0) return;
  // row tiles of a wavenumber with nk n-pairs (Plan::ktile_pref counts them the same way).  fp64: 2 floor(nk / 128) one-parity tiles of
  // 128 n-pairs (parity + 2 x tile), then the rest r = nk mod 128: none | ONE two-parity tile of 64 n-pairs (r <= 64) | two one-parity
  // tiles.  fp32: two-parity tiles of 64 n-pairs throughout (the one-parity tile is 2 % slower there, profiles/r5_fft_experiments.txt section 5).
  const int m = tm.x, rt = tm.y >> 16, ct = tm.y & 0xffff;
  const int nk = g.wrows[m] >> 1;
  if constexpr (sizeof(real_t) == 4) {
    const int left = (nk - rt * 64 + 15) >> 4;  // live 16-row groups of this tile
    if (left >= 4)
      leg_dir_tile2<true, false>(g, m, rt, ct, 4, FB, zrow, ldf, W, ldw, fd);
    else
      leg_dir_tile2<false, false>(g, m, rt, ct, left, FB, zrow, ldf, W, ldw, fd);
  } else {
    const int nfull = nk >> 7, r = nk - (nfull << 7);
    if (rt >= 2 * nfull && r <= 64) {
      leg_dir_tile2<false, false>(g, m, 2 * nfull, ct, (r + 15) >> 4, FB, zrow, ldf, W, ldw, fd);
      return;
    }
    const int par = rt & 1, kt = rt >> 1;        // (in the rest rt - 2 nfull is the parity: 2 nfull is even)
    const int left = (nk - kt * 128 + 15) >> 4;  // live 16-row groups of this tile
    if (left >= 8)
      leg_dir_tile<true, false>(g, m, par, kt, ct, 8, FB, zrow, ldf, W, ldw, fd);
    else
      leg_dir_tile<false, false>(g, m, par, kt, ct, left, FB, zrow, ldf, W, ldw, fd);
  }
}
#if EMI_LEG_WIDE_KERNEL
// fp32 library: the tiles of zonal wavenumber 0 accumulate in double (LegAcc<true>, ledir_mod.F90:133-171), from a tile map of their own
EMI_KERNEL_LB2(256, 2) void k_leg_dir_wide(EmiGeomDev g, const int2 *tilemap, const real_t *FB, const int zrow, int ldf, real_t *W, int ldw,
                                           const FuseDst *fd) {
  const int2 tm = tilemap[EMI_BID];
  if (tm.x < 0) return;
  const int m = tm.x, rt = tm.y >> 16, ct = tm.y & 0xffff;
  const int left = ((g.wrows[m] >> 1) - rt * 64 + 15) >> 4;
  leg_dir_tile2<false, true>(g, m, rt, ct, left < 4 ? left : 4, FB, zrow, ldf, W, ldw, fd);
}
#endif

// ==========================================================================================
// FFT engine in LDS (v2): in-place mixed-radix Cooley-Tukey on `nfl` fields of S complex points.
//   DIT: input at perm[] positions -> natural output.  DIF: natural input -> output at perm[].
//   tw[k] = exp(-2 pi i k/S); sgn=+1 conjugates.
//   * radices 2,3,4,5,8 are hard-coded register butterflies; 7 uses its DFT matrix from the twiddle table
//   * the host orders the factors so that every pass stride (lenp) of a 2-3-5-smooth size is a power
//     of two (odd radices last in DIT order) -> no integer divisions in those passes
//   * logical element i lives at LDS slot FPAD(i) = i ^ ((i >> 3) & 15) (XOR swizzle inside aligned
//     16-element blocks): stride-1 passes stay conflict free and the pass whose butterflies are
//     contiguous runs of 8 elements (stride 128 B between lanes) becomes conflict free too
//   * inter-pass twiddles come from per-pass tables laid out [t][j] so that a wave reads them
//     coalesced (the single table tw[j*t*S/len] is a 64-line gather per wave instruction)
//   * Bluestein: DIF passes -> [last DIF pass + pointwise filter + first DIT pass fused in
//     registers] -> DIT passes; the final DIT pass of the inverse transform multiplies by the chirp
//     and stores the real row straight to the user's grid array (no LDS round trip)
// ==========================================================================================

EMI_DEVFN real2 tw_get(const real2 *tw, int idx, int sgn) {
  real2 t = tw[idx];
  if (sgn > 0) t.y = -t.y;
  return t;
}
// multiply by -i (forward, sgn<0) or +i (inverse)
EMI_DEVFN real2 cmul_mi(real2 a, int sgn) { return (sgn < 0) ? mk2(a.y, -a.x) : mk2(-a.y, a.x); }

EMI_DEVFN void bf4(real2 &x0, real2 &x1, real2 &x2, real2 &x3, int sgn) {
  real2 a = cadd(x0, x2), b = csub(x0, x2), c = cadd(x1, x3), d = cmul_mi(csub(x1, x3), sgn);
  x0 = cadd(a, c);
  x1 = cadd(b, d);
  x2 = csub(a, c);
  x3 = csub(b, d);
}

template <int R>
EMI_DEVFN void butterfly(real2 *v, const real2 *tw, int S, int sgn) {
  if (R == 2) {
    real2 a = v[0], b = v[1];
    v[0] = cadd(a, b);
    v[1] = csub(a, b);
  } else if (R == 4) {
    bf4(v[0], v[1], v[2], v[3], sgn);
  } else if (R == 3) {
    const real_t s60 = 0.86602540378443864676;
    real2 t1 = cadd(v[1], v[2]);
    real2 t2 = mk2(v[0].x - (real_t)0.5 * t1.x, v[0].y - (real_t)0.5 * t1.y);
    real2 t3 = cscale(cmul_mi(csub(v[1], v[2]), sgn), s60);
    v[0] = cadd(v[0], t1);
    v[1] = cadd(t2, t3);
    v[2] = csub(t2, t3);
  } else if (R == 5) {
    const real_t c1 = 0.30901699437494742410, c2 = -0.80901699437494742410;
    const real_t s1 = 0.95105651629515357212, s2 = 0.58778525229247312917;
    real2 a1 = cadd(v[1], v[4]), a2 = cadd(v[2], v[3]), d1 = csub(v[1], v[4]), real2_ = csub(v[2], v[3]);
    real2 r1 = mk2(v[0].x + c1 * a1.x + c2 * a2.x, v[0].y + c1 * a1.y + c2 * a2.y);
    real2 r2 = mk2(v[0].x + c2 * a1.x + c1 * a2.x, v[0].y + c2 * a1.y + c1 * a2.y);
    real2 q1 = cmul_mi(mk2(s1 * d1.x + s2 * real2_.x, s1 * d1.y + s2 * real2_.y), sgn);
    real2 q2 = cmul_mi(mk2(s2 * d1.x - s1 * real2_.x, s2 * d1.y - s1 * real2_.y), sgn);
    v[0] = cadd(v[0], cadd(a1, a2));
    v[1] = cadd(r1, q1);
    v[4] = csub(r1, q1);
    v[2] = cadd(r2, q2);
    v[3] = csub(r2, q2);
  } else if (R == 8) {
    // n = 4 n1 + n2, k = k1 + 2 k2
    const real_t h = 0.70710678118654752440;
    real2 t0[4], t1[4];
#pragma unroll
    for (int n2 = 0; n2 < 4; n2++) {
      t0[n2] = cadd(v[n2], v[4 + n2]);
      t1[n2] = csub(v[n2], v[4 + n2]);
    }
    // t1[n2] *= W8^{n2}
    t1[1] = (sgn < 0) ? mk2(h * (t1[1].x + t1[1].y), h * (t1[1].y - t1[1].x)) : mk2(h * (t1[1].x - t1[1].y), h * (t1[1].y + t1[1].x));
    t1[2] = cmul_mi(t1[2], sgn);
    t1[3] = (sgn < 0) ? mk2(h * (t1[3].y - t1[3].x), -h * (t1[3].x + t1[3].y)) : mk2(-h * (t1[3].x + t1[3].y), h * (t1[3].x - t1[3].y));
    bf4(t0[0], t0[1], t0[2], t0[3], sgn);
    bf4(t1[0], t1[1], t1[2], t1[3], sgn);
#pragma unroll
    for (int k2 = 0; k2 < 4; k2++) {
      v[2 * k2] = t0[k2];
      v[2 * k2 + 1] = t1[k2];
    }
  } else if constexpr (R == 6 || R == 9 || R == 10) {
    // composite radices of the specialised kernels (work lengths 3072 = 8^3 * 6, 4608 = 8^3 * 9,
    // 5120 = 8^3 * 10: four LDS round trips instead of five): Cooley-Tukey inside the registers,
    // n = R2 n1 + n2, k = k1 + R1 k2 with constant twiddles W_R^{n2 k1}
    constexpr int R1 = (R == 10) ? 5 : 3, R2 = R / R1;
    real2 t[R2][R1];
#pragma unroll
    for (int n2 = 0; n2 < R2; n2++) {
#pragma unroll
      for (int n1 = 0; n1 < R1; n1++) t[n2][n1] = v[R2 * n1 + n2];
      butterfly<R1>(t[n2], tw, S, sgn);
    }
    // cos / sin of 2 pi j / R, j = 0 .. 4 (all that n2 k1 reaches)
    constexpr real_t c6[5] = {1.0, 0.5, -0.5, -1.0, -0.5}, s6[5] = {0.0, 0.86602540378443864676, 0.86602540378443864676, 0.0, -0.86602540378443864676};
    constexpr real_t c9[5] = {1.0, 0.76604444311897803520, 0.17364817766693034885, -0.5, -0.93969262078590838405};
    constexpr real_t s9[5] = {0.0, 0.64278760968653932632, 0.98480775301220805937, 0.86602540378443864676, 0.34202014332566873304};
    constexpr real_t c10[5] = {1.0, 0.80901699437494742410, 0.30901699437494742410, -0.30901699437494742410, -0.80901699437494742410};
    constexpr real_t s10[5] = {0.0, 0.58778525229247312917, 0.95105651629515357212, 0.95105651629515357212, 0.58778525229247312917};
#pragma unroll
    for (int n2 = 1; n2 < R2; n2++)
#pragma unroll
      for (int k1 = 1; k1 < R1; k1++) {
        const int j = n2 * k1;
        const real_t c = (R == 6) ? c6[j] : (R == 9) ? c9[j] : c10[j];
        const real_t sn = ((R == 6) ? s6[j] : (R == 9) ? s9[j] : s10[j]) * (real_t)sgn;  // forward: exp(-i..)
        const real2 a = t[n2][k1];
        t[n2][k1] = mk2(a.x * c - a.y * sn, a.x * sn + a.y * c);
      }
#pragma unroll
    for (int k1 = 0; k1 < R1; k1++) {
      real2 u[R2];
#pragma unroll
      for (int n2 = 0; n2 < R2; n2++) u[n2] = t[n2][k1];
      butterfly<R2>(u, tw, S, sgn);
#pragma unroll
      for (int k2 = 0; k2 < R2; k2++) v[k1 + R1 * k2] = u[k2];
    }
  } else {
    // generic small prime (7): DFT matrix rows from the twiddle table, fully unrolled
    real2 y[R], w[R];
    const int st = S / R;
#pragma unroll
    for (int e = 1; e < R; e++) w[e] = tw_get(tw, e * st, sgn);
#pragma unroll
    for (int u = 0; u < R; u++) {
      real2 s = v[0];
#pragma unroll
      for (int t = 1; t < R; t++) s = ((u * t) % R) ? cadd(s, cmul(v[t], w[(u * t) % R])) : cadd(s, v[t]);
      y[u] = s;
    }
#pragma unroll
    for (int u = 0; u < R; u++) v[u] = y[u];
  }
}

// Radix-8 / radix-4 butterflies whose upper half of the inputs is zero (the first DIF pass of a Bluestein convolution:
// the chirped row fills less than half of the work array): the first layer of adds degenerates to copies.  The
// results are those of butterfly<R> on (v[0..R/2-1], 0, ..., 0) (x + 0 and x - 0 are x).
template <int R>
EMI_DEVFN void butterfly_hz(real2 *v, int sgn) {
  if (R == 4) {
    const real2 x0 = v[0], x1 = v[1], d = cmul_mi(x1, sgn);
    v[0] = cadd(x0, x1);
    v[1] = cadd(x0, d);
    v[2] = csub(x0, x1);
    v[3] = csub(x0, d);
  } else {  // R == 8
    const real_t h = 0.70710678118654752440;
    real2 t0[4], t1[4];
#pragma unroll
    for (int n2 = 0; n2 < 4; n2++) t0[n2] = t1[n2] = v[n2];
    t1[1] = (sgn < 0) ? mk2(h * (t1[1].x + t1[1].y), h * (t1[1].y - t1[1].x)) : mk2(h * (t1[1].x - t1[1].y), h * (t1[1].y + t1[1].x));
    t1[2] = cmul_mi(t1[2], sgn);
    t1[3] = (sgn < 0) ? mk2(h * (t1[3].y - t1[3].x), -h * (t1[3].x + t1[3].y)) : mk2(-h * (t1[3].x + t1[3].y), h * (t1[3].x - t1[3].y));
    bf4(t0[0], t0[1], t0[2], t0[3], sgn);
    bf4(t1[0], t1[1], t1[2], t1[3], sgn);
#pragma unroll
    for (int k2 = 0; k2 < 4; k2++) {
      v[2 * k2] = t0[k2];
      v[2 * k2 + 1] = t1[k2];
    }
  }
}

// split a butterfly number q into (block, j) for stride lenp (power of two when sh >= 0)
EMI_DEVFN void split_q(int q, int lenp, int sh, int &blk, int &j) {
  if (sh >= 0) {
    blk = q >> sh;
    j = q & (lenp - 1);
  } else {
    blk = q / lenp;
    j = q - blk * lenp;
  }
}
EMI_DEVFN int log2_exact(int v) { return (v & (v - 1)) ? -1 : (31 - __builtin_clz((unsigned)v)); }

// one in-place pass over nfl fields.  MASK: logical elements >= nvalid read as zero (only the first
// DIF pass of a zero-padded Bluestein input).  TW: the pass has inter-pass twiddles (lenp > 1).
// The R-1 twiddles W_len^{j t} are fetched into registers first -- coalesced reads of the per-pass
// table [t-1][j], all in flight together with the LDS reads -- and applied before (DIT) or after
// (DIF) the butterfly.
// one butterfly q of one field
// NOUT: outputs t < NOUT are stored (the last pass of a Bluestein convolution, of which only the first sz <= (S+1)/2
// elements are ever read: the rest of the butterfly is dead code)
// NZ: inputs t >= NZ are zero by construction and are not read (first DIF pass of a Bluestein convolution)
template <int R, int DIF, int MASK, int TW, int NOUT = R, int NZ = R>
EMI_DEVFN void fft_bfly_at(real2 *af, int q, int S, int lenp, int sh, const real2 *tw, const real2 *ptw, int sgn, int nvalid) {
  const int len = lenp * R;
  int blk, j;
  if (R == 2 || R == 4 || R == 8) {  // power-of-two radices come first: their lenp is a power of two
    blk = q >> sh;
    j = q & (lenp - 1);
  } else {
    split_q(q, lenp, sh, blk, j);
  }
  const int base = blk * len + j;
  real2 w[R];
  if (TW) {
    const real2 *pw_ = ptw + j;
#pragma unroll
    for (int t = 1; t < R; t++) w[t] = pw_[(t - 1) * lenp];
  }
  real2 v[R];
#pragma unroll
  for (int t = 0; t < R; t++) {
    const int i = base + t * lenp;
    if (t >= NZ)
      v[t] = mk2(0.0, 0.0);
    else if (MASK)
      v[t] = (i < nvalid) ? af[FPAD(i)] : mk2(0.0, 0.0);
    else
      v[t] = af[FPAD(i)];
  }
  if (TW) {
    if (sgn > 0) {
#pragma unroll
      for (int t = 1; t < R; t++) w[t].y = -w[t].y;
    }
  }
  if (TW && !DIF) {
#pragma unroll
    for (int t = 1; t < R; t++) v[t] = cmul(v[t], w[t]);
  }
  if constexpr (NZ < R && NZ == R / 2 && (R == 8 || R == 4))
    butterfly_hz<R>(v, sgn);
  else
    butterfly<R>(v, tw, S, sgn);
  if (TW && DIF) {
#pragma unroll
    for (int t = 1; t < R; t++) v[t] = cmul(v[t], w[t]);
  }
#pragma unroll
  for (int t = 0; t < NOUT; t++) af[FPAD(base + t * lenp)] = v[t];
}
// FLAT = 0: field after field, butterfly q = tid + i * nthreads of each (long rows: every sweep is full
// anyway).  FLAT = 1: the (field, butterfly) pairs of the workgroup are dealt to the threads as one list,
// so short rows (fewer butterflies than threads) still fill the sweeps.
template <int R, int DIF, int MASK, int TW, int FLAT = 0, int NOUT = R, int NZ = R>
EMI_DEVFN void fft_pass_body(real2 *a, int nfl, int fstride, int S, int lenp, const real2 *tw, const real2 *ptw, int sgn, int nvalid) {
  const int nb = S / R, sh = log2_exact(lenp);
  if (FLAT) {
    for (int idx = EMI_TID; idx < nfl * nb; idx += EMI_NTHREADS) {
      const int fl = idx / nb, q = idx - fl * nb;
      fft_bfly_at<R, DIF, MASK, TW, NOUT, NZ>(a + (long long)fl * fstride, q, S, lenp, sh, tw, ptw, sgn, nvalid);
    }
  } else {
    for (int fl = 0; fl < nfl; fl++) {
      real2 *af = a + (long long)fl * fstride;
      for (int q = EMI_TID; q < nb; q += EMI_NTHREADS) fft_bfly_at<R, DIF, MASK, TW, NOUT, NZ>(af, q, S, lenp, sh, tw, ptw, sgn, nvalid);
    }
  }
}
template <int R, int DIF, int MASK>
EMI_DEVFN void fft_pass(real2 *a, int nfl, int fstride, int S, int lenp, const real2 *tw, const real2 *ptw, int sgn, int nvalid) {
  if (lenp > 1)
    fft_pass_body<R, DIF, MASK, 1>(a, nfl, fstride, S, lenp, tw, ptw, sgn, nvalid);
  else
    fft_pass_body<R, DIF, MASK, 0>(a, nfl, fstride, S, lenp, tw, ptw, sgn, nvalid);
}

#define FFT_DISPATCH(FN, r, ...)                  \
  switch (r) {                                    \
    case 2: FN<2>(__VA_ARGS__); break;            \
    case 3: FN<3>(__VA_ARGS__); break;            \
    case 4: FN<4>(__VA_ARGS__); break;            \
    case 5: FN<5>(__VA_ARGS__); break;            \
    case 7: FN<7>(__VA_ARGS__); break;            \
    case 8: FN<8>(__VA_ARGS__); break;            \
    default: break;                               \
  }
// Bluestein lengths are 2^a * {1,3,5,9,15}: their first DIT factor is a power of two
#define FFT_DISPATCH_POW2(FN, r, ...)             \
  switch (r) {                                    \
    case 2: FN<2>(__VA_ARGS__); break;            \
    case 4: FN<4>(__VA_ARGS__); break;            \
    case 8: FN<8>(__VA_ARGS__); break;            \
    default: break;                               \
  }

template <int R>
EMI_DEVFN void pass_dit(real2 *a, int nfl, int fs, int S, int lenp, const real2 *tw, const real2 *ptw, int sgn, int nvalid) {
  fft_pass<R, 0, 0>(a, nfl, fs, S, lenp, tw, ptw, sgn, nvalid);
}
template <int R>
EMI_DEVFN void pass_dif(real2 *a, int nfl, int fs, int S, int lenp, const real2 *tw, const real2 *ptw, int sgn, int nvalid) {
  if (nvalid < S)
    fft_pass<R, 1, 1>(a, nfl, fs, S, lenp, tw, ptw, sgn, nvalid);
  else
    fft_pass<R, 1, 0>(a, nfl, fs, S, lenp, tw, ptw, sgn, nvalid);
}

// DIT passes ip = first..last-1 (factor order); returns lenp after them
EMI_DEVFN int run_dit(real2 *a, int nfl, int fs, int S, const FftPlanDev &pl, const FftTabDev &T, int first, int last, int lenp, int sgn) {
  const int *fac = pl.fac;
  const real2 *tw = (const real2 *)T.tw + pl.tw_off;
  for (int ip = first; ip < last; ip++) {
    const int r = fac[ip];
    FFT_DISPATCH(pass_dit, r, a, nfl, fs, S, lenp, tw, (const real2 *)T.ptw + pl.ptw_off[ip], sgn, S);
    lenp *= r;
    EMI_SYNC();
  }
  return lenp;
}
// DIF passes over factors nfac-1 down to `stop` (inclusive); nvalid applies to the first one
EMI_DEVFN void run_dif(real2 *a, int nfl, int fs, int S, const FftPlanDev &pl, const FftTabDev &T, int stop, int sgn, int nvalid) {
  const int *fac = pl.fac;
  const real2 *tw = (const real2 *)T.tw + pl.tw_off;
  int lenp = S;
  for (int ip = pl.nfac - 1; ip >= stop; ip--) {
    const int r = fac[ip];
    lenp /= r;
    FFT_DISPATCH(pass_dif, r, a, nfl, fs, S, lenp, tw, (const real2 *)T.ptw + pl.ptw_off[ip], sgn, nvalid);
    nvalid = S;
    EMI_SYNC();
  }
}

// Bluestein middle: last DIF pass (radix fac[0], contiguous runs) * filter * first DIT pass
template <int R>
EMI_DEVFN void blue_middle_at(real2 *af, int q, int nb, int S, const real2 *tw, const real2 *bh, int conj_b, int nvalid) {
  const int base = q * R;
  real2 v[R], b[R];
#pragma unroll
  for (int t = 0; t < R; t++) b[t] = bh[t * nb + q];  // filter values, table [t][q]: coalesced, in flight with the LDS reads
#pragma unroll
  for (int t = 0; t < R; t++) v[t] = (base + t < nvalid) ? af[FPAD(base + t)] : mk2(0.0, 0.0);
  butterfly<R>(v, tw, S, -1);
#pragma unroll
  for (int t = 0; t < R; t++) v[t] = conj_b ? cmulc(v[t], b[t]) : cmul(v[t], b[t]);
  butterfly<R>(v, tw, S, +1);
#pragma unroll
  for (int t = 0; t < R; t++) af[FPAD(base + t)] = v[t];
}
template <int R, int FLAT = 0>
EMI_DEVFN void blue_middle(real2 *a, int nfl, int fs, int S, const real2 *tw, const real2 *bh, int conj_b, int nvalid) {
  const int nb = S / R;
  if (FLAT) {
    for (int idx = EMI_TID; idx < nfl * nb; idx += EMI_NTHREADS) {
      const int fl = idx / nb, q = idx - fl * nb;
      blue_middle_at<R>(a + (long long)fl * fs, q, nb, S, tw, bh, conj_b, nvalid);
    }
  } else {
    for (int fl = 0; fl < nfl; fl++) {
      real2 *af = a + (long long)fl * fs;
      for (int q = EMI_TID; q < nb; q += EMI_NTHREADS) blue_middle_at<R>(af, q, nb, S, tw, bh, conj_b, nvalid);
    }
  }
}

// circular convolution with the chirp filter: a (natural, logically zero beyond nvalid) -> natural,
// leaving the LAST DIT pass to the caller (returns its lenp) when defer_last != 0.
EMI_DEVFN int blue_conv(real2 *a, int nfl, int fs, const FftPlanDev &pl, const FftTabDev &T, int conj_b, int nvalid, int defer_last) {
  const int L = pl.S;
  const real2 *tw = (const real2 *)T.tw + pl.tw_off, *bh = (const real2 *)T.bhat + pl.bhat_off;
  run_dif(a, nfl, fs, L, pl, T, 1, -1, nvalid);
  const int r0 = pl.fac[0];
  const int nv0 = (pl.nfac == 1) ? nvalid : L;
  FFT_DISPATCH_POW2(blue_middle, r0, a, nfl, fs, L, tw, bh, conj_b, nv0);
  EMI_SYNC();
  const int last = defer_last ? pl.nfac - 1 : pl.nfac;
  return run_dit(a, nfl, fs, L, pl, T, 1, last < 1 ? 1 : last, r0, +1);
}

// ---- specialised kernels: butterfly legs by BYTE offset into the workgroup's LDS block ------------------------
// The FFT kernels are bound by vector-instruction issue (tools/valu_lds_probe.hip: 16 waves per CU overlap their LDS
// traffic with the arithmetic of the others almost completely, so a pass costs what its vector instructions cost), and a
// third of the vector instructions of a pass used to be address arithmetic: a shift, an AND, an XOR and a scaled add per
// leg for FPAD(base + t lenp), 64-bit adds for the twiddle addresses.  With the stride known at compile time the
// swizzled position of leg t is one XOR with a constant, or an immediate offset of the LDS instruction:
//   LENP = 1  (R = 8, base = 8 q):          FPAD(base) ^ t
//   LENP = 8  (R = 8, base = 64 b + j):     (base ^ 8 (b & 1)) ^ 9 t
//   LENP = 64:                              FPAD(base) ^ 8 (t & 1), + 64 t
//   LENP a multiple of 128:                 FPAD(base) + t LENP
// (a field's array starts at a multiple of 64 complex numbers -- every work length is one -- so the XORs, which stay
// below that, act on the byte offset), and the twiddles of leg t are read through a per-leg uniform base (scalar registers) plus the
// 32-bit lane offset.
template <int R, int LENP>
struct HotLegs {
  static constexpr int KIND = (LENP == 1 && R == 8) ? 1 : (LENP == 8 && R == 8) ? 2 : (LENP == 64) ? 3 : (LENP % 128 == 0) ? 4 : 0;
  static constexpr int SH = (sizeof(real2) == 16) ? 4 : 3;  // log2 of the bytes per complex number
  int b0, b1;
  EMI_DEVFN HotLegs(int fo, int base) {
    if constexpr (KIND == 2)
      b0 = fo + ((base ^ ((base >> 3) & 8)) << SH);
    else if constexpr (KIND == 0)
      b0 = base;  // element index; the field offset is added in at()
    else
      b0 = fo + (FPAD(base) << SH);
    b1 = (KIND == 3) ? (b0 ^ (8 << SH)) : fo;
  }
  EMI_DEVFN int at(int t) const {
    if constexpr (KIND == 1)
      return b0 ^ (t << SH);
    else if constexpr (KIND == 2)
      return b0 ^ ((9 * t) << SH);
    else if constexpr (KIND == 3)
      return ((t & 1) ? b1 : b0) + ((64 * t) << SH);
    else if constexpr (KIND == 4)
      return b0 + ((t * LENP) << SH);
    else
      return b1 + (FPAD(b0 + t * LENP) << SH);
  }
};
EMI_DEVFN real2 hot_ld(int boff) {
  EMI_LDS_DECL;
  return *(const real2 *)(EMI_LDS_PTR + boff);
}
EMI_DEVFN void hot_st(int boff, real2 v) {
  EMI_LDS_DECL;
  *(real2 *)(EMI_LDS_PTR + boff) = v;
}

// Grid arrays are blocked (ngpblks, nfld, nproma).  One latitude row of one field starts at point
// gp0 of the task's grid; GridRow resolves the NPROMA block of gp0 once (the only 64-bit division),
// every element then costs an add and a compare unless the row crosses a block boundary.
struct GridRow {
  real_t *p0;         // first element of (block of gp0, this field)
  unsigned rem0, np;  // gp0's offset inside its block; NPROMA
  long long bstride;  // elements from one block of this field to the next: nf_arr * NPROMA
};
EMI_DEVFN GridRow grid_row(const GridFld &gf, long long gp0, int nproma) {
  GridRow r;
  const long long blk0 = gp0 / nproma;
  r.rem0 = (unsigned)(gp0 - blk0 * nproma);
  r.np = (unsigned)nproma;
  r.bstride = (long long)gf.nf_arr * nproma;
  r.p0 = (real_t *)gf.base + (blk0 * gf.nf_arr + gf.fidx) * (long long)nproma;
  return r;
}
EMI_DEVFN real_t *grid_ptr(const GridRow &r, unsigned o) {  // o: point number within the latitude row
  unsigned q = r.rem0 + o;
  if (q < r.np) return r.p0 + q;
  const unsigned b = q / r.np;
  return r.p0 + (long long)b * r.bstride + (q - b * r.np);
}
// points o, o+1 contiguous and 2-element aligned: one real2 access
EMI_DEVFN bool grid_pair_ok(const GridRow &r, unsigned o) {
  return (r.rem0 + o + 1 < r.np) && ((((uintptr_t)(r.p0 + r.rem0 + o)) & (2 * sizeof(real_t) - 1)) == 0);
}

// final DIT pass of the inverse real transform, stored straight to the grid array:
// logical output z_i (i < sz): x_{2i} = Re, x_{2i+1} = Im (or x_i = Re z_i in complex mode),
// Bluestein: z_i = a'_i * conj(chirp_i) / L.
// BLUE (specialised kernels: always Bluestein): i = j + t lenp < sz <= (S+1)/2 only for t < (R+1)/2 -- the other
// outputs, their chirp values and the part of the butterfly that only feeds them are never generated.
// LENP > 0 (specialised kernels): the stride at compile time -- legs by byte offset (HotLegs), tables through per-leg
// uniform bases
template <int R, int BLUE = 0, int LENP = 0>
EMI_DEVFN void dit_last_to_grid(real2 *a, int nfl, int fs, int S, int lenp, const real2 *tw, const real2 *ptw, const FftPlanDev &pl,
                                const real2 *chirp, const GridFld *flds, int f0, long long gp0, int nproma) {
  constexpr int NOUT = BLUE ? (R + 1) / 2 : R;
  const int nb = S / R, sh = log2_exact(lenp), sz = pl.sz;
  const real_t invL = pl.blue ? (real_t)(1.0 / (double)S) : (real_t)1.0;
  for (int fl = 0; fl < nfl; fl++) {
    real2 *af = a + (long long)fl * fs;
    const GridRow gr = grid_row(flds[f0 + fl], gp0, nproma);
    const bool flat = (gr.rem0 + (unsigned)pl.n <= gr.np) && ((((uintptr_t)(gr.p0 + gr.rem0)) & (2 * sizeof(real_t) - 1)) == 0);
    for (int q = EMI_TID; q < nb; q += EMI_NTHREADS) {
      int blk, j;
      split_q(q, lenp, sh, blk, j);  // last pass: len == S, blk == 0
      EMI_OPAQUE(j);
      real2 v[R], w[R], ch[NOUT];
      if (lenp > 1) {
#pragma unroll
        for (int t = 1; t < R; t++) w[t] = (ptw + (t - 1) * lenp)[(unsigned)j];
      }
      if constexpr (LENP > 0) {
        const HotLegs<R, LENP> L(fl * fs * (int)sizeof(real2), j);
#pragma unroll
        for (int t = 0; t < R; t++) v[t] = hot_ld(L.at(t));
      } else {
#pragma unroll
        for (int t = 0; t < R; t++) v[t] = af[FPAD(j + t * lenp)];
      }
      if (lenp > 1) {
#pragma unroll
        for (int t = 1; t < R; t++) v[t] = cmulc(v[t], w[t]);  // inverse: conjugate twiddles
      }
      if (pl.blue && R <= 8) {  // chirp values of the outputs: in flight during the butterfly
#pragma unroll
        for (int t = 0; t < NOUT; t++) ch[t] = (j + t * lenp < sz) ? (chirp + t * lenp)[(unsigned)j] : mk2(0.0, 0.0);
      }
      butterfly<R>(v, tw, S, +1);
      if (pl.blue && R > 8) {  // composite radices: no registers to spare during the butterfly
#pragma unroll
        for (int t = 0; t < NOUT; t++) ch[t] = (j + t * lenp < sz) ? (chirp + t * lenp)[(unsigned)j] : mk2(0.0, 0.0);
      }
#pragma unroll
      for (int t = 0; t < NOUT; t++) {
        const int i = j + t * lenp;
        if (i < sz) {
          real2 z = v[t];
          if (pl.blue) z = cscale(cmulc(z, ch[t]), invL);
          if (!pl.cmode) {
            if (flat) {  // whole row inside one NPROMA block and 2-element aligned (uniform)
              *(real2 *)(gr.p0 + gr.rem0 + 2u * i) = z;
            } else if (grid_pair_ok(gr, 2u * i)) {
              *(real2 *)grid_ptr(gr, 2u * i) = z;
            } else {
              *grid_ptr(gr, 2u * i) = z.x;
              *grid_ptr(gr, 2u * i + 1) = z.y;
            }
          } else {
            *grid_ptr(gr, (unsigned)i) = z.x;
          }
        }
      }
    }
  }
}

template <int R>
EMI_DEVFN void dit_last_to_grid_any(real2 *a, int nfl, int fs, int S, int lenp, const real2 *tw, const real2 *ptw, const FftPlanDev &pl,
                                    const real2 *chirp, const GridFld *flds, int f0, long long gp0, int nproma) {
  dit_last_to_grid<R, 0>(a, nfl, fs, S, lenp, tw, ptw, pl, chirp, flds, f0, gp0, nproma);
}

// ==========================================================================================
// k_fft_inv: FOURIER_IN (fourier_in_mod.F90:64-76) + FSC (fsc_mod.F90:138-187) + FTINV
// (ftinv_mod.F90:65-84; FFTW c2r semantics, unnormalised) + TRLTOG local copy.
// ==========================================================================================
// FSC (fsc_mod.F90:138-187: x (a + i b k), a / b from the field mode; DIR_TRANSAD: times the weight) and the pairing of FOURIER_IN for one
// pair (k, sz - k): Z_k = (X_k + conj X_{sz-k}) + i w^k (X_k - conj X_{sz-k}) and Z_{sz-k} (w^{sz-k} = -conj w^k).  ONE expression tree for
// the batched one-task paths and the row-table paths (several tasks) of the inverse kernels, so that both round alike and the gathered
// fields of any decomposition stay bit-identical (tests/test_decomposition_invariance.py, the reference's own criterion).
EMI_DEVFN void fin_factors(int mode, real_t racthe, real_t &fa, real_t &fb) {
  fa = mode == GM_PLAIN ? (real_t)1.0 : (mode == GM_ACOS ? racthe : (real_t)0.0);
  fb = mode == GM_EWDER ? racthe : (mode == GM_EWDER_UV ? racthe * racthe : (real_t)0.0);
}
EMI_DEVFN void fin_pair(real2 xa, real2 xb, unsigned k, unsigned k2, real_t fa, real_t fb, real_t fs, real2 wk, real2 ck, real2 ck2, real2 &zk, real2 &zk2) {
  EMI_FP_STRICT();  // every operation rounds on its own: which multiply a fused multiply-add would absorb is the scheduler's choice, per call site
  const real_t ba = fb * (real_t)k, bb = fb * (real_t)k2;
  const real2 ya = mk2((xa.x * fa - xa.y * ba) * fs, (xa.y * fa + xa.x * ba) * fs);
  const real2 yb = mk2((xb.x * fa - xb.y * bb) * fs, (xb.y * fa + xb.x * bb) * fs);
  const real_t s1x = ya.x + yb.x, s1y = ya.y - yb.y, d1x = ya.x - yb.x, d1y = ya.y + yb.y;  // s = ya + conj yb, d = ya - conj yb
  const real_t px = wk.x * d1x - wk.y * d1y, py = wk.x * d1y + wk.y * d1x;                   // w d
  const real_t zx = s1x - py, zy = s1y + px;                                                   // s + i w d
  zk = mk2(zx * ck.x + zy * ck.y, zy * ck.x - zx * ck.y);                                      // times conj(ck)
  const real_t s2x = yb.x + ya.x, s2y = yb.y - ya.y, d2x = yb.x - ya.x, d2y = yb.y + ya.y;
  const real_t qx = -wk.x * d2x - wk.y * d2y, qy = -wk.x * d2y + wk.y * d2x;                  // (-conj w) d
  const real_t ux = s2x - qy, uy = s2y + qx;
  zk2 = mk2(ux * ck2.x + uy * ck2.y, uy * ck2.x - ux * ck2.y);
}
EMI_DEVFN real2 fin_raw(const real_t *FB, int row, int ldf, int src) { return *(const real2 *)(FB + (unsigned long long)(unsigned)row * (unsigned)ldf + 2 * src); }
// row < 2^31 and ldf > 0: the row offset is one 32 x 32 -> 64-bit multiply
EMI_DEVFN real2 fsc_load(const real_t *FB, int row, int ldf, const GridFld &gf, int k, real_t racthe) {
  real2 x = *(const real2 *)(FB + (unsigned long long)(unsigned)row * (unsigned)ldf + 2 * gf.src);
  if (gf.mode == GM_ACOS)
    x = cscale(x, racthe);
  else if (gf.mode == GM_EWDER)
    x = cscale(cmuli(x), racthe * (real_t)k);
  else if (gf.mode == GM_EWDER_UV)
    x = cscale(cmuli(x), racthe * racthe * (real_t)k);
  return x;
}

// `a`: the work array of the workgroup -- its LDS block (k_fft_inv), or, for rows whose work array exceeds the 160 KiB of
// LDS (fp64 rows beyond 10240 complex points: TCo2559, or any caller grid), a slice of a global scratch buffer
// (k_fft_inv_gm); the passes are the same code, the workgroup barrier orders global accesses as it orders LDS ones
EMI_DEVFN void fft_inv_body(const EmiGeomDev &g, const FftTabDev &T, const FftLaunchDev &Lc, const GridFld *flds, int nfld, const real_t *FB,
                            int ldf, int nproma, real2 *a) {
  const int bid = (int)xcd_swizzle(EMI_BID, Lc.nblocks, 8);
  const int li = bid / Lc.nchunk;
  const int lat = Lc.lats[li];
  const FftPlanDev &pl = T.plans[T.planid[lat]];
  const int f0 = (bid - li * Lc.nchunk) * pl.fbk;
  const int nfl = (nfld - f0) < pl.fbk ? (nfld - f0) : pl.fbk;
  const int n = pl.n, sz = pl.sz, S = pl.S, nmen = g.nmen[lat];
  const int fs = FFT_LDS_ELEMS(S);
  const real_t racthe = (real_t)g.racthe[lat];
  const real_t adjw = (real_t)(g.rw[lat] / (double)pl.n);  // DIR_TRANSAD only (Lc.adj)
  // row of (lat, m=k) in the FFT-side buffer: affine for one task (no table load in front of the
  // data load), through the exchange-order table otherwise
  const int fb0 = g.fbase[lat];
  const int *frow = g.fftrow ? g.fftrow + fb0 : nullptr;
#define FROW(k_) (frow ? frow[k_] : fb0 + (k_))
  const real2 *tw = (const real2 *)T.tw + pl.tw_off;
  const unsigned short *perm = T.perm + pl.perm_off;
  const real2 *rtw = (const real2 *)T.rtw + pl.rtw_off;
  const real2 *chirp = (const real2 *)T.chirp + pl.chirp_off;

  // ---- stage 1: logical input Z_k, k in [0,sz), to LDS (natural for Bluestein, perm[] for DIT)
  for (int fl = 0; fl < nfl; fl++) {
    const GridFld gf = flds[f0 + fl];
    real2 *af = a + (long long)fl * fs;
    if (!pl.cmode) {
      const int npair = sz / 2 + 1;  // k = 0..sz/2 pairs with sz-k
      for (int k = EMI_TID; k < npair; k += EMI_NTHREADS) {
        const int k2 = sz - k;
        real2 xa = (k <= nmen) ? fsc_load(FB, FROW(k), ldf, gf, k, racthe) : mk2(0, 0);
        real2 xb = (k2 <= nmen) ? fsc_load(FB, FROW(k2), ldf, gf, k2, racthe) : mk2(0, 0);
        if (Lc.adj) xa = cscale(xa, adjw), xb = cscale(xb, adjw);
        // Z_k = (X_k + conj X_{sz-k}) + i w^k (X_k - conj X_{sz-k}),  w = exp(+2 pi i/n)
        real2 wk = cconj(rtw[k]);
        real2 s1 = cadd(xa, cconj(xb)), d1 = csub(xa, cconj(xb));
        real2 zk = cadd(s1, cmuli(cmul(wk, d1)));
        af[FPAD(pl.blue ? k : (int)perm[k])] = pl.blue ? cmulc(zk, chirp[k]) : zk;
        if (k2 != k && k2 < sz) {
          // Z_{sz-k}: w^{sz-k} = -conj(w^k)
          real2 s2 = cadd(xb, cconj(xa)), real2_ = csub(xb, cconj(xa));
          real2 zk2 = cadd(s2, cmuli(cmul(mk2(-wk.x, wk.y), real2_)));
          af[FPAD(pl.blue ? k2 : (int)perm[k2])] = pl.blue ? cmulc(zk2, chirp[k2]) : zk2;
        }
      }
    } else {
      // complex mode (odd n): Z_k = X_k, Z_{n-k} = conj X_k
      for (int k = EMI_TID; k < sz; k += EMI_NTHREADS) {
        real2 z;
        if (2 * k <= n) {
          z = (k <= nmen) ? fsc_load(FB, FROW(k), ldf, gf, k, racthe) : mk2(0, 0);
          if (k == 0) z.y = 0.0;
        } else {
          z = (n - k <= nmen) ? cconj(fsc_load(FB, FROW(n - k), ldf, gf, n - k, racthe)) : mk2(0, 0);
        }
        if (Lc.adj) z = cscale(z, adjw);
        af[FPAD(pl.blue ? k : (int)perm[k])] = pl.blue ? cmulc(z, chirp[k]) : z;
      }
    }
  }
  EMI_SYNC();
  // ---- stage 2 (all passes but the last) and stage 3 (last DIT pass -> grid, TRLTOG local copy)
  const long long gp0 = g.gpoff[lat];
  int lenp;
  if (pl.blue) {
    if (pl.nfac == 1) {
      // degenerate single-factor filter length: no separate last pass to fuse with the store
      blue_conv(a, nfl, fs, pl, T, 1, sz, 0);
      const real_t invL = (real_t)(1.0 / (double)S);
      for (int fl = 0; fl < nfl; fl++) {
        const GridRow gr = grid_row(flds[f0 + fl], gp0, nproma);
        const real2 *af = a + (long long)fl * fs;
        for (int p = EMI_TID; p < n; p += EMI_NTHREADS) {
          const int i = pl.cmode ? p : (p >> 1);
          real2 z = cscale(cmulc(af[FPAD(i)], chirp[i]), invL);
          *grid_ptr(gr, (unsigned)p) = (pl.cmode || !(p & 1)) ? z.x : z.y;
        }
      }
      return;
    }
    lenp = blue_conv(a, nfl, fs, pl, T, 1, sz, 1);
  } else {
    if (pl.nfac == 0) {  // sz == 1
      for (int fl = EMI_TID; fl < nfl; fl += EMI_NTHREADS) {
        const GridRow gr = grid_row(flds[f0 + fl], gp0, nproma);
        real2 z = a[(long long)fl * fs];
        *grid_ptr(gr, 0u) = z.x;
        if (!pl.cmode) *grid_ptr(gr, 1u) = z.y;
      }
      return;
    }
    lenp = run_dit(a, nfl, fs, S, pl, T, 0, pl.nfac - 1, 1, +1);
  }
  const int rl = pl.fac[pl.nfac - 1];
  FFT_DISPATCH(dit_last_to_grid_any, rl, a, nfl, fs, S, lenp, tw, (const real2 *)T.ptw + pl.ptw_off[pl.nfac - 1], pl, chirp, flds, f0, gp0, nproma);
}

EMI_KERNEL_FFT(EMI_FFT_WAVES) void k_fft_inv(EmiGeomDev g, FftTabDev T, FftLaunchDev Lc, const GridFld *flds, int nfld, const real_t *FB,
                              int ldf, int nproma) {
  EMI_LDS_DECL;
  fft_inv_body(g, T, Lc, flds, nfld, FB, ldf, nproma, (real2 *)EMI_LDS_PTR);
}
EMI_KERNEL_FFT(EMI_FFT_WAVES) void k_fft_inv_gm(EmiGeomDev g, FftTabDev T, FftLaunchDev Lc, const GridFld *flds, int nfld, const real_t *FB,
                                 int ldf, int nproma, real2 *scratch, long long stride) {
  fft_inv_body(g, T, Lc, flds, nfld, FB, ldf, nproma, scratch + (long long)EMI_BID * stride);
}

// ==========================================================================================
// k_fft_dir: TRGTOL local copy + FTDIR (ftdir_mod.F90:67-84; r2c, scaled 1/NLOEN at
// tpm_fftw.F90:317-321) + FOURIER_OUT (fourier_out_mod.F90:64-76).  The Gaussian weight
// (ledir_mod.F90:118-124) and LDFOU2's 1/(a cos) (ldfou2_mod.F90:90-96) only depend on the
// latitude and are folded into the same scale factor.
// ==========================================================================================
EMI_DEVFN real_t fft_dir_mode_scale(int mode, real_t racthe) {
  return mode == GM_PLAIN ? (real_t)1.0 : (mode == GM_EWDER_UV ? racthe * racthe : racthe);
}
EMI_DEVFN void fft_dir_body(const EmiGeomDev &g, const FftTabDev &T, const FftLaunchDev &Lc, const GridFld *flds, int nfld, real_t *FB, int ldf,
                            int nproma, real2 *a) {
  const int bid = (int)xcd_swizzle(EMI_BID, Lc.nblocks, 8);
  const int li = bid / Lc.nchunk;
  const int lat = Lc.lats[li];
  const FftPlanDev &pl = T.plans[T.planid[lat]];
  const int f0 = (bid - li * Lc.nchunk) * pl.fbk;
  const int nfl = (nfld - f0) < pl.fbk ? (nfld - f0) : pl.fbk;
  const int n = pl.n, sz = pl.sz, S = pl.S, nmen = g.nmen[lat];
  const int fs = FFT_LDS_ELEMS(S);
  const int fb0 = g.fbase[lat];
  const int *frow = g.fftrow ? g.fftrow + fb0 : nullptr;
  const long long gp0 = g.gpoff[lat];
  const real2 *tw = (const real2 *)T.tw + pl.tw_off;
  const unsigned short *perm = T.perm + pl.perm_off;
  const real2 *rtw = (const real2 *)T.rtw + pl.rtw_off;
  const real2 *chirp = (const real2 *)T.chirp + pl.chirp_off;

  // ---- stage 1: z_l = x_{2l} + i x_{2l+1} (or x_l in complex mode)
  for (int fl = 0; fl < nfl; fl++) {
    const GridFld gf = flds[f0 + fl];
    real2 *af = a + (long long)fl * fs;
    const GridRow gr = grid_row(gf, gp0, nproma);
    const bool flat = (gr.rem0 + (unsigned)n <= gr.np) && ((((uintptr_t)(gr.p0 + gr.rem0)) & (2 * sizeof(real_t) - 1)) == 0);
    for (int lz = EMI_TID; lz < sz; lz += EMI_NTHREADS) {
      real2 z;
      if (!pl.cmode) {
        if (flat) {
          z = *(const real2 *)(gr.p0 + gr.rem0 + 2u * lz);
        } else if (grid_pair_ok(gr, 2u * lz)) {
          z = *(const real2 *)grid_ptr(gr, 2u * lz);
        } else {
          z.x = *grid_ptr(gr, 2u * lz);
          z.y = *grid_ptr(gr, 2u * lz + 1);
        }
      } else {
        z.x = *grid_ptr(gr, (unsigned)lz);
        z.y = 0.0;
      }
      af[FPAD(pl.blue ? lz : (int)perm[lz])] = pl.blue ? cmul(z, chirp[lz]) : z;
    }
  }
  EMI_SYNC();
  if (pl.blue)
    blue_conv(a, nfl, fs, pl, T, 0, sz, 0);
  else
    run_dit(a, nfl, fs, S, pl, T, 0, pl.nfac, 1, -1);
  // ---- stage 3: X_k, k = 0..NMEN
  const real_t invL = pl.blue ? (real_t)(1.0 / (double)S) : (real_t)1.0;
  const real_t base_scale = Lc.adj ? (real_t)1.0 : (real_t)(g.rw[lat] / (double)n);
  for (int fl = 0; fl < nfl; fl++) {
    const GridFld gf = flds[f0 + fl];
    const real2 *af = a + (long long)fl * fs;
    // 1 / (a cos): the wind fields (LDFOU2) and, in INV_TRANSAD, the adjoints of the derivative outputs of FSC (once
    // for the north-south and east-west derivatives of scalars, twice for the east-west derivatives of u, v)
    const real_t sc = base_scale * fft_dir_mode_scale(gf.mode, (real_t)g.racthe[lat]);
    for (int k = EMI_TID; k <= nmen; k += EMI_NTHREADS) {
      real2 x;
      if (!pl.cmode) {
        const int kb = (k == 0) ? 0 : sz - k;
        real2 za = af[FPAD(k)], zb = af[FPAD(kb)];
        if (pl.blue) {
          za = cscale(cmul(za, chirp[k]), invL);
          zb = cscale(cmul(zb, chirp[kb]), invL);
        }
        // X_k = 1/2 [ (Z_k + conj Z_{sz-k}) - i exp(-2 pi i k/n) (Z_k - conj Z_{sz-k}) ]
        real2 s1 = cadd(za, cconj(zb)), d1 = csub(za, cconj(zb));
        real2 t = cmuli(cmul(rtw[k], d1));
        x = mk2((real_t)0.5 * (s1.x - t.x), (real_t)0.5 * (s1.y - t.y));
      } else {
        x = af[FPAD(k)];
        if (pl.blue) x = cscale(cmul(x, chirp[k]), invL);
      }
      *(real2 *)(FB + (unsigned long long)(unsigned)FROW(k) * (unsigned)ldf + 2 * (f0 + fl)) = cscale(x, sc);
    }
  }
}

EMI_KERNEL_FFT(EMI_FFT_WAVES) void k_fft_dir(EmiGeomDev g, FftTabDev T, FftLaunchDev Lc, const GridFld *flds, int nfld, real_t *FB, int ldf,
                              int nproma) {
  EMI_LDS_DECL;
  fft_dir_body(g, T, Lc, flds, nfld, FB, ldf, nproma, (real2 *)EMI_LDS_PTR);
}
EMI_KERNEL_FFT(EMI_FFT_WAVES) void k_fft_dir_gm(EmiGeomDev g, FftTabDev T, FftLaunchDev Lc, const GridFld *flds, int nfld, real_t *FB, int ldf,
                                 int nproma, real2 *scratch, long long stride) {
  fft_dir_body(g, T, Lc, flds, nfld, FB, ldf, nproma, scratch + (long long)EMI_BID * stride);
}

// ==========================================================================================
// Specialised FFT kernels for the work lengths that carry an octahedral grid (93 % of its
// rows, by weight, are Bluestein rows; at TCo1279 all of the long ones use one of six lengths):
// Bluestein, even NLOEN, one field per workgroup, factor list known at compile time.  Same
// arithmetic, pass for pass, as the generic kernels above (results agree to fma-contraction
// rounding, tests/test_gpu_parity.py); what changes is that the pass sequence is straight-line code with constant strides -- no radix
// dispatch, no plan loops, a fraction of the scalar registers -- so nothing spills.
// ==========================================================================================
struct HotPlanC {
  int S, nfac, fac[5], nfl;
};
EMI_DEVFN constexpr HotPlanC hot_plan(int pc) {
  switch (pc) {
#define EMI_HOT_CASE(pc_, S_, nf_, a_, b_, c_, d_, e_, nfl_) \
  case pc_: return {S_, nf_, {a_, b_, c_, d_, e_}, nfl_};
    EMI_HOT_PLAN_LIST(EMI_HOT_CASE)
#undef EMI_HOT_CASE
    default: return {0, 0, {1, 1, 1, 1, 1}, 1};
  }
}
EMI_DEVFN constexpr int hot_lenp(int pc, int ip) {  // stride of factor ip = product of the factors before it
  int l = 1;
  for (int i = 0; i < ip; i++) l *= hot_plan(pc).fac[i];
  return l;
}

// one butterfly q of the field at byte offset fo: fft_bfly_at with the stride and the work length known at compile time.
// Round 5: the twiddle table of the pass comes through a buffer descriptor -- the lane offset j is one 32-bit register for all legs and
// the leg offsets (t - 1) LENP are scalar operands of the load, where the flat loads needed a 64-bit add (two vector instructions)
// per leg beyond the 4 KiB reach of their immediate offsets (SQ counters, profiles/r5a_pmc_fft.txt: 29 - 31 % of the vector
// instructions of these kernels were not floating point).
template <int R, int LENP, int S, int DIF, int TW, int NOUT, int NZ>
EMI_DEVFN void hot_bfly_at(int fo, int q, const EmiBuf &bt, int sgn) {
  constexpr int len = LENP * R;
  constexpr unsigned SZ2 = sizeof(real2);
  int blk = 0, j = q;
  if constexpr (len != S) {  // the last pass spans the whole array: block 0
    blk = (int)((unsigned)q / (unsigned)LENP);
    j = q - blk * LENP;
  }
  const int base = blk * len + j;
  real2 w[R];
  if (TW) {
#pragma unroll
    for (int t = 1; t < R; t++) w[t] = emi_buf_ld<real2>(bt, (unsigned)j * SZ2, (unsigned)((t - 1) * LENP) * SZ2);
  }
  const HotLegs<R, LENP> L(fo, base);
  real2 v[R];
#pragma unroll
  for (int t = 0; t < R; t++) {
    if (t >= NZ)
      v[t] = mk2(0.0, 0.0);
    else
      v[t] = hot_ld(L.at(t));
  }
  if (TW) {
    if (sgn > 0) {
#pragma unroll
      for (int t = 1; t < R; t++) w[t].y = -w[t].y;
    }
  }
  if (TW && !DIF) {
#pragma unroll
    for (int t = 1; t < R; t++) v[t] = cmul(v[t], w[t]);
  }
  if constexpr (NZ < R && NZ == R / 2 && (R == 8 || R == 4))
    butterfly_hz<R>(v, sgn);
  else
    butterfly<R>(v, nullptr, S, sgn);  // the specialised plans hold no radix that needs the root table (2, 3, 4, 5, 8, 6, 9, 10)
  if (TW && DIF) {
#pragma unroll
    for (int t = 1; t < R; t++) v[t] = cmul(v[t], w[t]);
  }
#pragma unroll
  for (int t = 0; t < NOUT; t++) hot_st(L.at(t), v[t]);
}
// FLAT as fft_pass_body; fsb: bytes between the fields of the workgroup
// NT: threads of the workgroup (a compile-time property of the plan, hot_threads): with one field per workgroup the
// butterfly loop has a compile-time trip count and disappears
template <int NT, int R, int LENP, int S, int DIF, int TW, int FLAT, int NOUT = R, int NZ = R>
EMI_DEVFN void hot_pass_body(int nfl, int fsb, const real2 *ptw, int sgn) {
  constexpr int nb = S / R;
  const EmiBuf bt = emi_buf_all(ptw);
  if (FLAT) {
    for (int idx = EMI_TID; idx < nfl * nb; idx += NT) {
      const int fl = (int)((unsigned)idx / (unsigned)nb), q = idx - fl * nb;
      hot_bfly_at<R, LENP, S, DIF, TW, NOUT, NZ>(fl * fsb, q, bt, sgn);
    }
  } else {
    for (int fl = 0; fl < nfl; fl++) {
      // several sweeps (2560, 4608, 5120 ...): a real loop -- unrolled, the sweeps' registers overlap and spill
#pragma nounroll
      for (int q0 = 0; q0 < nb; q0 += NT) {
        const int q = q0 + EMI_TID;
        if (q0 + NT <= nb || q < nb) hot_bfly_at<R, LENP, S, DIF, TW, NOUT, NZ>(fl * fsb, q, bt, sgn);
      }
    }
  }
}
// blue_middle_at for the specialised kernels (first factor 8: contiguous runs of 8, no zero padding left at this point); the filter
// spectrum [t][q] through a buffer descriptor (leg offsets t nb as scalar operands)
template <int R, int S>
EMI_DEVFN void hot_middle_at(int fo, int q, const EmiBuf &bb, int conj_b) {
  constexpr int nb = S / R;
  constexpr unsigned SZ2 = sizeof(real2);
  real2 v[R], b[R];
#pragma unroll
  for (int t = 0; t < R; t++) b[t] = emi_buf_ld<real2>(bb, (unsigned)q * SZ2, (unsigned)(t * nb) * SZ2);  // filter values, table [t][q]
  const HotLegs<R, 1> L(fo, q * R);
#pragma unroll
  for (int t = 0; t < R; t++) v[t] = hot_ld(L.at(t));
  butterfly<R>(v, nullptr, S, -1);
#pragma unroll
  for (int t = 0; t < R; t++) v[t] = conj_b ? cmulc(v[t], b[t]) : cmul(v[t], b[t]);
  butterfly<R>(v, nullptr, S, +1);
#pragma unroll
  for (int t = 0; t < R; t++) hot_st(L.at(t), v[t]);
}
template <int NT, int R, int S, int FLAT>
EMI_DEVFN void hot_middle(int nfl, int fsb, const real2 *bh, int conj_b) {
  constexpr int nb = S / R;
  const EmiBuf bb = emi_buf_all(bh);
  if (FLAT) {
    for (int idx = EMI_TID; idx < nfl * nb; idx += NT) {
      const int fl = (int)((unsigned)idx / (unsigned)nb), q = idx - fl * nb;
      hot_middle_at<R, S>(fl * fsb, q, bb, conj_b);
    }
  } else {
    for (int fl = 0; fl < nfl; fl++) {
#pragma nounroll
      for (int q0 = 0; q0 < nb; q0 += NT) {
        const int q = q0 + EMI_TID;
        if (q0 + NT <= nb || q < nb) hot_middle_at<R, S>(fl * fsb, q, bb, conj_b);
      }
    }
  }
}

// z times the chirp in the input stage of the direct kernels: ONE rounding sequence for the buffer path (row inside an NPROMA block) and
// the element-wise path (row cut by blocks) -- which of the two a row takes depends on the decomposition, the gathered fields must not
EMI_DEVFN real2 hot_chirp_mul(real2 z, real2 c) {
  EMI_FP_STRICT();
  return mk2(z.x * c.x - z.y * c.y, z.x * c.y + z.y * c.x);
}

// threads of the workgroup of plan pc in this precision: the host's rule (build_fft_plans: 256 / 512 / 1024 for <= 40 / 80 /
// 160 KiB of LDS), which it applies to exactly the (work length, fields per workgroup) a specialised kernel is matched with
EMI_DEVFN constexpr int hot_threads(int pc) {
  const long long need = (long long)hot_plan(pc).nfl * FFT_LDS_ELEMS(hot_plan(pc).S) * (long long)sizeof(real2);
  return need <= 40960 ? 256 : (need <= 81920 ? 512 : 1024);
}
// forward Bluestein chain on one field: DIF passes nfac-1..1, fused middle, DIT passes 1..last-1
// Workgroup barriers are only needed where data crosses waves.  With the butterfly -> thread map
// q = tid + i * nthreads used by every pass, the radix-8 butterflies q = 64 w .. 64 w + 63 of wave w
// cover exactly the points 512 w .. 512 w + 511 in every pass whose span (lenp * 8) is <= 512 -- the
// passes of the leading factors 8, 8, 8 (lenp 1, 8, 64) and the fused middle pass.  Between two such
// passes a wave only reads what it wrote itself, LDS instructions of one wave execute in order, and the
// workgroup barrier becomes a wave-level fence: 4 instead of 8 barriers per transform, and the waves of
// a workgroup drift apart so that LDS traffic of one overlaps the butterflies of another.
template <int PC>
EMI_DEVFN constexpr bool hot_local(int ip) {  // is the pass of factor ip wave-local (see above)?
  // with several fields per workgroup the (field, butterfly) list is dealt flat: a wave's 64 entries are
  // one field's aligned block only if the butterflies per field are a multiple of 64
  return hot_plan(PC).fac[ip] == 8 && hot_lenp(PC, ip) * 8 <= 512 && (hot_plan(PC).nfl == 1 || (hot_plan(PC).S / 8) % 64 == 0);
}
#define HOT_SYNC(ipa_, ipb_)                                        \
  do {                                                              \
    if constexpr (hot_local<PC>(ipa_) && hot_local<PC>(ipb_))       \
      EMI_WAVE_SYNC();                                              \
    else                                                            \
      EMI_SYNC();                                                   \
  } while (0)

// Bluestein convolution chain on the nfl fields of a workgroup: DIF passes nfac-1..1, fused middle, DIT
// passes 1..nfac-2 (or all of them when LASTDIT), as compile-time recursions over the factor index
// The first pass reads the legs t < NZ only; the input stages of the kernels clear the work array between the end of the chirped row and
// NZ LENP (hot_zero_to), so the pass needs no per-leg `index < sz` test
template <int PC, int IP>
EMI_DEVFN void hot_dif(int nfl, int fs, const FftPlanDev &pl, const real2 *ptw) {
  constexpr HotPlanC H = hot_plan(PC);
  if constexpr (IP >= 1) {
    if constexpr (IP == H.nfac - 1) {
      // first pass: the chirped row occupies sz <= S/2 elements, i.e. only the legs t < R/2 of an even radix
      constexpr int R = H.fac[IP], NZ = (R % 2 == 0) ? R / 2 : R;
      hot_pass_body<hot_threads(PC), R, hot_lenp(PC, IP), H.S, 1, 1, (H.nfl > 1), R, NZ>(nfl, fs * (int)sizeof(real2), ptw + pl.ptw_off[IP], -1);
    }
    else
      hot_pass_body<hot_threads(PC), H.fac[IP], hot_lenp(PC, IP), H.S, 1, 1, (H.nfl > 1)>(nfl, fs * (int)sizeof(real2), ptw + pl.ptw_off[IP], -1);
    HOT_SYNC(IP, IP - 1);
    hot_dif<PC, IP - 1>(nfl, fs, pl, ptw);
  }
}
// elements of the work array the first pass reads: the input stage writes (or clears) all of them
template <int PC>
EMI_DEVFN constexpr int hot_zero_to() {
  constexpr HotPlanC H = hot_plan(PC);
  constexpr int R = H.fac[H.nfac - 1];
  return (R % 2 == 0) ? H.S / 2 : H.S;
}
template <int PC, int IP, int END>
EMI_DEVFN void hot_dit(int nfl, int fs, const FftPlanDev &pl, const real2 *ptw) {
  constexpr HotPlanC H = hot_plan(PC);
  if constexpr (IP < END) {
    // the very last pass of the convolution (direct transform): only the first half of its outputs is read
    constexpr int R = H.fac[IP], NOUT = (IP == H.nfac - 1) ? (R + 1) / 2 : R;
    hot_pass_body<hot_threads(PC), R, hot_lenp(PC, IP), H.S, 0, 1, (H.nfl > 1), NOUT>(nfl, fs * (int)sizeof(real2), ptw + pl.ptw_off[IP], +1);
    // The pass that follows the last one of an inverse kernel's chain is hot_last_to_grid, which walks field after field: with several
    // fields per workgroup its butterflies sit in other waves than those of this pass (dealt flat), so the wave-level fence of HOT_SYNC is
    // not enough -- a race that showed (one row and field in 10^5, TCo399 / KF = 823) once round 5 had shortened the last pass.
    if constexpr (IP + 1 == END && END < H.nfac && H.nfl > 1)
      EMI_SYNC();
    else if constexpr (IP + 1 < H.nfac)
      HOT_SYNC(IP, IP + 1);
    else
      EMI_SYNC();
    hot_dit<PC, IP + 1, END>(nfl, fs, pl, ptw);
  }
}
template <int PC, int LASTDIT>
EMI_DEVFN void hot_conv(int nfl, int fs, const FftPlanDev &pl, const FftTabDev &T, int conj_b) {
  constexpr HotPlanC H = hot_plan(PC);
  const real2 *bh = (const real2 *)T.bhat + pl.bhat_off;
  const real2 *ptw = (const real2 *)T.ptw;
  hot_dif<PC, H.nfac - 1>(nfl, fs, pl, ptw);
  hot_middle<hot_threads(PC), H.fac[0], H.S, (H.nfl > 1)>(nfl, fs * (int)sizeof(real2), bh, conj_b);
  if constexpr (!LASTDIT && H.nfac == 2 && H.nfl > 1)
    EMI_SYNC();  // hot_last_to_grid follows directly (see hot_dit)
  else
    HOT_SYNC(0, 1);
  hot_dit<PC, 1, LASTDIT ? H.nfac : H.nfac - 1>(nfl, fs, pl, ptw);
}
// final DIT pass of the inverse transform of the specialised kernels, stored straight to the grid array (dit_last_to_grid with the
// stride at compile time): z_i = a'_i conj(chirp_i) / S, x_2i = Re z_i, x_(2i+1) = Im z_i for i = j + t LENP < sz <= (S + 1) / 2, i.e. the
// outputs t < (R + 1) / 2 only.  Twiddles, chirp and -- for a row inside one NPROMA block (the usual case) -- the grid row itself go
// through buffer descriptors: the chirp reads as zero and the store is dropped beyond the end of the row, so no output needs a
// compare, a select or a 64-bit address.
template <int NT, int R, int LENP, int S>
EMI_DEVFN void hot_last_to_grid(int nfl, int fsb, const real2 *ptw, const real2 *chirp, int sz, int n, const GridFld *flds, int f0, long long gp0, int nproma) {
  constexpr int NOUT = (R + 1) / 2, nb = S / R;
  constexpr unsigned SZ2 = sizeof(real2);
  const real_t invL = (real_t)(1.0 / (double)S);
  const EmiBuf bt = emi_buf_all(ptw), b_ch = emi_buf(chirp, (unsigned)sz * SZ2);
  for (int fl = 0; fl < nfl; fl++) {
    const GridRow gr = grid_row(flds[f0 + fl], gp0, nproma);
    const bool flat = (gr.rem0 + (unsigned)n <= gr.np) && ((((uintptr_t)(gr.p0 + gr.rem0)) & (2 * sizeof(real_t) - 1)) == 0);
    const EmiBuf b_out = emi_buf(gr.p0 + gr.rem0, flat ? (unsigned)n * (unsigned)sizeof(real_t) : 0u);
#pragma nounroll
    for (int q0 = 0; q0 < nb; q0 += NT) {
      int j = q0 + EMI_TID;  // last pass: one block, j = q
      if (q0 + NT <= nb || j < nb) {
        EMI_OPAQUE(j);
        real2 v[R], w[R], ch[NOUT];
#pragma unroll
        for (int t = 1; t < R; t++) w[t] = emi_buf_ld<real2>(bt, (unsigned)j * SZ2, (unsigned)((t - 1) * LENP) * SZ2);
        const HotLegs<R, LENP> L(fl * fsb, j);
#pragma unroll
        for (int t = 0; t < R; t++) v[t] = hot_ld(L.at(t));
#pragma unroll
        for (int t = 1; t < R; t++) v[t] = cmulc(v[t], w[t]);  // inverse: conjugate twiddles
        if constexpr (R <= 8) {  // chirp values of the outputs: in flight during the butterfly
#pragma unroll
          for (int t = 0; t < NOUT; t++) ch[t] = emi_buf_ld<real2>(b_ch, (unsigned)(j + t * LENP) * SZ2, 0);
        }
        butterfly<R>(v, nullptr, S, +1);
        if constexpr (R > 8) {  // composite radices: no registers to spare during the butterfly
#pragma unroll
          for (int t = 0; t < NOUT; t++) ch[t] = emi_buf_ld<real2>(b_ch, (unsigned)(j + t * LENP) * SZ2, 0);
        }
#pragma unroll
        for (int t = 0; t < NOUT; t++) {
          const real2 z = cscale(cmulc(v[t], ch[t]), invL);
          if (flat) {  // (uniform) i >= sz: dropped by the range check
            emi_buf_st<real2>(b_out, (unsigned)(j + t * LENP) * SZ2, 0, z);
          } else {
            const int i = j + t * LENP;
            if (i < sz) {
              if (grid_pair_ok(gr, 2u * i)) {
                *(real2 *)grid_ptr(gr, 2u * i) = z;
              } else {
                *grid_ptr(gr, 2u * i) = z.x;
                *grid_ptr(gr, 2u * i + 1) = z.y;
              }
            }
          }
        }
      }
    }
  }
}
#undef HOT_SYNC

template <int PC>
EMI_KERNEL_FFT(EMI_FFT_WAVES) void k_fft_inv_hot(EmiGeomDev g, FftTabDev T, FftLaunchDev Lc, const GridFld *flds, int nfld, const real_t *FB,
                                  int ldf, int nproma) {
  constexpr HotPlanC H = hot_plan(PC);
  EMI_LDS_DECL;
  real2 *a = (real2 *)EMI_LDS_PTR;
  const int bid = (int)xcd_swizzle(EMI_BID, Lc.nblocks, 8);
  const int li = bid / Lc.nchunk;
  const FftRowDev rw_ = Lc.rows[li];  // one 64-byte record: everything the input stage needs
  const FftPlanDev &pl = T.plans[rw_.planid];
  const int f0 = (bid - li * Lc.nchunk) * H.nfl;  // H.nfl fields per workgroup
  const int nfl = (H.nfl == 1) ? 1 : ((nfld - f0) < H.nfl ? (nfld - f0) : H.nfl);
  const int n = rw_.n, sz = rw_.sz, nmen = rw_.nmen;
  constexpr int fs = FFT_LDS_ELEMS(H.S);
  const real_t racthe = (real_t)rw_.racthe;
  const real_t adjw = (real_t)(rw_.rw / (double)rw_.n);  // DIR_TRANSAD only (Lc.adj)
  const int fb0 = rw_.fb0;
  const int *frow = g.fftrow ? g.fftrow + fb0 : nullptr;
  const real2 *rtw = (const real2 *)T.rtw + rw_.rtw_off;
  const real2 *chirp = (const real2 *)T.chirp + rw_.chirp_off;
  // stage 1 (FOURIER_IN + FSC): Z_k = (X_k + conj X_{sz-k}) + i w^k (X_k - conj X_{sz-k}), times the chirp
  // Branch-free, the Fourier-row loads of a field first (see k_fft_inv_r16: the plain loop cost four to five serialised memory round
  // trips per pair).
  {
    constexpr int NT = hot_threads(PC), TRIPS = (H.S / 4 + 1 + NT - 1) / NT;
    constexpr unsigned SZ2 = sizeof(real2);
    const unsigned t = (unsigned)EMI_TID, rowb = (unsigned)ldf * (unsigned)sizeof(real_t);
    const int npair = sz / 2 + 1;
    const EmiBuf b_rtw = emi_buf(rtw, (unsigned)(sz + 1) * SZ2), b_ch = emi_buf(chirp, (unsigned)sz * SZ2);
    const real_t fsc = Lc.adj ? adjw : (real_t)1.0;
    for (int fl = 0; fl < nfl; fl++) {
      const GridFld gf = flds[f0 + fl];
      real2 *a = (real2 *)EMI_LDS_PTR + (long long)fl * fs;
      const EmiBuf b_fb = emi_buf(FB + (unsigned long long)(unsigned)fb0 * (unsigned)ldf + 2 * gf.src, (unsigned)nmen * rowb + SZ2);
      real_t fa, fb;
      fin_factors(gf.mode, racthe, fa, fb);
      real2 xa[TRIPS], xb[TRIPS];
#pragma unroll
      for (int i = 0; i < TRIPS; i++) {
        const unsigned k = t + (unsigned)NT * i, k2 = (unsigned)sz - k;
        if (!frow) {  // (uniform) k > NMEN: zero (range check of the descriptor)
          xa[i] = emi_buf_ld<real2>(b_fb, k * rowb, 0);
          xb[i] = emi_buf_ld<real2>(b_fb, k2 * rowb, 0);
        } else {
          const unsigned ka = k < (unsigned)nmen ? k : (unsigned)nmen, kb = k2 < (unsigned)nmen ? k2 : (unsigned)nmen;
          const real2 va = fin_raw(FB, frow[ka], ldf, gf.src), vb = fin_raw(FB, frow[kb], ldf, gf.src);
          xa[i] = k <= (unsigned)nmen ? va : mk2(0, 0);
          xb[i] = k2 <= (unsigned)nmen ? vb : mk2(0, 0);
        }
      }
#pragma unroll
      for (int i = 0; i < TRIPS; i++) {
        const unsigned k = t + (unsigned)NT * i, k2 = (unsigned)sz - k;
        if (k < (unsigned)npair) {
          const real2 wk = cconj(emi_buf_ld<real2>(b_rtw, k * SZ2, 0));
          const real2 ck = emi_buf_ld<real2>(b_ch, k * SZ2, 0), ck2 = emi_buf_ld<real2>(b_ch, k2 * SZ2, 0);  // k2 = sz (k = 0): zero, slot unused
          real2 zk, zk2;
          fin_pair(xa[i], xb[i], k, k2, fa, fb, fsc, wk, ck, ck2, zk, zk2);
          a[FPAD(k)] = zk;
          a[FPAD(k2)] = zk2;  // k = 0: slot sz, a zero (its chirp value reads as zero)
        }
      }
      // the rest of what the first pass reads: zero (the pass itself then needs no `index < sz` test per leg)
      for (int i = sz + 1 + (int)t; i < hot_zero_to<PC>(); i += NT) a[FPAD(i)] = mk2(0.0, 0.0);
    }
  }
  EMI_SYNC();
  hot_conv<PC, 0>(nfl, fs, pl, T, 1);
  constexpr int last = H.nfac - 1;
  hot_last_to_grid<hot_threads(PC), H.fac[last], hot_lenp(PC, last), H.S>(nfl, fs * (int)sizeof(real2), (const real2 *)T.ptw + pl.ptw_off[last], chirp, sz, n, flds, f0,
                                                                          rw_.gpoff, nproma);
}

template <int PC>
EMI_KERNEL_FFT(EMI_FFT_WAVES) void k_fft_dir_hot(EmiGeomDev g, FftTabDev T, FftLaunchDev Lc, const GridFld *flds, int nfld, real_t *FB, int ldf,
                                  int nproma) {
  constexpr HotPlanC H = hot_plan(PC);
  EMI_LDS_DECL;
  real2 *a = (real2 *)EMI_LDS_PTR;
  const int bid = (int)xcd_swizzle(EMI_BID, Lc.nblocks, 8);
  const int li = bid / Lc.nchunk;
  const FftRowDev rw_ = Lc.rows[li];  // one 64-byte record: everything the input stage needs
  const FftPlanDev &pl = T.plans[rw_.planid];
  const int f0 = (bid - li * Lc.nchunk) * H.nfl;
  const int nfl = (H.nfl == 1) ? 1 : ((nfld - f0) < H.nfl ? (nfld - f0) : H.nfl);
  const int n = rw_.n, sz = rw_.sz, nmen = rw_.nmen;
  constexpr int fs = FFT_LDS_ELEMS(H.S);
  constexpr int NT = hot_threads(PC);
  constexpr unsigned SZ2 = sizeof(real2);
  const unsigned t = (unsigned)EMI_TID;
  const int fb0 = rw_.fb0;
  const int *frow = g.fftrow ? g.fftrow + fb0 : nullptr;
  const real2 *rtw = (const real2 *)T.rtw + rw_.rtw_off;
  const real2 *chirp = (const real2 *)T.chirp + rw_.chirp_off;
  const EmiBuf b_ch = emi_buf(chirp, (unsigned)sz * SZ2), b_rtw = emi_buf(rtw, (unsigned)(sz + 1) * SZ2);
  // stage 1 (TRGTOL local copy): z_l = x_{2l} + i x_{2l+1}, times the chirp.  A row inside one NPROMA block (the usual case) is one
  // buffer: its tail and the chirp beyond sz read as zero, so the loop also clears what the first pass reads past the row
  // (hot_zero_to) and neither needs a test per element; all loads of a thread first.
  for (int fl = 0; fl < nfl; fl++) {
    const GridFld gf = flds[f0 + fl];
    real2 *a = (real2 *)EMI_LDS_PTR + (long long)fl * fs;
    const GridRow gr = grid_row(gf, rw_.gpoff, nproma);
    const bool flat = (gr.rem0 + (unsigned)n <= gr.np) && ((((uintptr_t)(gr.p0 + gr.rem0)) & (2 * sizeof(real_t) - 1)) == 0);
    if (flat) {
      constexpr int TRIPS = (hot_zero_to<PC>() + NT - 1) / NT;
      const EmiBuf b_in = emi_buf(gr.p0 + gr.rem0, (unsigned)n * (unsigned)sizeof(real_t));
      real2 z[TRIPS], c[TRIPS];
#pragma unroll
      for (int i = 0; i < TRIPS; i++) {
        const unsigned off = (t + (unsigned)(NT * i)) * SZ2;
        z[i] = emi_buf_ld<real2>(b_in, off, 0);
        c[i] = emi_buf_ld<real2>(b_ch, off, 0);
      }
#pragma unroll
      for (int i = 0; i < TRIPS; i++) {
        const int lz = (int)t + NT * i;
        if ((i + 1) * NT <= hot_zero_to<PC>() || lz < hot_zero_to<PC>()) a[FPAD(lz)] = hot_chirp_mul(z[i], c[i]);
      }
    } else {
      for (int lz = EMI_TID; lz < hot_zero_to<PC>(); lz += NT) {
        real2 z = mk2(0.0, 0.0);
        if (lz < sz) {
          if (grid_pair_ok(gr, 2u * lz)) {
            z = *(const real2 *)grid_ptr(gr, 2u * lz);
          } else {
            z.x = *grid_ptr(gr, 2u * lz);
            z.y = *grid_ptr(gr, 2u * lz + 1);
          }
          z = hot_chirp_mul(z, chirp[lz]);
        }
        a[FPAD(lz)] = z;
      }
    }
  }
  EMI_SYNC();
  hot_conv<PC, 1>(nfl, fs, pl, T, 0);
  // stage 3 (FOURIER_OUT): X_k = 1/2 [ (Z_k + conj Z_{sz-k}) - i exp(-2 pi i k/n) (Z_k - conj Z_{sz-k}) ], k <= NMEN.  Tables through
  // buffer descriptors; with one task the rows of the latitude are consecutive and the output is one buffer too (lane offset k x row
  // bytes, one 32-bit multiply per thread)
  const real_t invL = (real_t)(1.0 / (double)H.S);
  const unsigned rowb = (unsigned)ldf * (unsigned)sizeof(real_t);
  for (int fl = 0; fl < nfl; fl++) {
    const GridFld gf = flds[f0 + fl];
    const real2 *a = (const real2 *)EMI_LDS_PTR + (long long)fl * fs;
    const real_t sc = (Lc.adj ? (real_t)1.0 : (real_t)(rw_.rw / (double)n)) * fft_dir_mode_scale(gf.mode, (real_t)rw_.racthe);
    const EmiBuf b_fb = emi_buf(FB + (unsigned long long)(unsigned)fb0 * (unsigned)ldf + 2 * (f0 + fl), frow ? 0u : (unsigned)nmen * rowb + SZ2);
    unsigned ko = t * rowb;
    for (int k = EMI_TID; k <= nmen; k += NT, ko += (unsigned)NT * rowb) {
      const int kb = (k == 0) ? 0 : sz - k;
      const real2 ca = emi_buf_ld<real2>(b_ch, (unsigned)k * SZ2, 0), cb = emi_buf_ld<real2>(b_ch, (unsigned)kb * SZ2, 0);
      const real2 wk = emi_buf_ld<real2>(b_rtw, (unsigned)k * SZ2, 0);
      real2 za = a[FPAD(k)], zb = a[FPAD(kb)];
      za = cscale(cmul(za, ca), invL);
      zb = cscale(cmul(zb, cb), invL);
      real2 s1 = cadd(za, cconj(zb)), d1 = csub(za, cconj(zb));
      real2 tt = cmuli(cmul(wk, d1));
      const real2 x = cscale(mk2((real_t)0.5 * (s1.x - tt.x), (real_t)0.5 * (s1.y - tt.y)), sc);
      if (!frow)
        emi_buf_st<real2>(b_fb, ko, 0, x);
      else
        *(real2 *)(FB + (unsigned long long)(unsigned)frow[k] * (unsigned)ldf + 2 * (f0 + fl)) = x;
    }
  }
}

// ==========================================================================================
// Register-resident Bluestein kernels k_fft_dir_r16<R1> / k_fft_inv_r16<R1> (round 3): the rows whose Bluestein work
// length is S = 256 R1, R1 in {8, 10, 12, 16, 18, 20} (TCo1279: every row longer than 1538 points), one field per
// workgroup.  Same transform as k_fft_*_hot (FTDIR / FTINV of ftdir_mod.F90:67-84, ftinv_mod.F90:65-84 through a chirp-z
// convolution), different machine mapping:
//   * S = R1 * 16 * 16.  A thread keeps its points in registers through the whole chain
//         A1 (radix R1, stride 256, 256 threads)  X  A2 (radix 16, stride 16)  L  A3 (radix 16) * filter * B3  L  B2  X  B1
//     (A = forward DIF passes, B = inverse DIT passes; A2 .. B2 run on 16 R1 threads holding 16 points each).  The zero half of
//     the padded row never enters A1 and the unread half of the result never leaves B1 (r16_first / r16_last).
//   * LDS is only the exchange medium: 4 round trips per row instead of the 7 of the in-place LDS kernels, and in fp64 one REAL
//     plane at a time (real parts, then imaginary parts), so a row needs R1 * 2176 bytes instead of R1 * 4096 and four
//     independent 4-wave workgroups share a CU instead of two 8-wave ones.  X exchanges cross waves (workgroup barriers);
//     L exchanges stay inside one 16-lane row of a wave (plane row `hi`), so they need no barrier at all.
//     Plane layout: block k0 (256 points) at k0 * 272; inside an L exchange point (k1, c) of the block at 17 k1 + c: every
//     ds_read_b64 / ds_write_b64 of the four exchanges is bank-conflict free (SQ_LDS_BANK_CONFLICT = 0 measured).
//   * twiddles: w_256^(c k1) of A2 / B2 from a 240-entry LDS copy; w_S^(t k0) of A1 / B1 as w^(t qa) w^(4 t qb), k0 = qa + 4 qb,
//     from at most 7 coalesced loads (15 table loads per pass cost more than the 9 extra complex products: tools/fft_r16_probe);
//     tables and grid rows through buffer descriptors (one lane offset, no 64-bit vector address arithmetic, free masking of
//     the row tail).
// Measured in tools/fft_r16_probe.hip (S = 4096, synthetic rows): 18.7 ns per row against 28.3 ns for k_fft_dir_hot<4>.
// ==========================================================================================
#define R16_ROWP 272
EMI_DEVFN constexpr int r16_threads(int R1) { return 16 * R1 > 256 ? ((16 * R1 + 63) / 64) * 64 : 256; }
EMI_DEVFN constexpr int r16_lds_bytes(int R1) { return R1 * R16_ROWP * 8 + 240 * 2 * (int)sizeof(real_t); }

// v *= exp(sgn 2 pi i J / N), J and N compile-time constants
template <int N, int J>
EMI_DEVFN real2 r16_cmul_w(real2 a, int sgn) {
  constexpr real_t c8[4] = {1.0, 0.707106781186547524401, 0.0, -0.707106781186547524401};
  constexpr real_t s8[4] = {0.0, 0.707106781186547524401, 1.0, 0.707106781186547524401};
  constexpr real_t c10[5] = {1.0, 0.809016994374947424102, 0.309016994374947424102, -0.309016994374947424102, -0.809016994374947424102};
  constexpr real_t s10[5] = {0.0, 0.587785252292473129169, 0.951056516295153572116, 0.951056516295153572116, 0.587785252292473129169};
  constexpr real_t c12[6] = {1.0, 0.866025403784438646764, 0.5, 0.0, -0.5, -0.866025403784438646764};
  constexpr real_t s12[6] = {0.0, 0.5, 0.866025403784438646764, 1.0, 0.866025403784438646764, 0.5};
  constexpr real_t c16[8] = {1.0, 0.923879532511286756128, 0.707106781186547524401, 0.382683432365089771728, 0.0, -0.382683432365089771728,
                             -0.707106781186547524401, -0.923879532511286756128};
  constexpr real_t s16[8] = {0.0, 0.382683432365089771728, 0.707106781186547524401, 0.923879532511286756128, 1.0, 0.923879532511286756128,
                             0.707106781186547524401, 0.382683432365089771728};
  constexpr real_t c18[9] = {1.0, 0.939692620785908384054, 0.766044443118978035202, 0.5, 0.173648177666930348852, -0.173648177666930348852,
                             -0.5, -0.766044443118978035202, -0.939692620785908384054};
  constexpr real_t s18[9] = {0.0, 0.342020143325668733044, 0.642787609686539326323, 0.866025403784438646764, 0.984807753012208059367,
                             0.984807753012208059367, 0.866025403784438646764, 0.642787609686539326323, 0.342020143325668733044};
  constexpr real_t c20[10] = {1.0, 0.951056516295153572116, 0.809016994374947424102, 0.587785252292473129169, 0.309016994374947424102, 0.0,
                              -0.309016994374947424102, -0.587785252292473129169, -0.809016994374947424102, -0.951056516295153572116};
  constexpr real_t s20[10] = {0.0, 0.309016994374947424102, 0.587785252292473129169, 0.809016994374947424102, 0.951056516295153572116, 1.0,
                              0.951056516295153572116, 0.809016994374947424102, 0.587785252292473129169, 0.309016994374947424102};
  static_assert(J >= 0 && 2 * J < N, "r16_cmul_w: first half of the circle only");
  if constexpr (J == 0) {
    return a;
  } else if constexpr (4 * J == N) {
    return cmuli(mk2(a.x * (real_t)sgn, a.y * (real_t)sgn));  // sgn i a; sgn = +-1 is a compile-time constant at every call
  } else {
    constexpr real_t c = (N == 8) ? c8[J % 4] : (N == 10) ? c10[J % 5] : (N == 12) ? c12[J % 6] : (N == 16) ? c16[J % 8] : (N == 18) ? c18[J % 9] : c20[J % 10];
    constexpr real_t s = (N == 8) ? s8[J % 4] : (N == 10) ? s10[J % 5] : (N == 12) ? s12[J % 6] : (N == 16) ? s16[J % 8] : (N == 18) ? s18[J % 9] : s20[J % 10];
    const real_t sn = s * (real_t)sgn;
    return mk2(a.x * c - a.y * sn, a.x * sn + a.y * c);
  }
}

// dst[J] = (ADD ? add[J] : 0) + src[J] W_N^(sgn J) for J = J0 .. H-1 (compile-time recursion: the twiddles are template constants)
template <int N, int J, int H, int ADD>
EMI_DEVFN void r16_tw_each(real2 *dst, const real2 *src, const real2 *add, int sgn) {
  if constexpr (J < H) {
    const real2 m = r16_cmul_w<N, J>(src[J], sgn);
    dst[J] = ADD ? cadd(add[J], m) : m;
    r16_tw_each<N, J + 1, H, ADD>(dst, src, add, sgn);
  }
}
// radix 16, natural order in and out: Y[k] = sum_a x[a] W16^(a k), a = 4 a1 + a0, k = k1 + 4 k0
EMI_DEVFN void r16_bf16(real2 *v, int sgn) {
#pragma unroll
  for (int a0 = 0; a0 < 4; a0++) bf4(v[a0], v[a0 + 4], v[a0 + 8], v[a0 + 12], sgn);  // -> u[a0][k1] at a0 + 4 k1
  v[5] = r16_cmul_w<16, 1>(v[5], sgn);
  v[6] = r16_cmul_w<16, 2>(v[6], sgn);
  v[7] = r16_cmul_w<16, 3>(v[7], sgn);
  v[9] = r16_cmul_w<16, 2>(v[9], sgn);
  v[10] = r16_cmul_w<16, 4>(v[10], sgn);
  v[11] = r16_cmul_w<16, 6>(v[11], sgn);
  v[13] = r16_cmul_w<16, 3>(v[13], sgn);
  v[14] = r16_cmul_w<16, 6>(v[14], sgn);
  {  // W16^9 = -W16^1
    const real2 m = r16_cmul_w<16, 1>(v[15], sgn);
    v[15] = mk2(-m.x, -m.y);
  }
#pragma unroll
  for (int k1 = 0; k1 < 4; k1++) bf4(v[4 * k1], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3], sgn);  // -> Y[k1 + 4 k0] at 4 k1 + k0
  real2 y[16];
#pragma unroll
  for (int j = 0; j < 16; j++) y[(j >> 2) + 4 * (j & 3)] = v[j];
#pragma unroll
  for (int j = 0; j < 16; j++) v[j] = y[j];
}
// first pass of the zero-padded convolution: radix R1 of (v[0 .. R1/2-1], 0, ..., 0) -> v[0 .. R1-1].
// a = (R1/2) a1 + a0 with a1 = 1 zero: Y[k1 + 2 k'] = DFT_{R1/2}( x[a0] W_R1^(a0 k1) )[k']
template <int R1>
EMI_DEVFN void r16_first(real2 *v, int sgn) {
  constexpr int H = R1 / 2;
  real2 e[H], o[H];
#pragma unroll
  for (int a0 = 0; a0 < H; a0++) e[a0] = v[a0];
  r16_tw_each<R1, 0, H, 0>(o, v, nullptr, sgn);
  butterfly<H>(e, nullptr, 0, sgn);
  butterfly<H>(o, nullptr, 0, sgn);
#pragma unroll
  for (int k = 0; k < H; k++) v[2 * k] = e[k], v[2 * k + 1] = o[k];
}
// last pass, of which only the outputs k < R1/2 are read: a = 2 a' + a1, Y[k] = E[k] + W_R1^k O[k]
template <int R1>
EMI_DEVFN void r16_last(real2 *v, int sgn) {
  constexpr int H = R1 / 2;
  real2 e[H], o[H];
#pragma unroll
  for (int a = 0; a < H; a++) e[a] = v[2 * a], o[a] = v[2 * a + 1];
  butterfly<H>(e, nullptr, 0, sgn);
  butterfly<H>(o, nullptr, 0, sgn);
  r16_tw_each<R1, 0, H, 1>(v, o, e, sgn);
}

// The convolution chain of one row.  In: v[a] = u[t + 256 a], a < R1/2 (threads t < 256), the chirped row.  Out: v[a] = (u * b)[t + 256 a],
// unscaled (S times the circular convolution with the filter whose spectrum is behind b_bh; CONJB: with its conjugate).
// `lds`: the plane (R1 * 272 * 8 bytes; no other use between entry and exit), tw2s: the LDS copy of w_256^(c k1).
template <int R1, int CONJB>
EMI_DEVFN void r16_conv(real2 *vio, const unsigned t, const EmiBuf &b_tw, const EmiBuf &b_bh, char *lds, const real2 *tw2s) {
  constexpr int H = R1 / 2, NMID = 16 * R1, NT = r16_threads(R1);
  constexpr bool PL = sizeof(real_t) == 8;  // fp64: one real plane at a time; fp32: a complex number is 8 bytes, one phase
  constexpr unsigned SZ2 = sizeof(real2);
  const unsigned hi = t >> 4, lo = t & 15;
  const bool edge = (NT == 256) ? true : (t < 256u), mid = (NMID == NT) ? true : (t < (unsigned)NMID);
  real_t *pr = (real_t *)lds;
  real2 *pc = (real2 *)lds;
  // Between the phases the points live in separate real / imaginary scalars: a real2 that crosses a branch is a 4-register
  // tuple even when only one half is alive (the other half already written to the plane), which at R1 = 18, 20 spilled
  real_t vx[R1], vy[R1], wx[16], wy[16];
#define R16_PACK(dst_, x_, y_, n_) \
  _Pragma("unroll") for (int i_ = 0; i_ < (n_); i_++) dst_[i_] = mk2(x_[i_], y_[i_])
#define R16_UNPACK(x_, y_, src_, n_) \
  _Pragma("unroll") for (int i_ = 0; i_ < (n_); i_++) x_[i_] = src_[i_].x, y_[i_] = src_[i_].y
  // ---- A1: radix R1 over a, then w_S^(t k0) = w^(t qa) w^(4 t qb), k0 = qa + 4 qb: two loads (w^t, w^4t), the other digits by
  // products (one or two more rounding errors of 1e-16 each) -- fewer registers held across the butterfly than a table row per digit
  if (edge) {
    real2 v[R1];
#pragma unroll
    for (int a = 0; a < H; a++) v[a] = vio[a];
    r16_first<R1>(v, -1);
    EMI_SCHED_FENCE();  // the twiddle loads stay behind the butterfly: with R1 = 18, 20 its 4 R1 data registers leave no room for them
    const real2 w1 = emi_buf_ld<real2>(b_tw, t * SZ2, 0), w4 = emi_buf_ld<real2>(b_tw, t * SZ2, 3u * 256u * SZ2);
    const real2 w2 = cmul(w1, w1), w3 = cmul(w2, w1);
    real2 wq = w4;  // w^(4 t qb)
#pragma unroll
    for (int k0 = 1; k0 < R1; k0++) {
      const int qa = k0 & 3, qb = k0 >> 2;
      if (qa == 0 && qb > 1) wq = cmul(wq, w4);
      const real2 wa = (qa == 1) ? w1 : (qa == 2) ? w2 : w3;
      v[k0] = cmul(v[k0], qb == 0 ? wa : (qa == 0 ? wq : cmul(wa, wq)));
    }
    R16_UNPACK(vx, vy, v, R1);
  }
  // ---- X1: thread t, block k0 -> thread (hi = k0, lo = c), slot b = t >> 4
  if constexpr (PL) {
    if (edge) {
#pragma unroll
      for (int k0 = 0; k0 < R1; k0++) pr[k0 * R16_ROWP + t] = vx[k0];
    }
    EMI_LDS_SYNC();
    if (mid) {
#pragma unroll
      for (int b = 0; b < 16; b++) wx[b] = pr[hi * R16_ROWP + 16 * b + lo];
    }
    EMI_LDS_SYNC();
    if (edge) {
#pragma unroll
      for (int k0 = 0; k0 < R1; k0++) pr[k0 * R16_ROWP + t] = vy[k0];
    }
    EMI_LDS_SYNC();
    if (mid) {
#pragma unroll
      for (int b = 0; b < 16; b++) wy[b] = pr[hi * R16_ROWP + 16 * b + lo];
    }
  } else {
    if (edge) {
#pragma unroll
      for (int k0 = 0; k0 < R1; k0++) pc[k0 * R16_ROWP + t] = mk2(vx[k0], vy[k0]);
    }
    EMI_LDS_SYNC();
    if (mid) {
#pragma unroll
      for (int b = 0; b < 16; b++) {
        const real2 q = pc[hi * R16_ROWP + 16 * b + lo];
        wx[b] = q.x, wy[b] = q.y;
      }
    }
  }
  // ---- A2 in thread (k0 = hi, c = lo): radix 16 over b, then w_256^(c k1)
  if (mid) {
    real2 w[16];
    R16_PACK(w, wx, wy, 16);
    r16_bf16(w, -1);
#pragma unroll
    for (int k1 = 1; k1 < 16; k1++) w[k1] = cmul(w[k1], tw2s[(k1 - 1) * 16 + lo]);
    R16_UNPACK(wx, wy, w, 16);
  }
  // ---- L1: thread (k0, c), value k1 -> thread (k0, k1), slot c.  Row hi of the plane belongs to this 16-lane row alone.
  if constexpr (PL) {
    EMI_WAVE_FENCE();
    if (mid) {
#pragma unroll
      for (int k1 = 0; k1 < 16; k1++) pr[hi * R16_ROWP + 17 * k1 + lo] = wx[k1];
    }
    EMI_WAVE_FENCE();
    if (mid) {
#pragma unroll
      for (int c = 0; c < 16; c++) wx[c] = pr[hi * R16_ROWP + 17 * lo + c];
    }
    EMI_WAVE_FENCE();
    if (mid) {
#pragma unroll
      for (int k1 = 0; k1 < 16; k1++) pr[hi * R16_ROWP + 17 * k1 + lo] = wy[k1];
    }
    EMI_WAVE_FENCE();
    if (mid) {
#pragma unroll
      for (int c = 0; c < 16; c++) wy[c] = pr[hi * R16_ROWP + 17 * lo + c];
    }
  } else {
    EMI_WAVE_FENCE();
    if (mid) {
#pragma unroll
      for (int k1 = 0; k1 < 16; k1++) pc[hi * R16_ROWP + 17 * k1 + lo] = mk2(wx[k1], wy[k1]);
    }
    EMI_WAVE_FENCE();
    if (mid) {
#pragma unroll
      for (int c = 0; c < 16; c++) {
        const real2 q = pc[hi * R16_ROWP + 17 * lo + c];
        wx[c] = q.x, wy[c] = q.y;
      }
    }
  }
  // ---- A3 (over c), filter, B3 (over k2) in thread (k0, k1) = t, then conj w_256^(c k1) for B2
  if (mid) {
    real2 w[16];
    R16_PACK(w, wx, wy, 16);
    r16_bf16(w, -1);  // w[k2]: spectrum at k0 + R1 (k1 + 16 k2)
#pragma unroll
    for (int k2 = 0; k2 < 16; k2++) {
      const real2 b = emi_buf_ld<real2>(b_bh, t * SZ2, (unsigned)k2 * (unsigned)NMID * SZ2);  // table [k2][t]
      w[k2] = CONJB ? cmulc(w[k2], b) : cmul(w[k2], b);
    }
    r16_bf16(w, +1);  // w[c]
#pragma unroll
    for (int c = 1; c < 16; c++) w[c] = cmulc(w[c], tw2s[(c - 1) * 16 + lo]);
    R16_UNPACK(wx, wy, w, 16);
  }
  // ---- L2: thread (k0, k1), value c -> thread (k0, c), slot k1
  if constexpr (PL) {
    EMI_WAVE_FENCE();
    if (mid) {
#pragma unroll
      for (int c = 0; c < 16; c++) pr[hi * R16_ROWP + 17 * lo + c] = wx[c];
    }
    EMI_WAVE_FENCE();
    if (mid) {
#pragma unroll
      for (int k1 = 0; k1 < 16; k1++) wx[k1] = pr[hi * R16_ROWP + 17 * k1 + lo];
    }
    EMI_WAVE_FENCE();
    if (mid) {
#pragma unroll
      for (int c = 0; c < 16; c++) pr[hi * R16_ROWP + 17 * lo + c] = wy[c];
    }
    EMI_WAVE_FENCE();
    if (mid) {
#pragma unroll
      for (int k1 = 0; k1 < 16; k1++) wy[k1] = pr[hi * R16_ROWP + 17 * k1 + lo];
    }
  } else {
    EMI_WAVE_FENCE();
    if (mid) {
#pragma unroll
      for (int c = 0; c < 16; c++) pc[hi * R16_ROWP + 17 * lo + c] = mk2(wx[c], wy[c]);
    }
    EMI_WAVE_FENCE();
    if (mid) {
#pragma unroll
      for (int k1 = 0; k1 < 16; k1++) {
        const real2 q = pc[hi * R16_ROWP + 17 * k1 + lo];
        wx[k1] = q.x, wy[k1] = q.y;
      }
    }
  }
  // ---- B2 in thread (k0, c): radix 16 over k1 -> b
  if (mid) {
    real2 w[16];
    R16_PACK(w, wx, wy, 16);
    r16_bf16(w, +1);
    R16_UNPACK(wx, wy, w, 16);
  }
  // ---- X2: thread (k0, c), value b -> thread t = 16 b + c, slot k0
  if constexpr (PL) {
    EMI_WAVE_FENCE();
    if (mid) {
#pragma unroll
      for (int b = 0; b < 16; b++) pr[hi * R16_ROWP + 16 * b + lo] = wx[b];
    }
    EMI_LDS_SYNC();
    if (edge) {
#pragma unroll
      for (int k0 = 0; k0 < R1; k0++) vx[k0] = pr[k0 * R16_ROWP + t];
    }
    EMI_LDS_SYNC();
    if (mid) {
#pragma unroll
      for (int b = 0; b < 16; b++) pr[hi * R16_ROWP + 16 * b + lo] = wy[b];
    }
    EMI_LDS_SYNC();
    if (edge) {
#pragma unroll
      for (int k0 = 0; k0 < R1; k0++) vy[k0] = pr[k0 * R16_ROWP + t];
    }
  } else {
    EMI_WAVE_FENCE();
    if (mid) {
#pragma unroll
      for (int b = 0; b < 16; b++) pc[hi * R16_ROWP + 16 * b + lo] = mk2(wx[b], wy[b]);
    }
    EMI_LDS_SYNC();
    if (edge) {
#pragma unroll
      for (int k0 = 0; k0 < R1; k0++) {
        const real2 q = pc[k0 * R16_ROWP + t];
        vx[k0] = q.x, vy[k0] = q.y;
      }
    }
  }
  // ---- B1 in thread t: conj w_S^(t k0), radix R1 over k0 -> a < R1/2
  if (edge) {
    real2 v[R1];
    R16_PACK(v, vx, vy, R1);
    const real2 w1 = emi_buf_ld<real2>(b_tw, t * SZ2, 0), w4 = emi_buf_ld<real2>(b_tw, t * SZ2, 3u * 256u * SZ2);
    const real2 w2 = cmul(w1, w1), w3 = cmul(w2, w1);
    real2 wq = w4;
#pragma unroll
    for (int k0 = 1; k0 < R1; k0++) {
      const int qa = k0 & 3, qb = k0 >> 2;
      if (qa == 0 && qb > 1) wq = cmul(wq, w4);
      const real2 wa = (qa == 1) ? w1 : (qa == 2) ? w2 : w3;
      v[k0] = cmulc(v[k0], qb == 0 ? wa : (qa == 0 ? wq : cmul(wa, wq)));
    }
    EMI_SCHED_FENCE();
    r16_last<R1>(v, +1);
#pragma unroll
    for (int a = 0; a < H; a++) vio[a] = v[a];
  }
#undef R16_PACK
#undef R16_UNPACK
}

// -DEMI_MR_STAMP (experiments only, as in emi_mr_body.h): thread 0 of every R1 = 16 workgroup adds the clock ticks between consecutive
// R16_STAMP points to emi_mr_stamp[] (tools/r16_stamp.py reads them: `python tools/r16_stamp.py inv|dir` with the instrumented build in $EMI_LIB)
#if defined(EMI_MR_STAMP) && !defined(EMI_LEG_STAMP) && !defined(EMI_CPU_EMU)
#define R16_STAMP_BEGIN() unsigned long long r16_acc[6] = {0}; unsigned long long r16_prev = __builtin_readcyclecounter()
#define R16_STAMP(i_) do { const unsigned long long n_ = __builtin_readcyclecounter(); r16_acc[i_] += n_ - r16_prev; r16_prev = n_; } while (0)
#define R16_STAMP_END(n_) do { if (R1 == 16 && EMI_TID == 0) { for (int i_ = 0; i_ < (n_); i_++) atomicAdd(&emi_mr_stamp[i_], r16_acc[i_]); atomicAdd(&emi_mr_stamp[7], 1ull); } } while (0)
#else
#define R16_STAMP_BEGIN() ((void)0)
#define R16_STAMP(i_) ((void)0)
#define R16_STAMP_END(n_) ((void)0)
#endif

// Waves per SIMD the register-resident kernels are compiled for.  fp32: four (106 - 116 registers).  fp64 (round 6, same-box A/B, TCo1279, ms per
// direction at four -> three waves): the direct kernels R1 = 8 | 10 | 12 | 16: 3.99 -> 3.71 | 5.11 -> 4.89 | 6.13 -> 5.62 | 15.35 -> 14.89 -- with
// 144 - 162 instead of 99 - 128 registers the chain keeps more of its table loads in flight and R1 = 12 sheds its scratch; the inverse kernels gain
// only at R1 = 12 (6.87 -> 6.41; R1 = 8 | 10 | 16: 4.22 -> 4.21 | 5.29 -> 5.40 | 15.89 -> 16.00).  LDS (26 - 39 KB) would allow four or more workgroups
// per CU either way; three run.
#ifndef EMI_R16_WAVES_F32
#define EMI_R16_WAVES_F32 4
#endif
#define EMI_R16_WAVES_DIR (sizeof(real_t) == 8 ? 3 : EMI_R16_WAVES_F32)
#define EMI_R16_WAVES_INV(R1_) (sizeof(real_t) == 8 ? ((R1_) == 12 ? 3 : 4) : EMI_R16_WAVES_F32)
template <int R1>
EMI_KERNEL_LB2(r16_threads(R1), EMI_R16_WAVES_DIR) void k_fft_dir_r16(EmiGeomDev g, FftTabDev T, FftLaunchDev Lc, const GridFld *flds, int nfld, real_t *FB, int ldf,
                                                   int nproma) {
  constexpr int H = R1 / 2, NT = r16_threads(R1), S = 256 * R1;
  constexpr unsigned SZ2 = sizeof(real2);
  EMI_LDS_DECL;
  char *lds = EMI_LDS_PTR;
  real2 *tw2s = (real2 *)(lds + R1 * R16_ROWP * 8), *zbuf = (real2 *)lds;
  const unsigned t = (unsigned)EMI_TID;
  const bool edge = (NT == 256) ? true : (t < 256u);
  const int bid = (int)xcd_swizzle(EMI_BID, Lc.nblocks, 8);
  const int li = bid / Lc.nchunk;
  const FftRowDev rw_ = Lc.rows[li];  // one 64-byte record instead of the chain lats -> planid -> plans -> offsets, nmen / fbase / gpoff [lat]
  const int f0 = bid - li * Lc.nchunk;  // one field per workgroup
  const int n = rw_.n, sz = rw_.sz, nmen = rw_.nmen;
  const int fb0 = rw_.fb0;
  const int *frow = g.fftrow ? g.fftrow + fb0 : nullptr;
  const real2 *rtw = (const real2 *)T.rtw + rw_.rtw_off;
  const EmiBuf b_ch = emi_buf((const real2 *)T.chirp + rw_.chirp_off, (unsigned)sz * SZ2);
  const EmiBuf b_tw = emi_buf((const real2 *)T.ptw + rw_.ptw_off0, 7u * 256u * SZ2);
  const EmiBuf b_bh = emi_buf((const real2 *)T.bhat + rw_.bhat_off, (unsigned)S * SZ2);
  R16_STAMP_BEGIN();
  if (t < 240u) tw2s[t] = ((const real2 *)T.tw256)[t];
  const GridFld gf = flds[f0];
  // stage 1 (TRGTOL local copy): z_l = x_{2l} + i x_{2l+1}, times the chirp; l = t + 256 a
  real2 v[H];
  if (edge) {
    const GridRow gr = grid_row(gf, rw_.gpoff, nproma);
    const bool flat = (gr.rem0 + (unsigned)n <= gr.np) && ((((uintptr_t)(gr.p0 + gr.rem0)) & (2 * sizeof(real_t) - 1)) == 0);
    if (flat) {  // whole row inside one NPROMA block and 2-element aligned (uniform): the row is one buffer, its tail reads as zero
      const EmiBuf b_in = emi_buf(gr.p0 + gr.rem0, (unsigned)n * (unsigned)sizeof(real_t));
#pragma unroll
      for (int a = 0; a < H; a++) {
        const unsigned off = (t + 256u * a) * SZ2;
        v[a] = cmul(emi_buf_ld<real2>(b_in, off, 0), emi_buf_ld<real2>(b_ch, off, 0));
      }
    } else {
#pragma unroll
      for (int a = 0; a < H; a++) {
        const unsigned lz = t + 256u * a;
        real2 z = mk2(0, 0);
        if (lz < (unsigned)sz) {
          if (grid_pair_ok(gr, 2u * lz)) {
            z = *(const real2 *)grid_ptr(gr, 2u * lz);
          } else {
            z.x = *grid_ptr(gr, 2u * lz);
            z.y = *grid_ptr(gr, 2u * lz + 1);
          }
        }
        v[a] = cmul(z, emi_buf_ld<real2>(b_ch, lz * SZ2, 0));
      }
    }
  }
  R16_STAMP(0);
  r16_conv<R1, 0>(v, t, b_tw, b_bh, lds, tw2s);
  R16_STAMP(1);
  // Z_i = conv_i chirp_i / S, i < sz, to LDS (complex, natural order: sz <= S/2 of them fit the plane)
  EMI_LDS_SYNC();  // every thread has read its last plane values
  if (edge) {
#pragma unroll
    for (int a = 0; a < H; a++) {
      const unsigned i = t + 256u * a;
      zbuf[i] = cmul(v[a], emi_buf_ld<real2>(b_ch, i * SZ2, 0));  // i >= sz: zero chirp, never read (1 / S: in the filter table)
    }
  }
  EMI_LDS_SYNC();
  R16_STAMP(2);
  // stage 3 (FOURIER_OUT): X_k = 1/2 [ (Z_k + conj Z_{sz-k}) - i exp(-2 pi i k/n) (Z_k - conj Z_{sz-k}) ], k <= NMEN
  const real_t sc = (real_t)0.5 * (Lc.adj ? (real_t)1.0 : (real_t)(rw_.rw / (double)n)) * fft_dir_mode_scale(gf.mode, (real_t)rw_.racthe);
  for (int k = (int)t; k <= nmen; k += NT) {
    const int kb = (k == 0) ? 0 : sz - k;
    const real2 za = zbuf[k], zb = zbuf[kb];
    const real2 s1 = cadd(za, cconj(zb)), d1 = csub(za, cconj(zb));
    const real2 tt = cmuli(cmul(rtw[k], d1));
    *(real2 *)(FB + (unsigned long long)(unsigned)FROW(k) * (unsigned)ldf + 2 * f0) = mk2((s1.x - tt.x) * sc, (s1.y - tt.y) * sc);
  }
  R16_STAMP(3);
  R16_STAMP_END(4);
}

template <int R1>
EMI_KERNEL_LB2(r16_threads(R1), EMI_R16_WAVES_INV(R1)) void k_fft_inv_r16(EmiGeomDev g, FftTabDev T, FftLaunchDev Lc, const GridFld *flds, int nfld, const real_t *FB,
                                                   int ldf, int nproma) {
  constexpr int H = R1 / 2, NT = r16_threads(R1), S = 256 * R1;
  constexpr unsigned SZ2 = sizeof(real2);
  EMI_LDS_DECL;
  char *lds = EMI_LDS_PTR;
  real2 *tw2s = (real2 *)(lds + R1 * R16_ROWP * 8), *zbuf = (real2 *)lds;
  const unsigned t = (unsigned)EMI_TID;
  const bool edge = (NT == 256) ? true : (t < 256u);
  const int bid = (int)xcd_swizzle(EMI_BID, Lc.nblocks, 8);
  const int li = bid / Lc.nchunk;
  const FftRowDev rw_ = Lc.rows[li];  // one 64-byte record (see k_fft_dir_r16)
  const int f0 = bid - li * Lc.nchunk;
  const int n = rw_.n, sz = rw_.sz, nmen = rw_.nmen;
  const real_t racthe = (real_t)rw_.racthe;
  const real_t adjw = (real_t)(rw_.rw / (double)rw_.n);  // DIR_TRANSAD only (Lc.adj)
  const int fb0 = rw_.fb0;
  const int *frow = g.fftrow ? g.fftrow + fb0 : nullptr;
  const real2 *rtw = (const real2 *)T.rtw + rw_.rtw_off;
  const real2 *chirp = (const real2 *)T.chirp + rw_.chirp_off;
  const EmiBuf b_ch = emi_buf(chirp, (unsigned)sz * SZ2);
  const EmiBuf b_tw = emi_buf((const real2 *)T.ptw + rw_.ptw_off0, 7u * 256u * SZ2);
  const EmiBuf b_bh = emi_buf((const real2 *)T.bhat + rw_.bhat_off, (unsigned)S * SZ2);
  R16_STAMP_BEGIN();
  if (t < 240u) tw2s[t] = ((const real2 *)T.tw256)[t];
  const GridFld gf = flds[f0];
  // stage 1 (FOURIER_IN + FSC): Z_k = (X_k + conj X_{sz-k}) + i w^k (X_k - conj X_{sz-k}), times conj(chirp), to LDS.
  // Round 4: branch-free, all Fourier-row loads of the thread first.  A wave-0 clock attribution had put 39 % of a workgroup's life
  // into this stage: the plain loop below compiles into four to five SERIALISED memory round trips per trip of the loop (row-table
  // branch, X_k behind its `k <= nmen` branch, X_{sz-k} behind its own, the FSC mode branches, the tables), twenty per row -- which
  // is also why round 3's "all loads first" experiments changed nothing: the waits sat at the ends of those branches.  With one task the
  // Fourier rows of a latitude are consecutive, so ONE buffer descriptor over rows 0 .. NMEN of this field range-checks k <= NMEN in
  // hardware (rows past NMEN read as zero), FSC is two uniform factors (x (a + i b k): a, b from the field mode), the tables come
  // through descriptors as well, and the second store is unconditional (k = 0 writes the unused slot sz; k = sz - k writes the same
  // value twice).  Rows addressed through the exchange-order table (several tasks) are looked up with a clamped index and selected.
  {
    constexpr int TRIPS = R1 / 4 + 1;  // pairs k = t + NT a < sz / 2 + 1 <= S / 4 + 1
    const int npair = sz / 2 + 1;
    const unsigned rowb = (unsigned)ldf * (unsigned)sizeof(real_t);
    const EmiBuf b_fb = emi_buf(FB + (unsigned long long)(unsigned)fb0 * (unsigned)ldf + 2 * gf.src, (unsigned)nmen * rowb + SZ2);
    const EmiBuf b_rtw = emi_buf(rtw, (unsigned)(sz + 1) * SZ2);
    real_t fa, fb;
    fin_factors(gf.mode, racthe, fa, fb);
    const real_t fs = Lc.adj ? adjw : (real_t)1.0;
    real2 xa[TRIPS], xb[TRIPS];
#pragma unroll
    for (int a = 0; a < TRIPS; a++) {
      const unsigned k = t + (unsigned)NT * a, k2 = (unsigned)sz - k;
      if (!frow) {  // (uniform) one descriptor: k > NMEN -- and the idle lanes k >= npair of the last trip, whose k2 wraps -- read zero
        xa[a] = emi_buf_ld<real2>(b_fb, k * rowb, 0);
        xb[a] = emi_buf_ld<real2>(b_fb, k2 * rowb, 0);
      } else {      // rows through the exchange-order table: clamped look-up, then a select
        const unsigned ka = k < (unsigned)nmen ? k : (unsigned)nmen, kb = k2 < (unsigned)nmen ? k2 : (unsigned)nmen;
        const real2 va = fin_raw(FB, frow[ka], ldf, gf.src), vb = fin_raw(FB, frow[kb], ldf, gf.src);
        xa[a] = k <= (unsigned)nmen ? va : mk2(0, 0);
        xb[a] = k2 <= (unsigned)nmen ? vb : mk2(0, 0);
      }
    }
    // ONE copy of the arithmetic for both ways of loading: the gathered fields of any decomposition stay bit-identical
#pragma unroll
    for (int a = 0; a < TRIPS; a++) {
      const unsigned k = t + (unsigned)NT * a, k2 = (unsigned)sz - k;
      if (k < (unsigned)npair) {
        const real2 wk = cconj(emi_buf_ld<real2>(b_rtw, k * SZ2, 0));
        const real2 ck = emi_buf_ld<real2>(b_ch, k * SZ2, 0), ck2 = emi_buf_ld<real2>(b_ch, k2 * SZ2, 0);  // k2 = sz (k = 0): zero, slot unused
        real2 zk, zk2;
        fin_pair(xa[a], xb[a], k, k2, fa, fb, fs, wk, ck, ck2, zk, zk2);
        zbuf[k] = zk;
        zbuf[k2] = zk2;
      }
    }
  }
  R16_STAMP(0);
  EMI_LDS_SYNC();
  real2 v[H];
  if (edge) {
#pragma unroll
    for (int a = 0; a < H; a++) {
      const unsigned l = t + 256u * a;
      v[a] = (l < (unsigned)sz) ? zbuf[l] : mk2(0, 0);
    }
  }
  EMI_LDS_SYNC();  // the plane is free for the exchanges
  R16_STAMP(1);
  r16_conv<R1, 1>(v, t, b_tw, b_bh, lds, tw2s);
  R16_STAMP(2);
  // stage 3 (TRLTOG local copy): z_i = conv_i conj(chirp_i) / S; x_{2i} = Re z_i, x_{2i+1} = Im z_i, straight from the registers
  if (edge) {
    const GridRow gr = grid_row(gf, rw_.gpoff, nproma);
    const bool flat = (gr.rem0 + (unsigned)n <= gr.np) && ((((uintptr_t)(gr.p0 + gr.rem0)) & (2 * sizeof(real_t) - 1)) == 0);
    if (flat) {
      const EmiBuf b_out = emi_buf(gr.p0 + gr.rem0, (unsigned)n * (unsigned)sizeof(real_t));
#pragma unroll
      for (int a = 0; a < H; a++) {
        const unsigned off = (t + 256u * a) * SZ2;
        emi_buf_st<real2>(b_out, off, 0, cmulc(v[a], emi_buf_ld<real2>(b_ch, off, 0)));  // i >= sz: dropped (1 / S: in the filter table)
      }
    } else {
#pragma unroll
      for (int a = 0; a < H; a++) {
        const unsigned i = t + 256u * a;
        if (i < (unsigned)sz) {
          const real2 z = cmulc(v[a], chirp[i]);
          if (grid_pair_ok(gr, 2u * i)) {
            *(real2 *)grid_ptr(gr, 2u * i) = z;
          } else {
            *grid_ptr(gr, 2u * i) = z.x;
            *grid_ptr(gr, 2u * i + 1) = z.y;
          }
        }
      }
    }
  }
  R16_STAMP(3);
  R16_STAMP_END(4);
}

// ==========================================================================================
// Split register-resident kernels k_fft_dir_r16p<R1> / k_fft_inv_r16p<R1> (round 6): a row whose half-length sz = 2 q is too long for ONE
// register-resident convolution (2 sz - 1 > 4096) as TWO of them.  One decimation step turns the complex transform of length sz into the
// transforms of its even and odd points, each of length q, and each of those is a chirp-z convolution of work length 256 R1 >= 2 q - 1 on
// the r16_conv chain above (NLOEN of an octahedral grid is a multiple of four, so q is an integer; the host checks it):
//     direct :  Z_k = E_k + w^k O_k,  Z_{k+q} = E_k - w^k O_k,  w = exp(-2 pi i / sz),  E / O = DFT_q of z_{2j} / z_{2j+1};
//     inverse:  z_{2j} = IDFT_q(Z_k + Z_{k+q}),  z_{2j+1} = IDFT_q((Z_k - Z_{k+q}) conj w^k).
// At TCo1279 these are the rows of 4100 .. 5136 points, which ran on the in-place LDS kernels k_fft_*_hot<23 | 24> (work lengths 4608 /
// 5120: 2.0 vector instructions per work point, 28.6 ns per row and field) -- the largest block of the FFT phase (48 ms per pair); at TCo2559
// (fp32) the rows of 4100 .. 8192 points (work lengths 4608 .. 8192 of the 1024-thread in-place kernels).  The grid row is read / written
// once as 32 contiguous bytes per lane (x_{4l} .. x_{4l+3} = z_{2l}, z_{2l+1}); the chirp, the filter spectrum and the twiddle tables are
// those of length q for both halves.
// The two convolutions run ONE AFTER THE OTHER on a 256-thread workgroup, and the half of the row that is not being convolved is PARKED in a
// second LDS region -- every thread parks and fetches its own slots, so no barrier is needed for it.  The chain keeps its 117 - 120 registers
// (fp64: 152 - 166 with the stage around it, three waves per SIMD; fp32: 114 - 124, four), nothing spills, and the exchanges couple four waves
// as in k_fft_*_r16.  LDS: plane + q complex numbers + the small twiddles (46 KB in fp64 at R1 = 10: three workgroups per CU instead of
// four; 34 KB in fp32: four).  Two earlier forms, both parity-green, both measured (profiles/r6_fft_experiments.txt): the other half held in
// REGISTERS (20 on top of a chain that needs 117 of 128: 264 - 292 bytes of scratch, reloaded between the exchange barriers -- slower than the
// kernels it was to replace), and two 4-wave halves side by side in a 512-thread workgroup, one convolution each (no scratch, but every
// barrier of the exchanges couples eight waves and fp64 has three times as many of them: fp64 54.4 against 49.6 ms per pair for these
// rows on the in-place kernels; this form: 44.6).
// Same arithmetic in every decomposition (one expression tree per stage): the gathered fields stay bit-identical.
// Replaces FTDIR / FTINV for these rows (ftdir_mod.F90:67-84, ftinv_mod.F90:65-84; FFTW plans of tpm_fftw.F90:251-377).
// ==========================================================================================
// FSC + the pairing of FOURIER_IN for one pair (k, k2 = sz - k), as fin_pair but without the chirp (applied behind the decimation step here)
EMI_DEVFN void fin_pair_nc(real2 xa, real2 xb, unsigned k, unsigned k2, real_t fa, real_t fb, real_t fs, real2 wk, real2 &zk, real2 &zk2) {
  EMI_FP_STRICT();
  const real_t ba = fb * (real_t)k, bb = fb * (real_t)k2;
  const real2 ya = mk2((xa.x * fa - xa.y * ba) * fs, (xa.y * fa + xa.x * ba) * fs);
  const real2 yb = mk2((xb.x * fa - xb.y * bb) * fs, (xb.y * fa + xb.x * bb) * fs);
  const real_t s1x = ya.x + yb.x, s1y = ya.y - yb.y, d1x = ya.x - yb.x, d1y = ya.y + yb.y;  // s = ya + conj yb, d = ya - conj yb
  const real_t px = wk.x * d1x - wk.y * d1y, py = wk.x * d1y + wk.y * d1x;                   // w d
  zk = mk2(s1x - py, s1y + px);                                                                // s + i w d
  const real_t s2x = yb.x + ya.x, s2y = yb.y - ya.y, d2x = yb.x - ya.x, d2y = yb.y + ya.y;
  const real_t qx = -wk.x * d2x - wk.y * d2y, qy = -wk.x * d2y + wk.y * d2x;                  // (-conj w) d
  zk2 = mk2(s2x - qy, s2y + qx);
}

EMI_DEVFN constexpr int r16p_lds_bytes(int R1) { return R1 * R16_ROWP * 8 + 128 * R1 * (int)sizeof(real2) + 240 * (int)sizeof(real2); }
#ifndef EMI_R16P_WAVES_F32
#define EMI_R16P_WAVES_F32 4  // fp32: 114 - 124 registers and 34 KB of LDS -- four workgroups per CU; fp64: 152 - 166 registers, 46 KB -- three
#endif
#define EMI_R16P_WAVES (sizeof(real_t) == 4 ? EMI_R16P_WAVES_F32 : 3)

template <int R1>
EMI_KERNEL_LB2(256, EMI_R16P_WAVES) void k_fft_dir_r16p(EmiGeomDev g, FftTabDev T, FftLaunchDev Lc, const GridFld *flds, int nfld, real_t *FB, int ldf, int nproma) {
  constexpr int H = R1 / 2, S = 256 * R1, PLB = R1 * R16_ROWP * 8;
  constexpr unsigned SZ2 = sizeof(real2);
  static_assert(r16_threads(R1) == 256, "k_fft_dir_r16p: 256 threads (R1 <= 16)");
  EMI_LDS_DECL;
  char *lds = EMI_LDS_PTR;
  real2 *zbuf = (real2 *)lds, *park = (real2 *)(lds + PLB), *tw2s = park + 128 * R1;
  const unsigned t = (unsigned)EMI_TID;
  const int bid = (int)xcd_swizzle(EMI_BID, Lc.nblocks, 8);
  const int li = bid / Lc.nchunk;
  const FftRowDev rw_ = Lc.rows[li];
  const int f0 = bid - li * Lc.nchunk;  // one field per workgroup
  const int n = rw_.n, sz = rw_.sz, nmen = rw_.nmen, q = sz >> 1;
  const int fb0 = rw_.fb0;
  const int *frow = g.fftrow ? g.fftrow + fb0 : nullptr;
  const real2 *rtw = (const real2 *)T.rtw + rw_.rtw_off;
  const EmiBuf b_ch = emi_buf((const real2 *)T.chirp + rw_.chirp_off, (unsigned)q * SZ2);
  const EmiBuf b_tw = emi_buf((const real2 *)T.ptw + rw_.ptw_off0, 7u * 256u * SZ2);
  const EmiBuf b_bh = emi_buf((const real2 *)T.bhat + rw_.bhat_off, (unsigned)S * SZ2);
  const EmiBuf b_rtw = emi_buf(rtw, (unsigned)(sz + 1) * SZ2);
  if (t < 240u) tw2s[t] = ((const real2 *)T.tw256)[t];
  const GridFld gf = flds[f0];
  // stage 1: z_{2l} = x_{4l} + i x_{4l+1} (kept) and z_{2l+1} = x_{4l+2} + i x_{4l+3} (parked), both times the chirp of length q; l = t + 256 a
  real2 v[H];
  {
    const GridRow gr = grid_row(gf, rw_.gpoff, nproma);
    const bool flat = (gr.rem0 + (unsigned)n <= gr.np) && ((((uintptr_t)(gr.p0 + gr.rem0)) & (2 * sizeof(real_t) - 1)) == 0);
    if (flat) {
      const EmiBuf b_in = emi_buf(gr.p0 + gr.rem0, (unsigned)n * (unsigned)sizeof(real_t));
#pragma unroll
      for (int a = 0; a < H; a++) {
        const unsigned l = t + 256u * a;
        const real2 c = emi_buf_ld<real2>(b_ch, l * SZ2, 0);
        v[a] = cmul(emi_buf_ld<real2>(b_in, 2u * l * SZ2, 0), c);
        park[l] = cmul(emi_buf_ld<real2>(b_in, (2u * l + 1u) * SZ2, 0), c);
      }
    } else {
#pragma unroll
      for (int a = 0; a < H; a++) {
        const unsigned l = t + 256u * a;
        real2 ze = mk2(0, 0), zo = mk2(0, 0);
        if (l < (unsigned)q) {
          if (grid_pair_ok(gr, 4u * l)) {
            ze = *(const real2 *)grid_ptr(gr, 4u * l);
          } else {
            ze.x = *grid_ptr(gr, 4u * l);
            ze.y = *grid_ptr(gr, 4u * l + 1);
          }
          if (grid_pair_ok(gr, 4u * l + 2)) {
            zo = *(const real2 *)grid_ptr(gr, 4u * l + 2);
          } else {
            zo.x = *grid_ptr(gr, 4u * l + 2);
            zo.y = *grid_ptr(gr, 4u * l + 3);
          }
        }
        const real2 c = emi_buf_ld<real2>(b_ch, l * SZ2, 0);
        v[a] = cmul(ze, c);
        park[l] = cmul(zo, c);
      }
    }
  }
  r16_conv<R1, 0>(v, t, b_tw, b_bh, lds, tw2s);
  // E_l = conv_l chirp_l takes the slot of the odd point it replaces (same thread, same slot: no barrier)
#pragma unroll
  for (int a = 0; a < H; a++) {
    const unsigned l = t + 256u * a;
    const real2 e = cmul(v[a], emi_buf_ld<real2>(b_ch, l * SZ2, 0));
    v[a] = park[l];
    park[l] = e;
  }
  r16_conv<R1, 0>(v, t, b_tw, b_bh, lds, tw2s);
  EMI_LDS_SYNC();  // every thread has read its last plane values
  // Z_i = E_i + w^i O_i -> park slot i, Z_{i+q} = E_i - w^i O_i -> plane slot i (both are read by other threads below)
#pragma unroll
  for (int a = 0; a < H; a++) {
    const unsigned i = t + 256u * a;
    if (i < (unsigned)q) {
      const real2 o = cmul(v[a], emi_buf_ld<real2>(b_ch, i * SZ2, 0));
      const real2 tw = cmul(emi_buf_ld<real2>(b_rtw, 2u * i * SZ2, 0), o), e = park[i];
      park[i] = cadd(e, tw);
      zbuf[i] = csub(e, tw);
    }
  }
  EMI_LDS_SYNC();
  // stage 3 (FOURIER_OUT): Z_k = park[k] (k < q) | zbuf[k - q] (k >= q)
  const real_t sc = (real_t)0.5 * (Lc.adj ? (real_t)1.0 : (real_t)(rw_.rw / (double)n)) * fft_dir_mode_scale(gf.mode, (real_t)rw_.racthe);
  for (int k = (int)t; k <= nmen; k += 256) {
    const int kb = (k == 0) ? 0 : sz - k;
    const real2 za = k < q ? park[k] : zbuf[k - q], zb = kb < q ? park[kb] : zbuf[kb - q];
    const real2 s1 = cadd(za, cconj(zb)), d1 = csub(za, cconj(zb));
    const real2 tt = cmuli(cmul(rtw[k], d1));
    *(real2 *)(FB + (unsigned long long)(unsigned)FROW(k) * (unsigned)ldf + 2 * f0) = mk2((s1.x - tt.x) * sc, (s1.y - tt.y) * sc);
  }
}

template <int R1>
EMI_KERNEL_LB2(256, EMI_R16P_WAVES) void k_fft_inv_r16p(EmiGeomDev g, FftTabDev T, FftLaunchDev Lc, const GridFld *flds, int nfld, const real_t *FB, int ldf, int nproma) {
  constexpr int H = R1 / 2, S = 256 * R1, PLB = R1 * R16_ROWP * 8;
  constexpr unsigned SZ2 = sizeof(real2);
  static_assert(r16_threads(R1) == 256, "k_fft_inv_r16p: 256 threads (R1 <= 16)");
  EMI_LDS_DECL;
  char *lds = EMI_LDS_PTR;
  real2 *zbuf = (real2 *)lds, *park = (real2 *)(lds + PLB), *tw2s = park + 128 * R1;
  const unsigned t = (unsigned)EMI_TID;
  const int bid = (int)xcd_swizzle(EMI_BID, Lc.nblocks, 8);
  const int li = bid / Lc.nchunk;
  const FftRowDev rw_ = Lc.rows[li];
  const int f0 = bid - li * Lc.nchunk;
  const int n = rw_.n, sz = rw_.sz, nmen = rw_.nmen, q = sz >> 1;
  const real_t racthe = (real_t)rw_.racthe;
  const real_t adjw = (real_t)(rw_.rw / (double)rw_.n);  // DIR_TRANSAD only (Lc.adj)
  const int fb0 = rw_.fb0;
  const int *frow = g.fftrow ? g.fftrow + fb0 : nullptr;
  const real2 *rtw = (const real2 *)T.rtw + rw_.rtw_off;
  const real2 *chirp = (const real2 *)T.chirp + rw_.chirp_off;
  const EmiBuf b_ch = emi_buf(chirp, (unsigned)q * SZ2);
  const EmiBuf b_tw = emi_buf((const real2 *)T.ptw + rw_.ptw_off0, 7u * 256u * SZ2);
  const EmiBuf b_bh = emi_buf((const real2 *)T.bhat + rw_.bhat_off, (unsigned)S * SZ2);
  const EmiBuf b_rtw = emi_buf(rtw, (unsigned)(sz + 1) * SZ2);
  if (t < 240u) tw2s[t] = ((const real2 *)T.tw256)[t];
  const GridFld gf = flds[f0];
  // stage 1 (FOURIER_IN + FSC + decimation): the thread of the pair (k, q - k), k <= q / 2, loads X_k, X_{sz-k}, X_{q-k}, X_{q+k}, forms Z_k, Z_{sz-k},
  // Z_{q-k}, Z_{q+k} and from them the spectrum of the even points (Z_j + Z_{j+q}) conj c_j -> plane slot j, and of the odd points
  // (Z_j - Z_{j+q}) conj w^j conj c_j -> park slot j, for j = k and j = q - k
  {
    constexpr int TRIPS = R1 / 4 + 1;  // pairs k = t + 256 a <= q / 2 <= S / 8
    const int npair = q / 2 + 1;
    const unsigned rowb = (unsigned)ldf * (unsigned)sizeof(real_t);
    const EmiBuf b_fb = emi_buf(FB + (unsigned long long)(unsigned)fb0 * (unsigned)ldf + 2 * gf.src, (unsigned)nmen * rowb + SZ2);
    real_t fa, fb;
    fin_factors(gf.mode, racthe, fa, fb);
    const real_t fs = Lc.adj ? adjw : (real_t)1.0;
    real2 xa[TRIPS], xb[TRIPS], xc[TRIPS], xd[TRIPS];
#pragma unroll
    for (int a = 0; a < TRIPS; a++) {
      const unsigned k = t + 256u * a;
      const bool on = k < (unsigned)npair;
      const unsigned k2 = (unsigned)sz - k, k3 = on ? (unsigned)q - k : 0u, k4 = (unsigned)q + k;
      if (!frow) {
        xa[a] = emi_buf_ld<real2>(b_fb, k * rowb, 0);
        xb[a] = emi_buf_ld<real2>(b_fb, k2 * rowb, 0);
        xc[a] = emi_buf_ld<real2>(b_fb, k3 * rowb, 0);
        xd[a] = emi_buf_ld<real2>(b_fb, k4 * rowb, 0);
      } else {
        const unsigned nm = (unsigned)nmen;
        const real2 va = fin_raw(FB, frow[k < nm ? k : nm], ldf, gf.src), vb = fin_raw(FB, frow[k2 < nm ? k2 : nm], ldf, gf.src);
        const real2 vc = fin_raw(FB, frow[k3 < nm ? k3 : nm], ldf, gf.src), vd = fin_raw(FB, frow[k4 < nm ? k4 : nm], ldf, gf.src);
        xa[a] = k <= nm ? va : mk2(0, 0);
        xb[a] = k2 <= nm ? vb : mk2(0, 0);
        xc[a] = k3 <= nm ? vc : mk2(0, 0);
        xd[a] = k4 <= nm ? vd : mk2(0, 0);
      }
    }
#pragma unroll
    for (int a = 0; a < TRIPS; a++) {
      const unsigned k = t + 256u * a;
      if (k < (unsigned)npair) {
        const unsigned k2 = (unsigned)sz - k, k3 = (unsigned)q - k, k4 = (unsigned)q + k;
        real2 z1, z2, z3, z4;
        fin_pair_nc(xa[a], xb[a], k, k2, fa, fb, fs, cconj(emi_buf_ld<real2>(b_rtw, k * SZ2, 0)), z1, z2);    // Z_k, Z_{sz-k}
        fin_pair_nc(xc[a], xd[a], k3, k4, fa, fb, fs, cconj(emi_buf_ld<real2>(b_rtw, k3 * SZ2, 0)), z3, z4);  // Z_{q-k}, Z_{q+k}
        const real2 c1 = emi_buf_ld<real2>(b_ch, k * SZ2, 0), c3 = emi_buf_ld<real2>(b_ch, k3 * SZ2, 0);      // k3 = q (k = 0): zero, slot unused
        const real2 w1 = emi_buf_ld<real2>(b_rtw, 2u * k * SZ2, 0), w3 = emi_buf_ld<real2>(b_rtw, 2u * k3 * SZ2, 0);
        zbuf[k] = cmulc(cadd(z1, z4), c1);
        park[k] = cmulc(cmulc(csub(z1, z4), w1), c1);
        if (k3 < (unsigned)q) {
          zbuf[k3] = cmulc(cadd(z3, z2), c3);
          park[k3] = cmulc(cmulc(csub(z3, z2), w3), c3);
        }
      }
    }
  }
  EMI_LDS_SYNC();
  real2 v[H];
#pragma unroll
  for (int a = 0; a < H; a++) {
    const unsigned l = t + 256u * a;
    v[a] = (l < (unsigned)q) ? zbuf[l] : mk2(0, 0);
  }
  EMI_LDS_SYNC();  // the plane is free for the exchanges
  r16_conv<R1, 1>(v, t, b_tw, b_bh, lds, tw2s);
  // z_{2l} = conv_l conj(chirp_l) / S waits in the slot of the odd spectrum value it replaces (same thread, same slot)
#pragma unroll
  for (int a = 0; a < H; a++) {
    const unsigned l = t + 256u * a;
    const real2 z = cmulc(v[a], emi_buf_ld<real2>(b_ch, l * SZ2, 0));
    v[a] = (l < (unsigned)q) ? park[l] : mk2(0, 0);
    if (l < (unsigned)q) park[l] = z;
  }
  r16_conv<R1, 1>(v, t, b_tw, b_bh, lds, tw2s);
  // stage 3 (TRLTOG local copy): x_{4l} .. x_{4l+3} = Re, Im z_{2l}, Re, Im z_{2l+1}: 32 contiguous bytes per lane
  {
    const GridRow gr = grid_row(gf, rw_.gpoff, nproma);
    const bool flat = (gr.rem0 + (unsigned)n <= gr.np) && ((((uintptr_t)(gr.p0 + gr.rem0)) & (2 * sizeof(real_t) - 1)) == 0);
    if (flat) {
      const EmiBuf b_out = emi_buf(gr.p0 + gr.rem0, (unsigned)n * (unsigned)sizeof(real_t));
#pragma unroll
      for (int a = 0; a < H; a++) {
        const unsigned l = t + 256u * a;
        if (l < (unsigned)q) {
          emi_buf_st<real2>(b_out, 2u * l * SZ2, 0, park[l]);
          emi_buf_st<real2>(b_out, (2u * l + 1u) * SZ2, 0, cmulc(v[a], emi_buf_ld<real2>(b_ch, l * SZ2, 0)));
        }
      }
    } else {
#pragma unroll
      for (int a = 0; a < H; a++) {
        const unsigned l = t + 256u * a;
        if (l < (unsigned)q) {
          const real2 ze = park[l], zo = cmulc(v[a], chirp[l]);
          if (grid_pair_ok(gr, 4u * l)) {
            *(real2 *)grid_ptr(gr, 4u * l) = ze;
          } else {
            *grid_ptr(gr, 4u * l) = ze.x;
            *grid_ptr(gr, 4u * l + 1) = ze.y;
          }
          if (grid_pair_ok(gr, 4u * l + 2)) {
            *(real2 *)grid_ptr(gr, 4u * l + 2) = zo;
          } else {
            *grid_ptr(gr, 4u * l + 2) = zo.x;
            *grid_ptr(gr, 4u * l + 3) = zo.y;
          }
        }
      }
    }
  }
}

#include "emi_mr_body.h"

// ==========================================================================================
// k_gridcopy: TRLTOG / TRGTOL between the V-sets (NPRTRV > 1; trltog_mod.F90:18-964, trgtol_mod.F90:18-968).  Fourier space
// holds the fields of one V-set on the latitudes of a whole W-set band; grid space holds ALL fields on the sub-band of one
// task.  The re-distribution is an all-to-all-v among the NPRTRV tasks of a band; this kernel packs / unpacks its blocks:
// nf fields, npts points, element p of field f from src[f] (point sp0 + p) to dst[f] (point dp0 + p), both addressed as
// NPROMA-blocked grid arrays (GridFld; a dense [field][point] block is the case nf_arr = 1, NPROMA >= npts).
// ==========================================================================================
EMI_KERNEL_LB(256) void k_gridcopy(const GridFld *src, const GridFld *dst, int nf, long long sp0, long long dp0, int npts, int snp, int dnp) {
  const long long id = (long long)EMI_BID * 256 + EMI_TID;
  const int f = (int)(id / npts);
  if (f >= nf) return;
  const long long p = id - (long long)f * npts;
  const GridFld s = src[f], d = dst[f];
  const long long sp = sp0 + p, dp = dp0 + p;
  const real_t v = ((const real_t *)s.base)[((sp / snp) * s.nf_arr + s.fidx) * (long long)snp + sp % snp];
  ((real_t *)d.base)[((dp / dnp) * d.nf_arr + d.fidx) * (long long)dnp + dp % dnp] = v;
}

// ==========================================================================================
// k_legpol: SUPOLF (supolf_mod.F90:13-251) for m >= 2 on the device -- the normalised associated
// Legendre functions P_n^m(mu), n = m+par, m+par+2, ..., of one (wavenumber, parity, latitude) per
// thread, by the reference's 4-term recurrence in n with its 1e+-100 rescaling, written straight into
// the panels P[k][lat] and PT[lat][k].  The arithmetic is the oracle's / the reference's operation for
// operation in double (fp contraction off: IEEE +,-,*,/,sqrt only), whatever the library precision;
// the rescaling bookkeeping is streamed: a value is final once the step four degrees above it has been
// taken (only that step and earlier ones can rescale it, supolf_mod.F90:222-236), so the two values in
// flight live in registers and nothing is revisited.  (m = 0, 1 use the ordinary recurrence on the host.)
// ==========================================================================================
EMI_DEVFN double legpol_final(double v, int corr) {
  const double big = 1.0e+100, eps = 2.220446049250313e-16;
  for (int j = 1; j <= corr; j++) {
    v /= big;
    if (v < eps) v = eps;  // sic: no ABS in the reference (supolf_mod.F90:241-243)
  }
  return v;
}
EMI_KERNEL_LB(64) void k_legpol(EmiGeomDev g, LegPolDev a) {
#ifndef EMI_CPU_EMU
#pragma clang fp contract(off)
#endif
  const int ml = a.blk[2 * EMI_BID], par = a.blk[2 * EMI_BID + 1] >> 16, jt = a.blk[2 * EMI_BID + 1] & 0xffff;
  const int m = g.mval[ml], N = g.nsmax, nmax = a.nmax;
  const int nd = g.lbase[ml + 1] - g.lbase[ml], j = jt * 64 + EMI_TID;
  if (j >= nd) return;
  const int ld = g.ldp[ml], ldk = g.ldk[ml];
  real_t *Pp = (real_t *)g.P + (par ? g.offA[ml] : g.offS[ml]) + j;                      // + k * ld
  real_t *PTp = (real_t *)g.PT + (par ? g.offTA[ml] : g.offTS[ml]) + (long long)j * ldk;  // + k
  const double *dcl = a.dcl + (long long)ml * (nmax + 1), *ddl = a.ddl + (long long)ml * (nmax + 1);
  const double eps = 2.220446049250313e-16, big = 1.0e+100, small = 1.0e-100;
  double x = a.mu[a.ndgnh - nd + j];
  double c2 = 1.0 - x * x, cs = sqrt(c2);
  if (fabs(cs) <= eps) {
    x = 1.0;
    cs = 0.0;
    c2 = 0.0;
  }
  int corr3 = 0;
  double lsita = 1.0;
  for (int i = 1; i <= m / 2; i++) {
    lsita *= c2;
    if (fabs(lsita) < small) {
      lsita *= big;
      corr3++;
    }
  }
  if (m & 1) lsita *= cs;
  // starting values n = m .. m+3 (supolf_mod.F90:177-205); this thread needs those of its parity
  double zfac = a.zfac[ml], zfac0 = 1.0, zfac1 = 1.0, mult = 0.0, st[4] = {0.0, 0.0, 0.0, 0.0};
  const int icmax = nmax - m < 3 ? nmax - m : 3;
  for (int ic = 0; ic <= icmax; ic++) {
    zfac0 *= (double)(2 * m + ic);
    switch (ic) {
      case 0: zfac1 = 1.0; mult = zfac; break;
      case 1: zfac1 = 1.0; zfac *= (double)(2 * m + 1); mult = zfac * x; break;
      case 2: zfac1 = 2.0; mult = 0.5 * zfac * ((double)(2 * m + 3) * x * x - 1.0); break;
      default: zfac1 = 6.0; zfac *= (double)(2 * m + 3); mult = (1.0 / 6.0) * x * zfac * ((double)(2 * m + 5) * x * x - 3.0); break;
    }
    st[ic] = lsita * mult * sqrt(2.0 * ((double)(m + ic) + 0.5) * zfac1 / zfac0);
  }
  const int nk = (N + 1 - m - par) / 2 + 1;  // outputs n = m + par + 2 k <= N + 1  (>= 1 for m <= N)
  double w0 = st[par], w1 = st[par + 2];
  int ev = 0, kout = 0;
  for (int n = m + par + 4; n <= nmax; n += 2) {
    if (fabs(w0) > big) {
      w0 /= big;
      w1 /= big;
      ev++;
    }
    const double nw = ((x * x - ddl[n - 2]) * w1 - dcl[n - 4] * w0) / dcl[n - 2];
    if (kout < nk) {
      const double v = legpol_final(w0, corr3 - ev);
      Pp[(long long)kout * ld] = (real_t)v;
      PTp[kout] = (real_t)v;
    }
    kout++;
    w0 = w1;
    w1 = nw;
  }
  // the last two values (or the starting values when no step was taken)
  if (kout < nk) {
    const double v = legpol_final(w0, corr3 - ev);
    Pp[(long long)kout * ld] = (real_t)v;
    PTp[kout] = (real_t)v;
  }
  kout++;
  if (kout < nk && m + par + 2 <= nmax) {
    const double v = legpol_final(w1, corr3 - ev);
    Pp[(long long)kout * ld] = (real_t)v;
    PTp[kout] = (real_t)v;
  }
}

// ==========================================================================================
// k_specnorm: SPNORMD (spnormd_mod.F90:40-57).  One block per field; deterministic order.
// ==========================================================================================
EMI_KERNEL_LB(256) void k_specnorm(EmiGeomDev g, long long nspec2, const real_t *sp, int stride, double *out) {
  EMI_LDS_DECL;
  double *red = (double *)EMI_LDS_PTR;
  const int f = EMI_BID;
  double s = 0.0;
  for (long long i = EMI_TID; i < nspec2; i += EMI_NTHREADS) {
    double v = (double)sp[i * stride + f];
    s += g.specw[i] * v * v;  // m = 0: real parts weight 1, imaginary 0; m > 0: weight 2
  }
  red[EMI_TID] = s;
  EMI_SYNC();
  for (int st = EMI_NTHREADS / 2; st > 0; st >>= 1) {
    if (EMI_TID < st) red[EMI_TID] += red[EMI_TID + st];
    EMI_SYNC();
  }
  if (EMI_TID == 0) out[f] = red[0];  // sum of squares; the host takes the root (after the task sum)
}

// ==========================================================================================
// k_gpnorm: GPNORM_TRANS_CTL (cpu/internal/gpnorm_trans_ctl_mod.F90:170-210): for one (latitude, field) the sum over the longitudes
// of the row, accumulated in double (the reference's ZAVE is JPRD), and the row's minimum and maximum.  One workgroup per
// (latitude, field); a fixed reduction tree, so the result does not depend on the decomposition.  out: [3][nfld][nlat].
// ==========================================================================================
EMI_KERNEL_LB(256) void k_gpnorm(const int *rowoff /* [nlat + 1] first local point of every latitude */, int nlat, const real_t *gp, int nf_arr,
                                 int nfld, int nproma, double *out) {
  EMI_LDS_DECL;
  double *red = (double *)EMI_LDS_PTR;
  const int f = EMI_BID / nlat, j = EMI_BID - f * nlat;
  const int p0 = rowoff[j], n = rowoff[j + 1] - p0;
  double s = 0.0, mn = 0.0, mx = 0.0;
  bool any = false;
  for (int i = EMI_TID; i < n; i += EMI_NTHREADS) {
    const long long p = (long long)p0 + i, blk = p / nproma;
    const double v = (double)gp[(blk * nf_arr + f) * nproma + (p - blk * nproma)];
    s += v;
    mn = any ? (v < mn ? v : mn) : v;
    mx = any ? (v > mx ? v : mx) : v;
    any = true;
  }
  if (!any) mn = mx = (double)gp[((long long)(p0 / nproma) * nf_arr + f) * nproma + p0 % nproma];  // rows are never empty: the first point
  const int T = EMI_NTHREADS;
  red[EMI_TID] = s, red[T + EMI_TID] = mn, red[2 * T + EMI_TID] = mx;
  EMI_SYNC();
  for (int st = T / 2; st > 0; st >>= 1) {
    if (EMI_TID < st) {
      red[EMI_TID] += red[EMI_TID + st];
      red[T + EMI_TID] = red[T + EMI_TID] < red[T + EMI_TID + st] ? red[T + EMI_TID] : red[T + EMI_TID + st];
      red[2 * T + EMI_TID] = red[2 * T + EMI_TID] > red[2 * T + EMI_TID + st] ? red[2 * T + EMI_TID] : red[2 * T + EMI_TID + st];
    }
    EMI_SYNC();
  }
  if (EMI_TID == 0) {
    const long long o = (long long)f * nlat + j, nn = (long long)nfld * nlat;
    out[o] = red[0], out[nn + o] = red[T], out[2 * nn + o] = red[2 * T];
  }
}

// ==========================================================================================
// k_vd2uv: VORDIV_TO_UV (cpu/external/vordiv_to_uv.F90 -> cpu/internal/vd2uv_mod.F90:79-120 = PRFI1B + VDTUV, vdtuv_mod.F90:97-143):
//   U_n = i m L_n D_n + (n-1) e_n L_(n-1) vor_(n-1) - (n+2) e_(n+1) L_(n+1) vor_(n+1)
//   V_n = i m L_n vor_n - (n-1) e_n L_(n-1) D_(n-1) + (n+2) e_(n+1) L_(n+1) D_(n+1),   L_n = RLAPIN(n) = -a^2 / (n (n+1)),
// coefficients n <= NSMAX, times 1 / a.  One thread per (spectral pair, field); no resolution handle is needed, only NSMAX and
// the wavenumbers of the task (the reference sets up a spectral-only resolution for the call).
// ==========================================================================================
EMI_KERNEL_LB(256) void k_vd2uv(Vd2uvDev d, const real_t *vor, const real_t *div, real_t *pu, real_t *pv) {
  const long long gi = (long long)EMI_BID * EMI_NTHREADS + EMI_TID;
  const int ip = (int)(gi / d.nfld), f = (int)(gi - (long long)ip * d.nfld);
  if (ip >= d.nspec2 / 2) return;
  const int ml = d.pairm[ip], m = d.mval[ml], N = d.nsmax;
  const long long isp = 2LL * ip;  // Re(m, n)
  const int n = m + (int)((isp - d.nasm0[ml]) >> 1);
  const double *eps = d.eps + d.ebase[ml] - m;
  auto get = [&](const real_t *a, long long i) { return mk2(a[i * d.nfld + f], m == 0 ? (real_t)0.0 : a[(i + 1) * d.nfld + f]); };
  const real_t zkm = (real_t)m, l_n = (real_t)d.lapin[n + 1], l_nm1 = (real_t)d.lapin[n], l_np1 = (real_t)d.lapin[n + 2];
  const real_t c1 = (real_t)(n - 1) * (real_t)eps[n] * l_nm1, c2 = (real_t)(n + 2) * (real_t)eps[n + 1] * l_np1, ra = (real_t)d.rra;
  const real2 z = mk2(0, 0);
  const real2 vm = n - 1 >= m ? get(vor, isp - 2) : z, vp = n + 1 <= N ? get(vor, isp + 2) : z, v0 = get(vor, isp);
  const real2 dm = n - 1 >= m ? get(div, isp - 2) : z, dp = n + 1 <= N ? get(div, isp + 2) : z, d0 = get(div, isp);
  real2 u = mk2(-zkm * l_n * d0.y + (c1 * vm.x - c2 * vp.x), zkm * l_n * d0.x + (c1 * vm.y - c2 * vp.y));
  real2 v = mk2(-zkm * l_n * v0.y - (c1 * dm.x - c2 * dp.x), zkm * l_n * v0.x - (c1 * dm.y - c2 * dp.y));
  if (m == 0) u.y = 0.0, v.y = 0.0;
  pu[isp * d.nfld + f] = u.x * ra, pu[(isp + 1) * d.nfld + f] = u.y * ra;
  pv[isp * d.nfld + f] = v.x * ra, pv[(isp + 1) * d.nfld + f] = v.y * ra;
}

#undef FROW
#undef emi_mfma_f64_16x16x4
