// emi_kernels.h -- device kernels of the MI355X spherical-harmonic transform.
//
// Pipeline (one field batch; see DESIGN.md §3 for layouts):
//   INV:  k_prepack_inv  (PRFI1B+VDTUV+SPNSDE)  user spectral -> W  [m][r=n-m][f][c]
//         k_leg_inv      (LEINV+ASRE1B, fp64 MFMA)          W -> FB [lat][m][f][c]
//         k_fft_inv      (FOURIER_IN+FSC+FTINV+TRLTOG)      FB -> user grid arrays
//   DIR:  k_fft_dir      (TRGTOL+FTDIR+FOURIER_OUT, w/NLOEN & 1/(a cos) folded in)  grid -> FB
//         k_leg_dir      (PRFI2B+LEDIR, fp64 MFMA)          FB -> W
//         k_postpack_dir (UVTVD+UPDSP)                      W -> user spectral
// Reference routines named in () are the CPU routines whose results each kernel reproduces
// (cpu/internal/*_mod.F90); none of the reference's GPU code is used.
#pragma once
#include "emi_rt.h"

// ------------------------------------------------------------------------------------------
// device-visible descriptors
// ------------------------------------------------------------------------------------------
struct EmiGeomDev {
  int nsmax, ndgl, ndgnh, ngptot;
  const int *nloen, *nmen, *ndglu, *gpoff;
  const int *nasm0;   // [nsmax+1] 0-based index of Re(m, n=m) in the user spectral dimension
  const int *fbase;   // [ndgl+1]  Fourier rows (lat,m<=NMEN) before latitude
  const int *wbase;   // [nsmax+2] packed-spectral rows before m (padded to 16)
  const int *wrows;   // [nsmax+1] padded row count of m (multiple of 16)
  const int *rowm;    // [wbase[nsmax+1]] row -> m
  const int *ebase;   // [nsmax+1] index of eps(n=m) in eps[] (n = m..N+2)
  const double *eps;  // REPSNM
  const double *lapin;  // RLAPIN(n) at [n+1], n=-1..N+2
  const double *rw, *racthe;
  const double *P;             // Legendre panels
  const long long *offS, *offA;  // [nsmax+1] element offsets of the even/odd (n-m) panels
  const int *ldp;              // [nsmax+1] padded latitude count (multiple of 64)
  const int *lattile_pref;     // [nsmax+2] prefix of ceil(ndglu/64)
  const int *ktile_pref;       // [nsmax+2] prefix of ceil((wrows/2)/64)
};

enum { SPK_COPY = 0, SPK_U = 1, SPK_V = 2, SPK_NSD = 3 };
struct SpecSrc {  // one Legendre-space input field of the inverse transform
  const double *a, *b;  // element (ispec) of field = a[ispec*sa + ia]
  int sa, ia, sb, ib;
  int kind, pad_;
};
enum { SPO_COPY = 0, SPO_VOR = 1, SPO_DIV = 2 };
struct SpecDst {  // one spectral output field of the direct transform
  double *dst;
  int stride, idx;
  int kind, src0, src1, pad_;  // src*: field index in W (U and V for vor/div)
};
enum { GM_PLAIN = 0, GM_ACOS = 1, GM_EWDER = 2, GM_EWDER_UV = 3 };
struct GridFld {  // one Fourier-space field <-> one user grid field
  double *base;   // array base; element (p) = base[((p/nproma)*nf_arr + fidx)*nproma + p%nproma]
  int nf_arr, fidx;
  int mode, src;  // src: field index inside FB (inverse only)
};

struct FftPlanDev {
  int n;      // row length (NLOEN)
  int sz;     // logical complex transform size: n/2 (n even) or n (cmode)
  int S;      // LDS work size per field (complex): sz, or Bluestein length L
  int cmode;  // 1: odd n, complex transform of the real row
  int blue;   // 1: Bluestein
  int nfac;
  int fac[14];
  int tw_off, perm_off, rtw_off, chirp_off, bhat_off;
  int fbk;  // fields per workgroup
  int lds_class, pad_;
};
struct FftTabDev {
  const d2 *tw;                // e^{-2 pi i k/S}
  const unsigned short *perm;  // DIT input position of natural index
  const d2 *rtw;               // e^{-2 pi i k/n}, k=0..sz/2
  const d2 *chirp;             // e^{-i pi k^2/sz}
  const d2 *bhat;              // DFT_L of the chirp filter, at perm positions
  const FftPlanDev *plans;
  const int *planid;           // [ndgl]
};

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------
EMI_DEVFN d2 mk2(double x, double y) {
  d2 r;
  r.x = x;
  r.y = y;
  return r;
}
EMI_DEVFN d2 cmul(d2 a, d2 b) { return mk2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
EMI_DEVFN d2 cmulc(d2 a, d2 b) { return mk2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }  // a*conj(b)
EMI_DEVFN d2 cadd(d2 a, d2 b) { return mk2(a.x + b.x, a.y + b.y); }
EMI_DEVFN d2 csub(d2 a, d2 b) { return mk2(a.x - b.x, a.y - b.y); }
EMI_DEVFN d2 cconj(d2 a) { return mk2(a.x, -a.y); }
EMI_DEVFN d2 cscale(d2 a, double s) { return mk2(a.x * s, a.y * s); }
EMI_DEVFN d2 cmuli(d2 a) { return mk2(-a.y, a.x); }  // i*a

EMI_DEVFN int upper_m(const int *pref, int nsmax, int t) {
  // largest m in [0,nsmax] with pref[m] <= t  (pref non-decreasing, pref[nsmax+1] > t)
  int lo = 0, hi = nsmax;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (pref[mid] <= t)
      lo = mid;
    else
      hi = mid - 1;
  }
  return lo;
}

// XCD-aware remap: consecutive logical tiles land on the same XCD (8 XCDs, round-robin
// dispatch), so tiles that share a Legendre panel share an L2.  Bijective for any grid size.
EMI_DEVFN long long xcd_swizzle(long long bid, long long nwg) {
  long long q = nwg >> 3, r = nwg & 7, x = bid & 7, k = bid >> 3;
  long long start = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return start + k;
}

// ==========================================================================================
// k_prepack_inv: PRFI1B (prfi1b_mod.F90:81-115) + VDTUV (vdtuv_mod.F90:97-143) + SPNSDE
// (spnsde_mod.F90:95-114).  One thread per (packed row, field); fields fastest.
//   W[(wbase[m]+r)*ldw + 2f + c],  r = n-m in [0, wrows[m])  (rows n > N+1 are zero)
// ==========================================================================================
EMI_DEVFN d2 spec_get(const double *a, int sa, int ia, long long isp, int m) {
  d2 v;
  v.x = a[isp * sa + ia];
  v.y = (m == 0) ? 0.0 : a[(isp + 1) * sa + ia];
  return v;
}

EMI_KERNEL_LB(256) void k_prepack_inv(EmiGeomDev g, const SpecSrc *flds, int nfld, int nfld_pad, double *W, int ldw,
                              long long nrows) {
  const int N = g.nsmax;
  long long total = nrows * nfld_pad;
  {
    const long long idx = (long long)EMI_BID * EMI_NTHREADS + EMI_TID;
    if (idx >= total) return;
    long long row = idx / nfld_pad;
    int f = (int)(idx - row * nfld_pad);
    d2 out = mk2(0.0, 0.0);
    if (f < nfld) {
      int m = g.rowm[row];
      int r = (int)(row - g.wbase[m]);
      int n = m + r;
      if (n <= N + 1) {
        SpecSrc s = flds[f];
        long long isp = g.nasm0[m] + 2LL * r;  // Re(m,n)
        if (s.kind == SPK_COPY) {
          if (n <= N) out = spec_get(s.a, s.sa, s.ia, isp, m);
        } else {
          const double *eps = g.eps + g.ebase[m] - m;  // eps[n], n=m..N+2
          double zn_m1 = (double)(n - 1), zn_p2 = (double)(n + 2);
          double e_n = eps[n], e_np1 = eps[n + 1];
          if (s.kind == SPK_NSD) {
            d2 fm = (n - 1 >= m) ? spec_get(s.a, s.sa, s.ia, isp - 2, m) : mk2(0, 0);
            d2 fp = (n + 1 <= N) ? spec_get(s.a, s.sa, s.ia, isp + 2, m) : mk2(0, 0);
            out.x = -zn_m1 * e_n * fm.x + zn_p2 * e_np1 * fp.x;
            out.y = -zn_m1 * e_n * fm.y + zn_p2 * e_np1 * fp.y;
          } else {
            // a = vorticity, b = divergence
            const double *pa = (s.kind == SPK_U) ? s.a : s.b;  // the field entering the +-(n-1),(n+2) terms
            const double *pb = (s.kind == SPK_U) ? s.b : s.a;  // the field entering the i*m term
            int sa = (s.kind == SPK_U) ? s.sa : s.sb, ia = (s.kind == SPK_U) ? s.ia : s.ib;
            int sb = (s.kind == SPK_U) ? s.sb : s.sa, ib = (s.kind == SPK_U) ? s.ib : s.ia;
            double l_n = g.lapin[n + 1], l_nm1 = g.lapin[n], l_np1 = g.lapin[n + 2];
            d2 xm = (n - 1 >= m) ? spec_get(pa, sa, ia, isp - 2, m) : mk2(0, 0);
            d2 xp = (n + 1 <= N) ? spec_get(pa, sa, ia, isp + 2, m) : mk2(0, 0);
            d2 y0 = (n <= N) ? spec_get(pb, sb, ib, isp, m) : mk2(0, 0);
            double zkm = (double)m;
            double c1 = zn_m1 * e_n * l_nm1, c2 = zn_p2 * e_np1 * l_np1;
            double sg = (s.kind == SPK_U) ? 1.0 : -1.0;
            // U = i m L_n D_n + c1 vor_{n-1} - c2 vor_{n+1};  V = i m L_n vor_n - c1 D_{n-1} + c2 D_{n+1}
            out.x = -zkm * l_n * y0.y + sg * (c1 * xm.x - c2 * xp.x);
            out.y = zkm * l_n * y0.x + sg * (c1 * xm.y - c2 * xp.y);
            if (m == 0) out.y = 0.0;
          }
        }
      }
    }
    *(d2 *)(W + row * ldw + 2 * f) = out;
  }
}

// ==========================================================================================
// k_postpack_dir: UVTVD (uvtvd_mod.F90:91-139) + UPDSP/UPDSPB (updsp_mod.F90:100-161,
// updspb_mod.F90:92-149).  One thread per (packed row with n<=N, output field).
// ==========================================================================================
EMI_KERNEL_LB(256) void k_postpack_dir(EmiGeomDev g, const SpecDst *flds, int nfld, const double *W, int ldw,
                               long long nrows) {
  const int N = g.nsmax;
  long long total = nrows * nfld;
  {
    const long long idx = (long long)EMI_BID * EMI_NTHREADS + EMI_TID;
    if (idx >= total) return;
    long long row = idx / nfld;
    int f = (int)(idx - row * nfld);
    int m = g.rowm[row];
    int r = (int)(row - g.wbase[m]);
    int n = m + r;
    if (n > N) return;
    SpecDst s = flds[f];
    d2 out;
    if (s.kind == SPO_COPY) {
      out = *(const d2 *)(W + row * ldw + 2 * s.src0);
    } else {
      const double *eps = g.eps + g.ebase[m] - m;
      // vor: x=V (i m term), y=U ; div: x=U, y=V with opposite sign on the n-terms
      int fx = (s.kind == SPO_VOR) ? s.src1 : s.src0;
      int fy = (s.kind == SPO_VOR) ? s.src0 : s.src1;
      double sg = (s.kind == SPO_VOR) ? 1.0 : -1.0;
      d2 x0 = *(const d2 *)(W + row * ldw + 2 * fx);
      d2 yp = *(const d2 *)(W + (row + 1) * ldw + 2 * fy);                        // n+1 (<= N+1 stored)
      d2 ym = (n - 1 >= m) ? *(const d2 *)(W + (row - 1) * ldw + 2 * fy) : mk2(0, 0);  // n-1
      double zkm = (double)m, c1 = (double)n * eps[n + 1], c2 = (double)(n + 1) * eps[n];
      // vor_n = i m V_n - n e_{n+1} U_{n+1} + (n+1) e_n U_{n-1}
      // div_n = i m U_n + n e_{n+1} V_{n+1} - (n+1) e_n V_{n-1}
      out.x = -zkm * x0.y + sg * (-c1 * yp.x + c2 * ym.x);
      out.y = zkm * x0.x + sg * (-c1 * yp.y + c2 * ym.y);
      if (m == 0 && n == 0) out = mk2(0, 0);  // updsp_mod.F90:113-126
    }
    if (m == 0) out.y = 0.0;  // updspb_mod.F90:106,117
    long long isp = g.nasm0[m] + 2LL * r;
    s.dst[isp * s.stride + s.idx] = out.x;
    s.dst[(isp + 1) * s.stride + s.idx] = out.y;
  }
}

// ==========================================================================================
// Legendre transforms on the fp64 matrix cores.
//   workgroup = 256 threads = 4 waves; MFMA v_mfma_f64_16x16x4_f64.
//   LDS tiles: As[2 parities][8 k][LG_LDA], Bs[2][8 k][LG_LDB]; the paddings (16 doubles) put
//   the two k-rows a half-wave reads with one ds_read_b64 on disjoint bank halves.
// ==========================================================================================
#define LG_THREADS 256
#define LG_BN 128
#define LG_LDA 80
#define LG_LDB 144
#define LG_LDS_BYTES ((2 * 8 * LG_LDA + 2 * 8 * LG_LDB) * 8)

// ---- inverse: FB[lat][m][col] = sum_n P[lat,n] W[m][n][col]; north = S+A, south = S-A
// (leinv_mod.F90:92-186 DGEMM('N','N') x2, asre1b_mod.F90:83-102)
// tile: 64 latitudes x 128 columns, both parities; wave (wm, wn) owns 32 lat x 64 col.
EMI_KERNEL_LB(256) void k_leg_inv(EmiGeomDev g, int ncoltiles, const double *W, int ldw, double *FB, int ldf, long long ntiles) {
  EMI_LDS_DECL;
  double *As = (double *)EMI_LDS_PTR;
  double *Bs = As + 2 * 8 * LG_LDA;
  const int tid = EMI_TID, w = tid >> 6, l = tid & 63;
  const int wm = w & 1, wn = w >> 1;
  long long tile = xcd_swizzle(EMI_BID, ntiles);
  int ct = (int)(tile % ncoltiles);
  int t2 = (int)(tile / ncoltiles);
  const int m = upper_m(g.lattile_pref, g.nsmax, t2);
  const int lt = t2 - g.lattile_pref[m];
  const int ld = g.ldp[m];
  const double *PS = g.P + g.offS[m], *PA = g.P + g.offA[m];
  const int lat0 = lt * 64, col0 = ct * LG_BN;
  const int nst = g.wrows[m] >> 4;
  const long long wb = g.wbase[m];

  v4d acc[2][2][4];
  for (int p = 0; p < 2; p++)
    for (int i = 0; i < 2; i++)
      for (int j = 0; j < 4; j++) acc[p][i][j] = (v4d){0.0, 0.0, 0.0, 0.0};

  const int arow = tid >> 5, ac2 = tid & 31;
  d2 ra0, ra1, rb[4];
  {
    ra0 = *(const d2 *)(PS + (long long)(arow)*ld + lat0 + 2 * ac2);
    ra1 = *(const d2 *)(PA + (long long)(arow)*ld + lat0 + 2 * ac2);
    for (int i = 0; i < 4; i++) {
      int idx = tid + 256 * i, brow = idx >> 6, bc2 = idx & 63;
      rb[i] = *(const d2 *)(W + (wb + brow) * ldw + col0 + 2 * bc2);
    }
  }
  for (int s = 0; s < nst; s++) {
    if (s > 0) EMI_SYNC();
    *(d2 *)(As + (0 * 8 + arow) * LG_LDA + 2 * ac2) = ra0;
    *(d2 *)(As + (1 * 8 + arow) * LG_LDA + 2 * ac2) = ra1;
    for (int i = 0; i < 4; i++) {
      int idx = tid + 256 * i, brow = idx >> 6, bc2 = idx & 63;
      *(d2 *)(Bs + ((brow & 1) * 8 + (brow >> 1)) * LG_LDB + 2 * bc2) = rb[i];
    }
    EMI_SYNC();
    if (s + 1 < nst) {
      ra0 = *(const d2 *)(PS + (long long)(8 * (s + 1) + arow) * ld + lat0 + 2 * ac2);
      ra1 = *(const d2 *)(PA + (long long)(8 * (s + 1) + arow) * ld + lat0 + 2 * ac2);
      for (int i = 0; i < 4; i++) {
        int idx = tid + 256 * i, brow = idx >> 6, bc2 = idx & 63;
        rb[i] = *(const d2 *)(W + (wb + 16 * (s + 1) + brow) * ldw + col0 + 2 * bc2);
      }
    }
    for (int p = 0; p < 2; p++)
      for (int ks = 0; ks < 2; ks++) {
        const int kk = 4 * ks + (l >> 4);
        double a[2], b[4];
        for (int i = 0; i < 2; i++) a[i] = As[(p * 8 + kk) * LG_LDA + wm * 32 + i * 16 + (l & 15)];
        for (int j = 0; j < 4; j++) b[j] = Bs[(p * 8 + kk) * LG_LDB + wn * 64 + j * 16 + (l & 15)];
        for (int i = 0; i < 2; i++)
          for (int j = 0; j < 4; j++) acc[p][i][j] = emi_mfma_f64_16x16x4(a[i], b[j], acc[p][i][j]);
      }
  }
  // epilogue (ASRE1B): rows = latitudes
  const int ndglu = g.ndglu[m] < g.ndgnh ? g.ndglu[m] : g.ndgnh;
  const int isl0 = g.ndgnh - ndglu;  // 0-based first northern latitude with m <= NMEN
  for (int i = 0; i < 2; i++)
    for (int q = 0; q < 4; q++) {
      int j = lat0 + wm * 32 + i * 16 + (l >> 4) + 4 * q;
      if (j < ndglu) {
        int latn = isl0 + j, lats = g.ndgl - 1 - latn;
        double *pn = FB + ((long long)g.fbase[latn] + m) * ldf + col0 + wn * 64 + (l & 15);
        double *ps = FB + ((long long)g.fbase[lats] + m) * ldf + col0 + wn * 64 + (l & 15);
        for (int jn = 0; jn < 4; jn++) {
          double sv = acc[0][i][jn][q], av = acc[1][i][jn][q];
          pn[jn * 16] = sv + av;
          ps[jn * 16] = sv - av;
        }
      }
    }
}

// ---- direct: W[m][n][col] = sum_lat P[lat,n] * (FB_north +- FB_south)[lat][col]
// (prfi2b_mod.F90:82-94, ledir_mod.F90:100-267 DGEMM('T','N') x2; Gaussian weights and
//  1/(a cos) were folded into FB by k_fft_dir)
// tile: 64 k (n-pairs) x 2 parities x 128 columns; wave (par, wn) owns 64 k x 64 col.
EMI_KERNEL_LB(256) void k_leg_dir(EmiGeomDev g, int ncoltiles, const double *FB, int ldf, double *W, int ldw, long long ntiles) {
  EMI_LDS_DECL;
  double *As = (double *)EMI_LDS_PTR;
  double *Bs = As + 2 * 8 * LG_LDA;
  const int tid = EMI_TID, w = tid >> 6, l = tid & 63;
  const int par = w & 1, wn = w >> 1;
  long long tile = xcd_swizzle(EMI_BID, ntiles);
  int ct = (int)(tile % ncoltiles);
  int t2 = (int)(tile / ncoltiles);
  const int m = upper_m(g.ktile_pref, g.nsmax, t2);
  const int kt = t2 - g.ktile_pref[m];
  const int ld = g.ldp[m];
  const double *PS = g.P + g.offS[m], *PA = g.P + g.offA[m];
  const int k0 = kt * 64, col0 = ct * LG_BN;
  const int nkpad = g.wrows[m] >> 1;
  const int ndglu = g.ndglu[m] < g.ndgnh ? g.ndglu[m] : g.ndgnh;
  const int isl0 = g.ndgnh - ndglu;
  const int nst = (ndglu + 7) >> 3;
  const long long wb = g.wbase[m];

  v4d acc[4][4];
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};

  const int arow = tid >> 2, ac2 = tid & 3;
  const bool aok = (k0 + arow) < nkpad;
  d2 ra0, ra1, rn[2], rs[2];
#define LEGDIR_LOAD(s_)                                                                         \
  {                                                                                             \
    ra0 = aok ? *(const d2 *)(PS + (long long)(k0 + arow) * ld + 8 * (s_) + 2 * ac2) : mk2(0, 0); \
    ra1 = aok ? *(const d2 *)(PA + (long long)(k0 + arow) * ld + 8 * (s_) + 2 * ac2) : mk2(0, 0); \
    for (int i = 0; i < 2; i++) {                                                               \
      int idx = tid + 256 * i, brow = idx >> 6, bc2 = idx & 63;                                 \
      int j = 8 * (s_) + brow;                                                                  \
      if (j < ndglu) {                                                                          \
        int latn = isl0 + j, lats = g.ndgl - 1 - latn;                                          \
        rn[i] = *(const d2 *)(FB + ((long long)g.fbase[latn] + m) * ldf + col0 + 2 * bc2);      \
        rs[i] = *(const d2 *)(FB + ((long long)g.fbase[lats] + m) * ldf + col0 + 2 * bc2);      \
      } else {                                                                                  \
        rn[i] = mk2(0, 0);                                                                      \
        rs[i] = mk2(0, 0);                                                                      \
      }                                                                                         \
    }                                                                                           \
  }
  LEGDIR_LOAD(0);
  for (int s = 0; s < nst; s++) {
    if (s > 0) EMI_SYNC();
    // transpose the P tile: As[par][kk = latitude in stage][k index]
    As[(0 * 8 + 2 * ac2) * LG_LDA + arow] = ra0.x;
    As[(0 * 8 + 2 * ac2 + 1) * LG_LDA + arow] = ra0.y;
    As[(1 * 8 + 2 * ac2) * LG_LDA + arow] = ra1.x;
    As[(1 * 8 + 2 * ac2 + 1) * LG_LDA + arow] = ra1.y;
    for (int i = 0; i < 2; i++) {
      int idx = tid + 256 * i, brow = idx >> 6, bc2 = idx & 63;
      *(d2 *)(Bs + (0 * 8 + brow) * LG_LDB + 2 * bc2) = cadd(rn[i], rs[i]);  // symmetric part
      *(d2 *)(Bs + (1 * 8 + brow) * LG_LDB + 2 * bc2) = csub(rn[i], rs[i]);  // antisymmetric part
    }
    EMI_SYNC();
    if (s + 1 < nst) LEGDIR_LOAD(s + 1);
    for (int ks = 0; ks < 2; ks++) {
      const int kk = 4 * ks + (l >> 4);
      double a[4], b[4];
      for (int i = 0; i < 4; i++) a[i] = As[(par * 8 + kk) * LG_LDA + i * 16 + (l & 15)];
      for (int j = 0; j < 4; j++) b[j] = Bs[(par * 8 + kk) * LG_LDB + wn * 64 + j * 16 + (l & 15)];
      for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) acc[i][j] = emi_mfma_f64_16x16x4(a[i], b[j], acc[i][j]);
    }
  }
#undef LEGDIR_LOAD
  for (int i = 0; i < 4; i++)
    for (int q = 0; q < 4; q++) {
      int k = k0 + i * 16 + (l >> 4) + 4 * q;
      if (k < nkpad) {
        double *pw = W + (wb + 2 * k + par) * ldw + col0 + wn * 64 + (l & 15);
        for (int jn = 0; jn < 4; jn++) pw[jn * 16] = acc[i][jn][q];
      }
    }
}

// ==========================================================================================
// FFT engine in LDS: in-place mixed-radix Cooley-Tukey on `nfl` fields of S complex points.
//   DIT: input at perm[] positions -> natural output.  DIF: natural input -> output at perm[].
//   tw[k] = exp(-2 pi i k/S); sgn=+1 conjugates.  Radices 2,4 specialised; 3,5,7,11,13 via the
//   generic small-prime butterfly (DFT matrix read from tw).
// ==========================================================================================
#define FFT_THREADS 256
#define FFT_MAXR 13

EMI_DEVFN d2 tw_get(const d2 *tw, int idx, int sgn) {
  d2 t = tw[idx];
  if (sgn > 0) t.y = -t.y;
  return t;
}

template <int R>
EMI_DEVFN void butterfly(d2 *v, const d2 *tw, int S, int sgn) {
  if (R == 2) {
    d2 a = v[0], b = v[1];
    v[0] = cadd(a, b);
    v[1] = csub(a, b);
  } else if (R == 4) {
    d2 a = cadd(v[0], v[2]), b = csub(v[0], v[2]), c = cadd(v[1], v[3]), d = csub(v[1], v[3]);
    // forward (sgn<0): W_4 = -i
    d2 di = (sgn < 0) ? mk2(d.y, -d.x) : mk2(-d.y, d.x);
    v[0] = cadd(a, c);
    v[1] = cadd(b, di);
    v[2] = csub(a, c);
    v[3] = csub(b, di);
  } else {
    d2 y[R];
    const int st = S / R;
    for (int u = 0; u < R; u++) {
      d2 s = v[0];
      for (int t = 1; t < R; t++) s = cadd(s, cmul(v[t], tw_get(tw, ((u * t) % R) * st, sgn)));
      y[u] = s;
    }
    for (int u = 0; u < R; u++) v[u] = y[u];
  }
}

template <int R, int DIF>
EMI_DEVFN void fft_pass(d2 *a, int nfl, int S, int lenp, const d2 *tw, int sgn) {
  const int len = lenp * R, nb = S / R, tst = S / len;
  for (int idx = EMI_TID; idx < nfl * nb; idx += FFT_THREADS) {
    int fld = idx / nb, q = idx - fld * nb;
    int blk = q / lenp, j = q - blk * lenp;
    d2 *p = a + (long long)fld * S + blk * len + j;
    d2 v[R];
    for (int t = 0; t < R; t++) v[t] = p[t * lenp];
    if (!DIF && j > 0)
      for (int t = 1; t < R; t++) v[t] = cmul(v[t], tw_get(tw, j * t * tst, sgn));
    butterfly<R>(v, tw, S, sgn);
    if (DIF && j > 0)
      for (int t = 1; t < R; t++) v[t] = cmul(v[t], tw_get(tw, j * t * tst, sgn));
    for (int t = 0; t < R; t++) p[t * lenp] = v[t];
  }
}

template <int DIF>
EMI_DEVFN void fft_run(d2 *a, int nfl, int S, const FftPlanDev &pl, const d2 *tw, int sgn) {
  int lenp = DIF ? S : 1;
  for (int ip = 0; ip < pl.nfac; ip++) {
    const int r = DIF ? pl.fac[pl.nfac - 1 - ip] : pl.fac[ip];
    if (DIF) lenp /= r;
    switch (r) {
      case 2: fft_pass<2, DIF>(a, nfl, S, lenp, tw, sgn); break;
      case 3: fft_pass<3, DIF>(a, nfl, S, lenp, tw, sgn); break;
      case 4: fft_pass<4, DIF>(a, nfl, S, lenp, tw, sgn); break;
      case 5: fft_pass<5, DIF>(a, nfl, S, lenp, tw, sgn); break;
      case 7: fft_pass<7, DIF>(a, nfl, S, lenp, tw, sgn); break;
      case 11: fft_pass<11, DIF>(a, nfl, S, lenp, tw, sgn); break;
      case 13: fft_pass<13, DIF>(a, nfl, S, lenp, tw, sgn); break;
      default: break;
    }
    if (!DIF) lenp *= r;
    EMI_SYNC();
  }
}

// Bluestein middle part: a (natural, zero padded to L) -> circular convolution with the chirp
// filter -> natural.  conj_b selects the inverse-transform filter.
EMI_DEVFN void blue_conv(d2 *a, int nfl, const FftPlanDev &pl, const FftTabDev &T, int conj_b) {
  const int L = pl.S;
  const d2 *tw = T.tw + pl.tw_off, *bh = T.bhat + pl.bhat_off;
  fft_run<1>(a, nfl, L, pl, tw, -1);
  for (int idx = EMI_TID; idx < nfl * L; idx += FFT_THREADS) {
    int pos = idx % L;
    d2 b = bh[pos];
    a[idx] = conj_b ? cmulc(a[idx], b) : cmul(a[idx], b);
  }
  EMI_SYNC();
  fft_run<0>(a, nfl, L, pl, tw, +1);
}

EMI_DEVFN long long grid_index(const GridFld &gf, long long p, int nproma) {
  long long blk = p / nproma;
  return (blk * gf.nf_arr + gf.fidx) * (long long)nproma + (p - blk * nproma);
}

// block -> (latitude, field chunk) through a per-class prefix table
struct FftLaunchDev {
  const int *lats;      // latitudes of this LDS class
  const int *blk_pref;  // [nlat_class+1] prefix of chunks per latitude
  int nlat;
};
EMI_DEVFN int fft_find(const int *pref, int n, int b) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (pref[mid] <= b)
      lo = mid;
    else
      hi = mid - 1;
  }
  return lo;
}

// ==========================================================================================
// k_fft_inv: FOURIER_IN (fourier_in_mod.F90:64-76) + FSC (fsc_mod.F90:138-187) + FTINV
// (ftinv_mod.F90:65-84; FFTW c2r semantics, unnormalised) + TRLTOG local copy.
// ==========================================================================================
EMI_KERNEL_LB(256) void k_fft_inv(EmiGeomDev g, FftTabDev T, FftLaunchDev Lc, const GridFld *flds, int nfld, const double *FB,
                          int ldf, int nproma) {
  EMI_LDS_DECL;
  d2 *a = (d2 *)EMI_LDS_PTR;
  const int li = fft_find(Lc.blk_pref, Lc.nlat, EMI_BID);
  const int lat = Lc.lats[li];
  const FftPlanDev pl = T.plans[T.planid[lat]];
  const int f0 = (EMI_BID - Lc.blk_pref[li]) * pl.fbk;
  const int nfl = (nfld - f0) < pl.fbk ? (nfld - f0) : pl.fbk;
  const int n = pl.n, sz = pl.sz, S = pl.S, nmen = g.nmen[lat];
  const double racthe = g.racthe[lat];
  const long long frow = g.fbase[lat];
  const d2 *tw = T.tw + pl.tw_off;
  const unsigned short *perm = T.perm + pl.perm_off;
  const d2 *rtw = T.rtw + pl.rtw_off;
  const d2 *chirp = T.chirp + pl.chirp_off;

  if (pl.blue) {
    for (int idx = EMI_TID; idx < nfl * S; idx += FFT_THREADS) a[idx] = mk2(0, 0);
    EMI_SYNC();
  }
  // ---- stage 1: logical input Z_k, k in [0,sz)
  // field is the fastest index of the work split so that FB reads are contiguous in f
#define LOADX(k_, fl_, out_)                                                        \
  {                                                                                 \
    out_ = mk2(0, 0);                                                               \
    if ((k_) <= nmen) {                                                             \
      GridFld gf_ = flds[f0 + (fl_)];                                               \
      d2 x_ = *(const d2 *)(FB + (frow + (k_)) * ldf + 2 * gf_.src);                \
      if (gf_.mode == GM_ACOS)                                                      \
        x_ = cscale(x_, racthe);                                                    \
      else if (gf_.mode == GM_EWDER)                                                \
        x_ = cscale(cmuli(x_), racthe * (double)(k_));                              \
      else if (gf_.mode == GM_EWDER_UV)                                             \
        x_ = cscale(cmuli(x_), racthe * racthe * (double)(k_));                     \
      out_ = x_;                                                                    \
    }                                                                               \
  }
  if (!pl.cmode) {
    const int npair = sz / 2 + 1;  // k = 0..sz/2 pairs with sz-k
    for (int idx = EMI_TID; idx < nfl * npair; idx += FFT_THREADS) {
      int k = idx / nfl, fl = idx - k * nfl;
      int k2 = sz - k;
      d2 xa, xb;
      LOADX(k, fl, xa);
      LOADX(k2, fl, xb);  // k=0 -> index sz (Nyquist) -> zero since nmen < sz
      // Z_k = (X_k + conj X_{sz-k}) + i w^k (X_k - conj X_{sz-k}),  w = exp(+2 pi i/n)
      d2 wk = cconj(rtw[k]);
      d2 s1 = cadd(xa, cconj(xb)), d1 = csub(xa, cconj(xb));
      d2 zk = cadd(s1, cmuli(cmul(wk, d1)));
      d2 *af = a + (long long)fl * S;
      if (pl.blue)
        af[k] = cmulc(zk, chirp[k]);  // inverse chirp = conj
      else
        af[perm[k]] = zk;
      if (k2 != k && k2 < sz) {
        // Z_{sz-k}: w^{sz-k} = -conj(w^k)
        d2 s2 = cadd(xb, cconj(xa)), d2_ = csub(xb, cconj(xa));
        d2 wk2 = mk2(-wk.x, wk.y);
        d2 zk2 = cadd(s2, cmuli(cmul(wk2, d2_)));
        if (pl.blue)
          af[k2] = cmulc(zk2, chirp[k2]);
        else
          af[perm[k2]] = zk2;
      }
    }
  } else {
    // complex mode (odd n): Z_k = X_k, Z_{n-k} = conj X_k
    for (int idx = EMI_TID; idx < nfl * sz; idx += FFT_THREADS) {
      int k = idx / nfl, fl = idx - k * nfl;
      d2 z;
      if (2 * k <= n) {
        LOADX(k, fl, z);
        if (k == 0) z.y = 0.0;
      } else {
        LOADX(n - k, fl, z);
        z = cconj(z);
      }
      d2 *af = a + (long long)fl * S;
      if (pl.blue)
        af[k] = cmulc(z, chirp[k]);
      else
        af[perm[k]] = z;
    }
  }
#undef LOADX
  EMI_SYNC();
  // ---- stage 2
  if (pl.blue)
    blue_conv(a, nfl, pl, T, 1);
  else
    fft_run<0>(a, nfl, S, pl, tw, +1);
  // ---- stage 3: write the row (TRLTOG local copy)
  const long long gp0 = g.gpoff[lat];
  const double invL = pl.blue ? 1.0 / (double)S : 1.0;
  for (int idx = EMI_TID; idx < nfl * n; idx += FFT_THREADS) {
    int fl = idx / n, p = idx - fl * n;
    const d2 *af = a + (long long)fl * S;
    double v;
    if (!pl.cmode) {
      int lz = p >> 1;
      d2 z = af[lz];
      if (pl.blue) z = cscale(cmulc(z, chirp[lz]), invL);
      v = (p & 1) ? z.y : z.x;
    } else {
      d2 z = af[p];
      if (pl.blue) z = cscale(cmulc(z, chirp[p]), invL);
      v = z.x;
    }
    GridFld gf = flds[f0 + fl];
    gf.base[grid_index(gf, gp0 + p, nproma)] = v;
  }
}

// ==========================================================================================
// k_fft_dir: TRGTOL local copy + FTDIR (ftdir_mod.F90:67-84; r2c, scaled 1/NLOEN at
// tpm_fftw.F90:317-321) + FOURIER_OUT (fourier_out_mod.F90:64-76).  The Gaussian weight
// (ledir_mod.F90:118-124) and LDFOU2's 1/(a cos) (ldfou2_mod.F90:90-96) only depend on the
// latitude and are folded into the same scale factor.
// ==========================================================================================
EMI_KERNEL_LB(256) void k_fft_dir(EmiGeomDev g, FftTabDev T, FftLaunchDev Lc, const GridFld *flds, int nfld, double *FB, int ldf,
                          int nproma) {
  EMI_LDS_DECL;
  d2 *a = (d2 *)EMI_LDS_PTR;
  const int li = fft_find(Lc.blk_pref, Lc.nlat, EMI_BID);
  const int lat = Lc.lats[li];
  const FftPlanDev pl = T.plans[T.planid[lat]];
  const int f0 = (EMI_BID - Lc.blk_pref[li]) * pl.fbk;
  const int nfl = (nfld - f0) < pl.fbk ? (nfld - f0) : pl.fbk;
  const int n = pl.n, sz = pl.sz, S = pl.S, nmen = g.nmen[lat];
  const long long frow = g.fbase[lat];
  const long long gp0 = g.gpoff[lat];
  const d2 *tw = T.tw + pl.tw_off;
  const unsigned short *perm = T.perm + pl.perm_off;
  const d2 *rtw = T.rtw + pl.rtw_off;
  const d2 *chirp = T.chirp + pl.chirp_off;

  if (pl.blue) {
    for (int idx = EMI_TID; idx < nfl * S; idx += FFT_THREADS) a[idx] = mk2(0, 0);
    EMI_SYNC();
  }
  // ---- stage 1: z_l = x_{2l} + i x_{2l+1} (or x_l in complex mode)
  for (int idx = EMI_TID; idx < nfl * sz; idx += FFT_THREADS) {
    int fl = idx / sz, lz = idx - fl * sz;
    GridFld gf = flds[f0 + fl];
    d2 z;
    if (!pl.cmode) {
      z.x = gf.base[grid_index(gf, gp0 + 2 * lz, nproma)];
      z.y = gf.base[grid_index(gf, gp0 + 2 * lz + 1, nproma)];
    } else {
      z.x = gf.base[grid_index(gf, gp0 + lz, nproma)];
      z.y = 0.0;
    }
    d2 *af = a + (long long)fl * S;
    if (pl.blue)
      af[lz] = cmul(z, chirp[lz]);
    else
      af[perm[lz]] = z;
  }
  EMI_SYNC();
  if (pl.blue)
    blue_conv(a, nfl, pl, T, 0);
  else
    fft_run<0>(a, nfl, S, pl, tw, -1);
  // ---- stage 3: X_k, k = 0..NMEN
  const double invL = pl.blue ? 1.0 / (double)S : 1.0;
  const double base_scale = g.rw[lat] / (double)n;
  for (int idx = EMI_TID; idx < nfl * (nmen + 1); idx += FFT_THREADS) {
    int k = idx / nfl, fl = idx - k * nfl;
    const d2 *af = a + (long long)fl * S;
    d2 x;
    if (!pl.cmode) {
      int ka = (k == sz) ? 0 : k, kb = (k == 0) ? 0 : sz - k;
      d2 za = af[ka], zb = af[kb];
      if (pl.blue) {
        za = cscale(cmul(za, chirp[ka]), invL);
        zb = cscale(cmul(zb, chirp[kb]), invL);
      }
      // X_k = 1/2 [ (Z_k + conj Z_{sz-k}) - i exp(-2 pi i k/n) (Z_k - conj Z_{sz-k}) ]
      d2 s1 = cadd(za, cconj(zb)), d1 = csub(za, cconj(zb));
      d2 t = cmuli(cmul(rtw[k], d1));
      x = mk2(0.5 * (s1.x - t.x), 0.5 * (s1.y - t.y));
    } else {
      x = af[k];
      if (pl.blue) x = cscale(cmul(x, chirp[k]), invL);
    }
    GridFld gf = flds[f0 + fl];
    double sc = base_scale * ((gf.mode == GM_ACOS) ? g.racthe[lat] : 1.0);
    *(d2 *)(FB + (frow + k) * ldf + 2 * (f0 + fl)) = cscale(x, sc);
  }
}

// ==========================================================================================
// k_specnorm: SPNORMD (spnormd_mod.F90:40-57).  One block per field; deterministic order.
// ==========================================================================================
EMI_KERNEL_LB(256) void k_specnorm(EmiGeomDev g, const double *sp, int stride, double *out) {
  EMI_LDS_DECL;
  double *red = (double *)EMI_LDS_PTR;
  const int f = EMI_BID, N = g.nsmax;
  const long long nspec2 = (long long)(N + 1) * (N + 2);
  double s = 0.0;
  for (long long i = EMI_TID; i < nspec2; i += EMI_NTHREADS) {
    double v = sp[i * stride + f];
    // m = 0 block is the first 2(N+1) entries: real parts only, weight 1; others weight 2
    if (i < 2LL * (N + 1))
      s += (i & 1) ? 0.0 : v * v;
    else
      s += 2.0 * v * v;
  }
  red[EMI_TID] = s;
  EMI_SYNC();
  for (int st = EMI_NTHREADS / 2; st > 0; st >>= 1) {
    if (EMI_TID < st) red[EMI_TID] += red[EMI_TID + st];
    EMI_SYNC();
  }
  if (EMI_TID == 0) out[f] = sqrt(red[0]);
}
