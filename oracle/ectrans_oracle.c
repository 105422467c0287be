/*
 * ectrans_oracle.c -- TEST INFRASTRUCTURE ONLY (see ectrans_oracle.h).
 *
 * Plain-C restatement of the ecTrans 1.7.0 CPU path for ONE MPI task (LDMPOFF=.TRUE.).
 * Every routine cites the reference file:line it follows (paths relative to
 * /root/reference/src/trans).  Index conventions are kept 1-based/"as in Fortran" through
 * small macros so the code can be read side by side with the reference.
 *
 * Parity pin: reference golden vectors tests/test_ectrans4py/data (tl149) -- 1e-10 abs.
 */
#include "ectrans_oracle.h"

#include <complex.h>
#include <omp.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef double complex cplx;
#define ORC_NB 8 /* columns per block of the restated DGEMMs (LEINV, LEDIR) */

/* ------------------------------------------------------------------------------------ */
/* state (tpm_dim.F90:22-47, tpm_geometry.F90:21-35, tpm_fields.F90:20-37, tpm_flt.F90)  */
/* ------------------------------------------------------------------------------------ */
struct orc_trans {
  int nsmax, ntmax, ndgl, ndgnh, ndlon, lreduced;
  int nspec2, ngptot;
  int nlei1, nlei3, nled3, nled4;
  double ra;
  int *nloen;   /* [ndgl] */
  int *nmen;    /* [ndgl] */
  int *ndglu;   /* [0..nsmax] */
  int *nasm0;   /* [0..nsmax] 1-based */
  int *npmt;    /* [0..nsmax] offset of m in REPSNM (0-based start of JN=m) */
  int *gpoff;   /* [ndgl] first grid point (0-based) of each latitude */
  double *rmu, *rw, *r1mu2, *racthe; /* [ndgl] */
  double *repsnm;                    /* sum_m (ntmax+3-m) */
  double *rn;                        /* [-1..ntmax+3] stored with +1 shift */
  double *rlapin;                    /* [-1..nsmax+2] stored with +1 shift */
  double **rpnma, **rpnms;           /* per m, column-major (idglu x ila|ils) */
  int lazy;                          /* panels are rebuilt per wavenumber inside LTINV / LTDIR (orc_setup_lazy) */
  int sp_mode;                       /* LEDIR as libtrans_sp computes it (orc_set_sp_mode) */
  void *pol;                         /* pol_t of INI_POL, kept for the lazy mode */
};

static void *xcalloc(size_t n, size_t s) {
  void *p = calloc(n ? n : 1, s);
  if (!p) {
    fprintf(stderr, "oracle: out of memory (%zu x %zu)\n", n, s);
    abort();
  }
  return p;
}

/* ------------------------------------------------------------------------------------ */
/* FFTW 3.3.10 semantics (third-party, absent from /root/reference):                     */
/*   r2c: Y_k = sum_j x_j exp(-2 pi i jk/n), k=0..n/2    (dfftw_plan_many_dft_r2c)       */
/*   c2r: x_j = sum_k Y_k exp(+2 pi i jk/n) with Hermitian completion, unnormalised      */
/* Implemented as a textbook recursive mixed-radix DIT + Bluestein for large primes.     */
/* ------------------------------------------------------------------------------------ */
#define ORC_PMAX 37

static int smallest_factor(int n) {
  if (n % 2 == 0) return 2;
  for (int p = 3; (long)p * p <= n; p += 2)
    if (n % p == 0) return p;
  return n;
}

static void make_twiddles(int n, int sign, cplx *w) {
  const double tpi = 2.0 * M_PI;
  for (int k = 0; k < n; k++) {
    double a = tpi * (double)k / (double)n;
    w[k] = cos(a) + I * (sign * sin(a));
  }
}

static void cfft_pow2(int n, cplx *x, const cplx *w /* n twiddles of sign */) {
  /* iterative radix-2, bit reversal first */
  for (int i = 1, j = 0; i < n; i++) {
    int bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) {
      cplx t = x[i];
      x[i] = x[j];
      x[j] = t;
    }
  }
  for (int len = 2; len <= n; len <<= 1) {
    int half = len >> 1, step = n / len;
    for (int i = 0; i < n; i += len)
      for (int k = 0; k < half; k++) {
        cplx u = x[i + k], v = x[i + k + half] * w[k * step];
        x[i + k] = u + v;
        x[i + k + half] = u - v;
      }
  }
}

/* Tables of one Bluestein length and sign (twiddles of the power-of-two work length, chirp, filter
 * spectrum): they depend on (n, sign) only, so a plan computes them once instead of once per transform. */
typedef struct {
  int n, sign, m;
  cplx *wf, *wb, *c, *b;
} bluetab;
#define ORC_NBLUE 4
typedef struct {
  bluetab tab[ORC_NBLUE];
  int ntab;
} bluecache;
static _Thread_local bluecache *cur_blue = NULL; /* set by r2c_plan / c2r_plan around cfft_rec */

static void bluetab_make(bluetab *t, int n, int sign) {
  int m = 1;
  while (m < 2 * n - 1) m <<= 1;
  t->n = n, t->sign = sign, t->m = m;
  t->c = xcalloc(n, sizeof(cplx));
  t->b = xcalloc(m, sizeof(cplx));
  t->wf = xcalloc(m, sizeof(cplx)), t->wb = xcalloc(m, sizeof(cplx));
  make_twiddles(m, -1, t->wf);
  make_twiddles(m, +1, t->wb);
  for (int j = 0; j < n; j++) {
    long j2 = ((long)j * j) % (2L * n);
    double ang = M_PI * (double)j2 / (double)n;
    t->c[j] = cos(ang) + I * (sign * sin(ang)); /* exp(sign i pi j^2/n) */
  }
  t->b[0] = conj(t->c[0]);
  for (int j = 1; j < n; j++) t->b[j] = t->b[m - j] = conj(t->c[j]);
  cfft_pow2(m, t->b, t->wf);
}
static void bluetab_free(bluetab *t) { free(t->c), free(t->b), free(t->wf), free(t->wb); }

static void bluestein(int n, int sign, const cplx *in, int istride, cplx *out) {
  bluetab local, *t = NULL;
  int own = 0;
  if (cur_blue) {
    for (int i = 0; i < cur_blue->ntab; i++)
      if (cur_blue->tab[i].n == n && cur_blue->tab[i].sign == sign) t = &cur_blue->tab[i];
    if (!t && cur_blue->ntab < ORC_NBLUE) {
      t = &cur_blue->tab[cur_blue->ntab++];
      bluetab_make(t, n, sign);
    }
  }
  if (!t) {
    bluetab_make(&local, n, sign);
    t = &local;
    own = 1;
  }
  const int m = t->m;
  const cplx *c = t->c, *b = t->b;
  cplx *a = xcalloc(m, sizeof(cplx));
  for (int j = 0; j < n; j++) a[j] = in[(size_t)j * istride] * c[j];
  cfft_pow2(m, a, t->wf);
  for (int k = 0; k < m; k++) a[k] *= b[k];
  cfft_pow2(m, a, t->wb);
  for (int k = 0; k < n; k++) out[k] = a[k] * c[k] / (double)m;
  free(a);
  if (own) bluetab_free(&local);
}

/* out[0..n) = DFT_sign(in[0], in[stride], ...); wtop: table of size ntop (n | ntop) */
static void cfft_rec(int n, int sign, const cplx *in, int istride, cplx *out, const cplx *wtop,
                     int ntop) {
  if (n == 1) {
    out[0] = in[0];
    return;
  }
  int p = smallest_factor(n);
  if (p > ORC_PMAX) {
    bluestein(n, sign, in, istride, out);
    return;
  }
  int m = n / p;
  for (int q = 0; q < p; q++) cfft_rec(m, sign, in + (size_t)q * istride, istride * p, out + (size_t)q * m, wtop, ntop);
  int tstep = ntop / n, pstep = ntop / p;
  cplx t[ORC_PMAX], y[ORC_PMAX];
  for (int k = 0; k < m; k++) {
    for (int q = 0; q < p; q++) t[q] = out[(size_t)q * m + k] * wtop[((long)q * k % n) * tstep];
    for (int r = 0; r < p; r++) {
      cplx s = t[0];
      for (int q = 1; q < p; q++) s += t[q] * wtop[((long)q * r % p) * pstep];
      y[r] = s;
    }
    for (int r = 0; r < p; r++) out[(size_t)r * m + k] = y[r];
  }
}

typedef struct {
  int n;
  cplx *wf, *wb;
  bluecache blue;
} fftplan;

static fftplan fftplan_make(int n) {
  fftplan p;
  p.n = n;
  p.wf = xcalloc(n, sizeof(cplx));
  p.wb = xcalloc(n, sizeof(cplx));
  make_twiddles(n, -1, p.wf);
  make_twiddles(n, +1, p.wb);
  p.blue.ntab = 0;
  return p;
}
static void fftplan_free(fftplan *p) {
  free(p->wf);
  free(p->wb);
  for (int i = 0; i < p->blue.ntab; i++) bluetab_free(&p->blue.tab[i]);
  p->blue.ntab = 0;
}

static void r2c_plan(fftplan *p, const double *in, cplx *work /* 2n */, double *out) {
  int n = p->n;
  cplx *a = work, *b = work + n;
  for (int j = 0; j < n; j++) a[j] = in[j];
  cur_blue = &p->blue;
  cfft_rec(n, -1, a, 1, b, p->wf, n);
  cur_blue = NULL;
  for (int k = 0; k <= n / 2; k++) {
    out[2 * k] = creal(b[k]);
    out[2 * k + 1] = cimag(b[k]);
  }
}

static void c2r_plan(fftplan *p, const double *in, cplx *work /* 2n */, double *out) {
  int n = p->n;
  cplx *a = work, *b = work + n;
  /* Hermitian completion; imaginary parts of k=0 (and k=n/2 for even n) are ignored, as
   * FFTW's c2r does. */
  a[0] = in[0];
  for (int k = 1; k <= n / 2; k++) {
    cplx v = in[2 * k] + I * in[2 * k + 1];
    if (2 * k == n) {
      a[k] = creal(v);
    } else {
      a[k] = v;
      a[n - k] = conj(v);
    }
  }
  cur_blue = &p->blue;
  cfft_rec(n, +1, a, 1, b, p->wb, n);
  cur_blue = NULL;
  for (int j = 0; j < n; j++) out[j] = creal(b[j]);
}

void orc_fft_r2c(int n, const double *in, double *out) {
  fftplan p = fftplan_make(n);
  cplx *w = xcalloc(2 * (size_t)n, sizeof(cplx));
  r2c_plan(&p, in, w, out);
  free(w);
  fftplan_free(&p);
}
void orc_fft_c2r(int n, const double *in, double *out) {
  fftplan p = fftplan_make(n);
  cplx *w = xcalloc(2 * (size_t)n, sizeof(cplx));
  c2r_plan(&p, in, w, out);
  free(w);
  fftplan_free(&p);
}

/* ------------------------------------------------------------------------------------ */
/* INI_POL (common/internal/tpm_pol.F90:31-99): coefficient tables for SUPOL/SUPOLF      */
/* ------------------------------------------------------------------------------------ */
typedef struct {
  int nmax;
  double *dfa, *dfb, *dff, *dfg, *dfi; /* [0..nmax] */
  double *dda, *ddi, *ddh;             /* [0..nmax] */
} pol_t;

static pol_t ini_pol(int knsmax) {
  pol_t p;
  p.nmax = knsmax;
  size_t n = (size_t)knsmax + 1;
  p.dfa = xcalloc(n, 8), p.dfb = xcalloc(n, 8), p.dff = xcalloc(n, 8), p.dfg = xcalloc(n, 8);
  p.dfi = xcalloc(n, 8), p.dda = xcalloc(n, 8), p.ddi = xcalloc(n, 8), p.ddh = xcalloc(n, 8);
  for (int jn = 1; jn <= knsmax; jn++) { /* tpm_pol.F90:73-80, 92-96 */
    p.dfa[jn] = 1.0 / sqrt((double)(jn * (jn + 1)));
    p.dfb[jn] = sqrt((double)(2 * jn + 1) / (double)(jn * (jn + 1)));
    p.dff[jn] = (double)(2 * jn - 1) / (double)jn;
    p.dfg[jn] = (double)(jn - 1) / (double)jn;
    p.dfi[jn] = (double)jn;
    p.dda[jn] = 1.0 / sqrt((double)(jn * (jn + 1)));
    p.ddi[jn] = (double)jn;
    p.ddh[jn] = sqrt((double)(2 * jn + 1) / (double)(2 * jn));
  }
  return p;
}
static void end_pol(pol_t *p) {
  free(p->dfa), free(p->dfb), free(p->dff), free(p->dfg), free(p->dfi);
  free(p->dda), free(p->ddi), free(p->ddh);
}

/* statement functions DC/DD/DE of tpm_pol.F90:43-52 (Belousov coefficients) */
static double pol_dc(int n, int m) {
  return sqrt(((double)(2 * n + 1) * (double)(n + m - 1) * (double)(n + m - 3)) /
              ((double)(2 * n - 3) * (double)(n + m) * (double)(n + m - 2)));
}
static double pol_dd(int n, int m) {
  return sqrt(((double)(2 * n + 1) * (double)(n + m - 1) * (double)(n - m + 1)) /
              ((double)(2 * n - 1) * (double)(n + m) * (double)(n + m - 2)));
}
static double pol_de(int n, int m) {
  return sqrt(((double)(2 * n + 1) * (double)(n - m)) / ((double)(2 * n - 1) * (double)(n + m)));
}

/* ------------------------------------------------------------------------------------ */
/* SUPOLF (common/internal/supolf_mod.F90:13-251): P_n^m(mu), n=0..knsmax for fixed m   */
/* kcheap: 1 all, 2 only even n-m, 3 only odd n-m (supolf_mod.F90:92-99, 209-219)       */
/* ------------------------------------------------------------------------------------ */
static double supolf_dcl(int kkl, int km) { /* supolf_mod.F90:79-81 */
  return sqrt(((double)(kkl - km + 1) * (double)(kkl - km + 2) * (double)(kkl + km + 1) *
               (double)(kkl + km + 2)) /
              ((double)(2 * kkl + 1) * (double)(2 * kkl + 3) * (double)(2 * kkl + 3) *
               (double)(2 * kkl + 5)));
}
static double supolf_ddl(int kkl, int km) { /* supolf_mod.F90:82-83 */
  return (2.0 * (double)kkl * (double)(kkl + 1) - 2.0 * (double)(km * km) - 1.0) /
         ((double)(2 * kkl - 1) * (double)(2 * kkl + 3));
}

static void supolf(const pol_t *pol, int km, int knsmax, double ddmu, double *ddpol, int kcheap,
                   int *icorr /* scratch [0..knsmax] */) {
  const double zeps = 2.220446049250313e-16; /* EPSILON(1._JPRD) */
  int icorr3 = 0;
  double dlx = ddmu;
  double zcos2 = 1.0 - dlx * dlx;
  double zcos = sqrt(zcos2), zcos_r;
  if (fabs(zcos) <= zeps) { /* supolf_mod.F90:112-119 */
    dlx = 1.0;
    zcos = 0.0;
    zcos_r = 0.0;
    zcos2 = 0.0;
  } else {
    zcos_r = 1.0 / zcos;
  }
  double dlkm2 = 1.0, dlkm1 = dlx, dlk, dl1;
  if (km == 0) { /* supolf_mod.F90:124-132 */
    ddpol[0] = dlkm2;
    ddpol[1] = dlkm1 * pol->dfb[1] / pol->dfa[1];
    for (int jn = 2; jn <= knsmax; jn++) {
      dlk = pol->dff[jn] * dlx * dlkm1 - pol->dfg[jn] * dlkm2;
      ddpol[jn] = dlk * pol->dfb[jn] / pol->dfa[jn];
      dlkm2 = dlkm1;
      dlkm1 = dlk;
    }
  } else if (km == 1) { /* supolf_mod.F90:133-142 */
    ddpol[0] = 0.0;
    ddpol[1] = zcos * pol->dfb[1];
    for (int jn = 2; jn <= knsmax; jn++) {
      dlk = pol->dff[jn] * dlx * dlkm1 - pol->dfg[jn] * dlkm2;
      dl1 = pol->dfi[jn] * (dlkm1 - dlx * dlk) * zcos_r;
      ddpol[jn] = dl1 * pol->dfb[jn];
      dlkm2 = dlkm1;
      dlkm1 = dlk;
    }
  } else { /* supolf_mod.F90:143-246 */
    const double zscale = 1.0e+100, ziscale = 1.0e-100;
    double zlsita = 1.0;
    for (int jn = 1; jn <= km / 2; jn++) {
      zlsita *= zcos2;
      if (fabs(zlsita) < ziscale) {
        zlsita *= zscale;
        icorr3++;
      }
    }
    if (km % 2 == 1) zlsita *= zcos;
    double zfac = 1.0;
    for (int jn = 1; jn <= km - 1; jn++) {
      zfac *= sqrt((double)(2 * jn - 1));
      zfac /= sqrt((double)(2 * jn));
    }
    zfac *= sqrt((double)(2 * km - 1));
    double zfac0 = 1.0, zfac1 = 1.0, zmult = 0.0;
    int icmax = knsmax - km < 3 ? knsmax - km : 3;
    for (int ic = 0; ic <= icmax; ic++) {
      zfac0 *= (double)(2 * km + ic);
      switch (ic) {
      case 0:
        zfac1 = 1.0;
        zmult = zfac;
        break;
      case 1:
        zfac1 = 1.0;
        zfac *= (double)(2 * km + 1);
        zmult = zfac * dlx;
        break;
      case 2:
        zfac1 = 2.0;
        zmult = 0.5 * zfac * ((double)(2 * km + 3) * dlx * dlx - 1.0);
        break;
      case 3:
        zfac1 = 6.0;
        zfac *= (double)(2 * km + 3);
        zmult = (1.0 / 6.0) * dlx * zfac * ((double)(2 * km + 5) * dlx * dlx - 3.0);
        break;
      }
      ddpol[km + ic] = zlsita * zmult * sqrt(2.0 * ((double)(km + ic) + 0.5) * zfac1 / zfac0);
    }
    for (int jn = 0; jn <= knsmax; jn++) icorr[jn] = icorr3; /* ICORR(1:KNSMAX) */
    int istart = (kcheap != 3) ? 0 : 1;
    int iinc = (kcheap == 2 || kcheap == 3) ? 2 : 1;
    for (int jn = km + istart + 4; jn <= knsmax; jn += iinc) {
      if (fabs(ddpol[jn - 4]) > zscale) {
        for (int j = jn - 4; j <= jn - 1; j++) ddpol[j] /= zscale;
        for (int j = jn - 4; j <= knsmax; j++) icorr[j] -= 1;
      }
      ddpol[jn] = ((dlx * dlx - supolf_ddl(jn - 2, km)) * ddpol[jn - 2] -
                   supolf_dcl(jn - 4, km) * ddpol[jn - 4]) /
                  supolf_dcl(jn - 2, km);
    }
    for (int jn = km + istart; jn <= knsmax; jn += iinc) {
      for (int jc = 1; jc <= icorr[jn]; jc++) {
        ddpol[jn] /= zscale;
        if (ddpol[jn] < zeps) ddpol[jn] = zeps; /* supolf_mod.F90:241-243 (sic: no ABS) */
      }
    }
  }
}

/* ------------------------------------------------------------------------------------ */
/* SUPOL (common/internal/supol_mod.F90:13-171): Belousov, all (m,n) for one latitude    */
/* pfn: ZFN(0:nmax,0:nmax) row-major [jn][jk]; pol out: [jm][jn] with ld = nmax+1        */
/* ------------------------------------------------------------------------------------ */
static void supol(const pol_t *pol, int knsmax, double pddmu, const double *pfn, double *p) {
  const int ld = knsmax + 1;
#define PFN(jn, jk) pfn[(size_t)(jn) * ld + (jk)]
#define POL(jm, jn) p[(size_t)(jm) * ld + (jn)]
  double zdlx = pddmu;
  double zdlx1 = acos(zdlx);
  double zdlsita = sqrt(1.0 - zdlx * zdlx), zdl1sita;
  POL(0, 0) = 1.0;
  if (fabs(zdlsita) <= sqrt(2.220446049250313e-16)) { /* supol_mod.F90:93-99 */
    zdlx = 1.0;
    zdlsita = 0.0;
    zdl1sita = 0.0;
  } else {
    zdl1sita = 1.0 / zdlsita;
  }
  for (int jn = 2; jn <= knsmax; jn += 2) { /* supol_mod.F90:106-118 */
    double zdlk = 0.5 * PFN(jn, 0), zdlldn = 0.0;
    for (int jk = 2; jk <= jn; jk += 2) {
      zdlk += PFN(jn, jk) * cos(pol->ddi[jk] * zdlx1);
      zdlldn += pol->dda[jn] * PFN(jn, jk) * pol->ddi[jk] * sin(pol->ddi[jk] * zdlx1);
    }
    POL(0, jn) = zdlk;
    POL(1, jn) = zdlldn;
  }
  for (int jn = 1; jn <= knsmax; jn += 2) { /* supol_mod.F90:122-134 */
    double zdlk = 0.0, zdlldn = 0.0;
    for (int jk = 1; jk <= jn; jk += 2) {
      zdlk += PFN(jn, jk) * cos(pol->ddi[jk] * zdlx1);
      zdlldn += pol->dda[jn] * PFN(jn, jk) * pol->ddi[jk] * sin(pol->ddi[jk] * zdlx1);
    }
    POL(0, jn) = zdlk;
    POL(1, jn) = zdlldn;
  }
  double zdls = zdl1sita * 2.2250738585072014e-308; /* TINY */
  for (int jn = 2; jn <= knsmax; jn++) {            /* supol_mod.F90:148-151 */
    POL(jn, jn) = POL(jn - 1, jn - 1) * zdlsita * pol->ddh[jn];
    if (fabs(POL(jn, jn)) < zdls) POL(jn, jn) = 0.0;
  }
  for (int jn = 3; jn <= knsmax; jn++) /* supol_mod.F90:158-166 */
    for (int jm = 2; jm <= jn - 1; jm++)
      POL(jm, jn) = pol_dc(jn, jm) * POL(jm - 2, jn - 2) - pol_dd(jn, jm) * POL(jm - 2, jn - 1) * zdlx +
                    pol_de(jn, jm) * POL(jm, jn - 1) * zdlx;
#undef PFN
#undef POL
}

/* ZFN row jn: Fourier coefficients of ordinary Legendre polynomial (suleg_mod.F90:249-263) */
static void zfn_row(int jn, double *row /* [0..jn] */) {
  double zfnn = 2.0;
  for (int jgl = 1; jgl <= jn; jgl++) zfnn *= sqrt(1.0 - 0.25 / ((double)jgl * (double)jgl));
  int iodd = jn % 2;
  for (int k = 0; k <= jn; k++) row[k] = 0.0;
  row[jn] = zfnn;
  for (int jgl = 2; jgl <= jn - iodd; jgl += 2)
    row[jn - jgl] = row[jn - jgl + 2] * (double)((jgl - 1) * (2 * jn - jgl + 2)) / (double)(jgl * (2 * jn - jgl + 1));
}

/* CPLEDN (common/internal/cpledn_mod.F90:94-129) */
static void cpledn(int kn, int kodd, const double *pfn, double px, int kflag, double *pw, double *pxn,
                   double *pxmod) {
  double zdlx = px, zdlk = 0.0, zdlldn = 0.0;
  if (kodd == 0) zdlk = 0.5 * pfn[0];
  int ik = 1;
  if (kflag == 0) {
    for (int jn = 2 - kodd; jn <= kn; jn += 2) {
      zdlk += pfn[ik] * cos((double)jn * zdlx);
      zdlldn -= pfn[ik] * (double)jn * sin((double)jn * zdlx);
      ik++;
    }
    double zdlmod = -zdlk / zdlldn;
    *pxn = zdlx + zdlmod;
    *pxmod = zdlmod;
  }
  if (kflag == 1) {
    for (int jn = 2 - kodd; jn <= kn; jn += 2) {
      zdlldn -= pfn[ik] * (double)jn * sin((double)jn * zdlx);
      ik++;
    }
    *pw = (double)(2 * kn + 1) / (zdlldn * zdlldn);
  }
}

/* GAWL (common/internal/gawl_mod.F90:84-110) */
static int gawl(const double *pfn, double *pl, double *pw, double peps, int kn) {
  int itemax = 20, iflag = 0, iodd = kn % 2, kiter = 0;
  double zx = *pl, zw = 0.0, zxn = *pl, pmod = 0.0;
  for (int jter = 1; jter <= itemax + 1; jter++) {
    kiter = jter;
    cpledn(kn, iodd, pfn, zx, iflag, &zw, &zxn, &pmod);
    zx = zxn;
    if (iflag == 1) break;
    if (fabs(pmod) <= peps * 1000.0) iflag = 1;
  }
  *pl = zxn;
  *pw = zw;
  return kiter;
}

/* SUGAW, LLOLD branch (common/internal/sugaw_mod.F90:157-190, 363-368) */
static void sugaw(int kdgl, double *pl /* mu */, double *pw) {
  const double zeps = 2.220446049250313e-16;
  int kn = kdgl, ins2 = kdgl / 2, iodd = kdgl % 2;
  double *rowfull = xcalloc((size_t)kdgl + 1, 8);
  double *zfn = xcalloc((size_t)kdgl / 2 + 2, 8);
  double *zl = xcalloc((size_t)kdgl + 1, 8);
  zfn_row(kdgl, rowfull);
  int ik = iodd;
  for (int jgl = iodd; jgl <= kdgl; jgl += 2) zfn[ik++] = rowfull[jgl];
  const double zpi = 2.0 * asin(1.0);
  for (int jgl = 1; jgl <= ins2; jgl++) {
    double z = (double)(4 * jgl - 1) * zpi / (double)(4 * kn + 2);
    zl[jgl] = z + 1.0 / (tan(z) * (double)(8 * kn * kn)); /* KN**2 in int32 as in Fortran */
  }
  int fail = 0;
#pragma omp parallel for schedule(static)
  for (int jgl = ins2; jgl >= 1; jgl--) {
    int it = gawl(zfn, &zl[jgl], &pw[jgl - 1], zeps, kn);
    if (it > 20) fail = 1;
  }
  if (fail) {
    fprintf(stderr, "oracle: FAILURE IN SUGAW\n");
    abort();
  }
  for (int jgl = 1; jgl <= ins2; jgl++) pl[jgl - 1] = cos(zl[jgl]);
  for (int jgl = 1; jgl <= kdgl / 2; jgl++) {
    int isym = kdgl - jgl + 1;
    pl[isym - 1] = -pl[jgl - 1];
    pw[isym - 1] = pw[jgl - 1];
  }
  free(rowfull);
  free(zfn);
  free(zl);
}

/* SETUP_GEOM (common/internal/setup_geom_mod.F90:44-97) */
static void setup_geom(orc_trans *t) {
  int ndgl = t->ndgl, nsmax = t->nsmax, ndgnh = t->ndgnh;
  int nsmaxlin = ndgl - 1;
  int *nmen = t->nmen, *nloen = t->nloen;
#define NLOEN(j) nloen[(j)-1]
#define NMEN(j) nmen[(j)-1]
#define IMIN(a, b) ((a) < (b) ? (a) : (b))
#define IMAX(a, b) ((a) > (b) ? (a) : (b))
  if (nsmax >= nsmaxlin || !t->lreduced) {
    for (int jgl = 1; jgl <= ndgl; jgl++) NMEN(jgl) = IMIN(nsmax, (NLOEN(jgl) - 1) / 2);
  } else if (nsmax >= ndgl * 2 / 3 - 1) {
    int ifac = 3 * (nsmaxlin - nsmax) / ndgl; /* integer arithmetic as in the reference */
#define ZSQM2(j) ((double)ifac * t->r1mu2[(j)-1])
    NMEN(1) = IMIN(nsmax, (int)((double)(NLOEN(1) - 1) / (2.0 + ZSQM2(1))));
    for (int jgl = 2; jgl <= ndgnh; jgl++)
      NMEN(jgl) = IMIN(nsmax, IMAX(NMEN(jgl - 1), (int)((double)(NLOEN(jgl) - 1) / (2.0 + ZSQM2(jgl)))));
    NMEN(ndgl) = IMIN(nsmax, (int)((double)(NLOEN(ndgl) - 1) / (2.0 + ZSQM2(ndgl))));
    for (int jgl = ndgl - 1; jgl >= ndgnh + 1; jgl--)
      NMEN(jgl) = IMIN(nsmax, IMAX(NMEN(jgl + 1), (int)((double)(NLOEN(jgl) - 1) / (2.0 + ZSQM2(jgl)))));
#undef ZSQM2
  } else {
#define ZSQM2(j) (t->r1mu2[(j)-1])
    NMEN(1) = IMIN(nsmax, (int)((double)(NLOEN(1) - 1) / (2.0 + ZSQM2(1))) - 1);
    for (int jgl = 2; jgl <= ndgnh; jgl++)
      NMEN(jgl) = IMIN(nsmax, IMAX(NMEN(jgl - 1), (int)((double)(NLOEN(jgl) - 1) / (2.0 + ZSQM2(jgl))) - 1));
    NMEN(ndgl) = IMIN(nsmax, (int)((double)(NLOEN(ndgl) - 1) / (2.0 + ZSQM2(ndgl))) - 1);
    for (int jgl = ndgl - 1; jgl >= ndgnh + 1; jgl--)
      NMEN(jgl) = IMIN(nsmax, IMAX(NMEN(jgl + 1), (int)((double)(NLOEN(jgl) - 1) / (2.0 + ZSQM2(jgl))) - 1));
#undef ZSQM2
  }
  for (int jm = 0; jm <= nsmax; jm++) t->ndglu[jm] = 0;
  for (int jgl = 1; jgl <= ndgnh; jgl++)
    for (int jm = 0; jm <= NMEN(jgl) && jm <= nsmax; jm++) t->ndglu[jm] += 1;
}

/* RPNMA / RPNMS of one wavenumber by SUPOLF (suleg_mod.F90:609-662, 891-944) */
static void supolf_panels(const orc_trans *t, const pol_t *pol, int im, double **ppa, double **pps) {
  const int nsmax = t->nsmax, imaxn = t->ntmax + 1;
  double *zlpol = xcalloc((size_t)imaxn + 3, 8);
  int *icorr = xcalloc((size_t)imaxn + 3, sizeof(int));
  int ila = (nsmax - im + 2) / 2, ils = (nsmax - im + 3) / 2;
  int idglu = IMIN(t->ndgnh, t->ndglu[im]);
  int isl = IMAX(t->ndgnh - t->ndglu[im] + 1, 1);
  double *pa = xcalloc((size_t)idglu * ila, 8), *ps = xcalloc((size_t)idglu * ils, 8);
  int inmaxa = ((imaxn - im) % 2 == 0) ? imaxn + 1 : imaxn; /* suleg_mod.F90:631-635 */
  int inmaxs = ((imaxn - im) % 2 == 0) ? imaxn : imaxn + 1; /* suleg_mod.F90:913-917 */
  for (int jgl = 1; jgl <= idglu; jgl++) {
    double mu = t->rmu[isl + jgl - 1 - 1];
    supolf(pol, im, inmaxa, mu, zlpol, 3, icorr);
    for (int ji = 1; ji <= ila; ji++) /* column ILA-JI+1 <- n = m+2(ji-1)+1 */
      pa[(size_t)(ila - ji) * idglu + (jgl - 1)] = zlpol[im + 2 * (ji - 1) + 1];
    supolf(pol, im, inmaxs, mu, zlpol, 2, icorr);
    for (int ji = 1; ji <= ils; ji++) ps[(size_t)(ils - ji) * idglu + (jgl - 1)] = zlpol[im + 2 * (ji - 1)];
  }
  free(zlpol);
  free(icorr);
  *ppa = pa;
  *pps = ps;
}

/* The Legendre functions of ONE latitude and wavenumber exactly as the panels hold them (same SUPOLF calls as
 * supolf_panels: suleg_mod.F90:609-662, 891-944): out[n - km] = P_n^km(mu(jgl)), n = km .. nsmax; jgl 1-based, any hemisphere
 * (P_n^m(-mu) = (-1)^(n-m) P_n^m(mu) is NOT applied: the value at the latitude's own mu is returned).  For test yardsticks
 * that evaluate sampled rows of a transform independently (tests/common.py::fp32_yardstick).                          */
void orc_legpol(const orc_trans *t, int km, int jgl, double *out) {
  const int nsmax = t->nsmax, imaxn = t->ntmax + 1;
  pol_t tmp, *pol = (pol_t *)t->pol;
  if (!pol) {
    tmp = ini_pol(t->ntmax + 3);
    pol = &tmp;
  }
  double *zlpol = xcalloc((size_t)imaxn + 3, 8);
  int *icorr = xcalloc((size_t)imaxn + 3, sizeof(int));
  const int inmaxa = ((imaxn - km) % 2 == 0) ? imaxn + 1 : imaxn, inmaxs = ((imaxn - km) % 2 == 0) ? imaxn : imaxn + 1;
  const double mu = t->rmu[jgl - 1];
  supolf(pol, km, inmaxa, mu, zlpol, 3, icorr);
  for (int n = km + 1; n <= nsmax; n += 2) out[n - km] = zlpol[n];
  supolf(pol, km, inmaxs, mu, zlpol, 2, icorr);
  for (int n = km; n <= nsmax; n += 2) out[n - km] = zlpol[n];
  free(zlpol);
  free(icorr);
  if (pol == &tmp) end_pol(&tmp);
}

/* ------------------------------------------------------------------------------------ */
/* SETUP_TRANS (cpu/external/setup_trans.F90:169-428) -> SETUP_DIMS, SUMP_TRANS_PRELEG,  */
/* PRE_SULEG, SULEG, SETUP_GEOM for NPROC=1                                              */
/* ------------------------------------------------------------------------------------ */
static orc_trans *orc_setup_impl(int nsmax, int ndgl, const int *nloen_in, int belusov, double ra, int lazy) {
  if (ndgl <= 0 || ndgl % 2 != 0) return NULL; /* setup_trans.F90:268-270 */
  orc_trans *t = xcalloc(1, sizeof(*t));
  t->nsmax = nsmax;
  t->ntmax = nsmax; /* setup_trans.F90:304-308 */
  t->ndgl = ndgl;
  t->ra = ra;
  t->nloen = xcalloc(ndgl, sizeof(int));
  int ndlon = 2 * ndgl;
  if (nloen_in) {
    ndlon = 0;
    for (int j = 0; j < ndgl; j++) {
      if (nloen_in[j] <= 0) {
        free(t->nloen);
        free(t);
        return NULL;
      }
      if (nloen_in[j] > ndlon) ndlon = nloen_in[j];
    }
    for (int j = 0; j < ndgl; j++)
      if (nloen_in[j] != ndlon) t->lreduced = 1;
  }
  t->ndlon = ndlon;
  for (int j = 0; j < ndgl; j++) t->nloen[j] = t->lreduced ? nloen_in[j] : ndlon;
  /* SETUP_DIMS (setup_dims_mod.F90:29-47) */
  t->nspec2 = (nsmax + 1) * (nsmax + 2);
  t->ndgnh = (ndgl + 1) / 2;
  t->nlei1 = nsmax + 4 + (nsmax + 4 + 1) % 2;
  t->nlei3 = t->ndgnh + (t->ndgnh + 2) % 2;
  t->nled3 = t->ntmax + 2 + (t->ntmax + 3) % 2;
  t->nled4 = t->ntmax + 3 + (t->ntmax + 4) % 2;
  /* SUWAVEDI with one W-set (suwavedi_mod.F90:118-137): NASM0 */
  t->nasm0 = xcalloc((size_t)nsmax + 1, sizeof(int));
  t->npmt = xcalloc((size_t)nsmax + 1, sizeof(int));
  {
    int ipos = 1, inm = 0;
    for (int jm = 0; jm <= nsmax; jm++) {
      t->nasm0[jm] = ipos;
      ipos += (nsmax - jm + 1) * 2;
      t->npmt[jm] = inm; /* start of REPSNM block for m: JN=m..ntmax+2 (pre_suleg_mod.F90:36-42) */
      inm += t->ntmax + 3 - jm;
    }
    t->repsnm = xcalloc((size_t)inm, 8);
  }
  /* PRE_SULEG (pre_suleg_mod.F90:55-69) */
  for (int jm = 0; jm <= nsmax; jm++)
    for (int jn = jm; jn <= t->ntmax + 2; jn++)
      t->repsnm[t->npmt[jm] + jn - jm] = sqrt((double)(jn * jn - jm * jm) / (double)(4 * jn * jn - 1));
  t->rn = xcalloc((size_t)t->ntmax + 5, 8);
  for (int jn = -1; jn <= t->ntmax + 3; jn++) t->rn[jn + 1] = (double)jn;
  t->rlapin = xcalloc((size_t)nsmax + 4, 8);
  for (int jn = 1; jn <= nsmax + 2; jn++) t->rlapin[jn + 1] = -(ra * ra / (double)(jn * (jn + 1)));
  /* SULEG 3.1: Gaussian latitudes and weights (suleg_mod.F90:241-293) */
  t->rmu = xcalloc(ndgl, 8), t->rw = xcalloc(ndgl, 8);
  t->r1mu2 = xcalloc(ndgl, 8), t->racthe = xcalloc(ndgl, 8);
  sugaw(ndgl, t->rmu, t->rw);
  for (int j = 0; j < ndgl; j++) { /* suleg_mod.F90:386-394 */
    double ztheta = asin(t->rmu[j]);
    double zcos = cos(ztheta);
    t->r1mu2[j] = zcos * zcos;
    t->racthe[j] = 1.0 / zcos / ra;
  }
  t->nmen = xcalloc(ndgl, sizeof(int));
  t->ndglu = xcalloc((size_t)nsmax + 1, sizeof(int));
  setup_geom(t);
  /* grid-point offsets: one task owns all latitudes, N->S (sustaonl: NSTA=1, NONL=NLOEN) */
  t->gpoff = xcalloc(ndgl, sizeof(int));
  {
    int off = 0;
    for (int j = 0; j < ndgl; j++) {
      t->gpoff[j] = off;
      off += t->nloen[j];
    }
    t->ngptot = off;
  }
  /* Legendre panels RPNMA/RPNMS (suleg_mod.F90:609-615,721-726 | 891-897,1001-1006) */
  t->rpnma = xcalloc((size_t)nsmax + 1, sizeof(double *));
  t->rpnms = xcalloc((size_t)nsmax + 1, sizeof(double *));
  pol_t pol = ini_pol(t->ntmax + 3); /* suleg_mod.F90:241 */
  if (!belusov) {
    if (lazy) {
      /* panels on demand (supolf_panels called from LTINV / LTDIR): nothing is stored, so a TCo2559 oracle needs
       * megabytes instead of 51 GiB; the values are the same, operation for operation */
      t->lazy = 1;
      pol_t *keep = xcalloc(1, sizeof(pol_t));
      *keep = pol;
      t->pol = keep;
      return t;
    }
#pragma omp parallel for schedule(dynamic, 1)
    for (int im = 0; im <= nsmax; im++) supolf_panels(t, &pol, im, &t->rpnma[im], &t->rpnms[im]);
  } else {
    /* Belousov: RPNM(lat, NPMS(m)+..) from SUPOL per latitude with INSMAX=NTMAX+1
     * (suleg_mod.F90:402-468); panels then cut out at suleg_mod.F90:745-766, 1025-1046:
     * RPNMA(JGL,JI) = RPNM(ISL+JGL-1, NPMS(IM)+IA+(JI-1)*2), the NPMS block of m holding
     * n = INSMAX..m (descending).                                                       */
    int insmax = t->ntmax + 1, ld = insmax + 1;
    double *zfn = xcalloc((size_t)ld * ld, 8);
    for (int jn = 1; jn <= insmax; jn++) zfn_row(jn, zfn + (size_t)jn * ld);
    zfn[0] = 2.0;
    for (int im = 0; im <= nsmax; im++) {
      int ila = (nsmax - im + 2) / 2, ils = (nsmax - im + 3) / 2;
      int idglu = IMIN(t->ndgnh, t->ndglu[im]);
      t->rpnma[im] = xcalloc((size_t)idglu * ila, 8);
      t->rpnms[im] = xcalloc((size_t)idglu * ils, 8);
    }
#pragma omp parallel
    {
      double *zlfpol = xcalloc((size_t)ld * ld, 8);
#pragma omp for schedule(dynamic, 1)
      for (int jgl = 1; jgl <= t->ndgnh; jgl++) {
        supol(&pol, insmax, t->rmu[jgl - 1], zfn, zlfpol);
        for (int im = 0; im <= nsmax; im++) {
          int isl = IMAX(t->ndgnh - t->ndglu[im] + 1, 1);
          if (jgl < isl) continue;
          int ila = (nsmax - im + 2) / 2, ils = (nsmax - im + 3) / 2;
          int idglu = IMIN(t->ndgnh, t->ndglu[im]);
          int ia = 1 + (nsmax - im + 2) % 2, is = 1 + (nsmax - im + 1) % 2;
          /* block position k (1-based) holds n = insmax - (k-1) */
          for (int ji = 1; ji <= ila; ji++) {
            int n = insmax - (ia + (ji - 1) * 2 - 1);
            t->rpnma[im][(size_t)(ji - 1) * idglu + (jgl - isl)] = zlfpol[(size_t)im * ld + n];
          }
          for (int ji = 1; ji <= ils; ji++) {
            int n = insmax - (is + (ji - 1) * 2 - 1);
            t->rpnms[im][(size_t)(ji - 1) * idglu + (jgl - isl)] = zlfpol[(size_t)im * ld + n];
          }
        }
      }
      free(zlfpol);
    }
    free(zfn);
  }
  end_pol(&pol);
  return t;
}

orc_trans *orc_setup(int nsmax, int ndgl, const int *nloen_in, int belusov, double ra) {
  return orc_setup_impl(nsmax, ndgl, nloen_in, belusov, ra, 0);
}
orc_trans *orc_setup_lazy(int nsmax, int ndgl, const int *nloen_in, double ra) {
  return orc_setup_impl(nsmax, ndgl, nloen_in, 0, ra, 1);
}

void orc_set_sp_mode(orc_trans *t, int on) { t->sp_mode = on; }

void orc_free(orc_trans *t) {
  if (!t) return;
  if (t->pol) {
    end_pol((pol_t *)t->pol);
    free(t->pol);
  }
  for (int m = 0; m <= t->nsmax; m++) {
    free(t->rpnma[m]);
    free(t->rpnms[m]);
  }
  free(t->rpnma), free(t->rpnms);
  free(t->nloen), free(t->nmen), free(t->ndglu), free(t->nasm0), free(t->npmt), free(t->gpoff);
  free(t->rmu), free(t->rw), free(t->r1mu2), free(t->racthe);
  free(t->repsnm), free(t->rn), free(t->rlapin);
  free(t);
}

int orc_nspec2(const orc_trans *t) { return t->nspec2; }
int orc_ngptot(const orc_trans *t) { return t->ngptot; }
const double *orc_rmu(const orc_trans *t) { return t->rmu; }
const double *orc_rw(const orc_trans *t) { return t->rw; }
const int *orc_nmen(const orc_trans *t) { return t->nmen; }
const int *orc_ndglu(const orc_trans *t) { return t->ndglu; }
const int *orc_nasm0(const orc_trans *t) { return t->nasm0; }
const double *orc_rpnma(const orc_trans *t, int m, int *rows, int *cols) {
  *rows = IMIN(t->ndgnh, t->ndglu[m]);
  *cols = (t->nsmax - m + 2) / 2;
  return t->rpnma[m]; /* NULL after orc_setup_lazy */
}
const double *orc_rpnms(const orc_trans *t, int m, int *rows, int *cols) {
  *rows = IMIN(t->ndgnh, t->ndglu[m]);
  *cols = (t->nsmax - m + 3) / 2;
  return t->rpnms[m];
}

/* PREPSNM (cpu/internal/prepsnm_mod.F90:74-80) */
static void prepsnm(const orc_trans *t, int km, double *pepsnm /* [0..ntmax+2] */) {
  for (int jn = 0; jn < km; jn++) pepsnm[jn] = 0.0;
  for (int jn = km; jn <= t->ntmax + 2; jn++) pepsnm[jn] = t->repsnm[t->npmt[km] + jn - km];
}

/* Column-major 2-D work array helper: A(i,j) 1-based with leading dimension ld */
#define A2(a, ld, i, j) (a)[(size_t)((j)-1) * (ld) + ((i)-1)]

/* PRFI1B (cpu/internal/prfi1b_mod.F90:81-115).  psp: sp[ispec*nfld + f] */
static void prfi1b(const orc_trans *t, int km, double *pia, int ld, const double *psp, int nfld_arr,
                   int kfields) {
  int ilcm = t->nsmax + 1 - km, ioff = t->nasm0[km];
  for (int j = 1; j <= ilcm; j++) {
    int inm = ioff + (ilcm - j) * 2;
    for (int jfld = 1; jfld <= kfields; jfld++) {
      int ir = 2 * (jfld - 1) + 1, ii = ir + 1;
      A2(pia, ld, j + 2, ir) = psp[(size_t)(inm - 1) * nfld_arr + (jfld - 1)];
      A2(pia, ld, j + 2, ii) = psp[(size_t)(inm + 1 - 1) * nfld_arr + (jfld - 1)];
    }
  }
  for (int jfld = 1; jfld <= 2 * kfields; jfld++) {
    A2(pia, ld, 1, jfld) = 0.0;
    A2(pia, ld, 2, jfld) = 0.0;
    A2(pia, ld, ilcm + 3, jfld) = 0.0;
  }
}

/* VDTUV (cpu/internal/vdtuv_mod.F90:97-143).  Work arrays ZN/ZLAPIN/ZEPSNM(-1:..) */
static void vdtuv(const orc_trans *t, int km, int kfield, const double *pepsnm, const double *pvor,
                  const double *pdiv, double *pu, double *pv, int ld) {
  int ismax = t->nsmax;
  int nw = ismax + 6;
  double *zn = xcalloc(nw, 8), *zlapin = xcalloc(nw, 8), *zepsnm = xcalloc(nw, 8);
#define ZN(i) zn[(i) + 1]
#define ZLAPIN(i) zlapin[(i) + 1]
#define ZEPSNM(i) zepsnm[(i) + 1]
  double zkm = (double)km;
  for (int jn = km - 1; jn <= ismax + 2; jn++) {
    int ij = ismax + 3 - jn;
    ZN(ij) = t->rn[jn + 1];
    ZLAPIN(ij) = t->rlapin[jn + 1];
    if (jn >= 0) ZEPSNM(ij) = pepsnm[jn];
  }
  ZN(0) = t->rn[ismax + 3 + 1];
  if (km == 0) {
    for (int j = 1; j <= kfield; j++) {
      int ir = 2 * j - 1;
      for (int ji = 2; ji <= ismax + 3 - km; ji++) {
        A2(pu, ld, ji, ir) = +ZN(ji + 1) * ZEPSNM(ji) * ZLAPIN(ji + 1) * A2(pvor, ld, ji + 1, ir) -
                             ZN(ji - 2) * ZEPSNM(ji - 1) * ZLAPIN(ji - 1) * A2(pvor, ld, ji - 1, ir);
        A2(pv, ld, ji, ir) = -ZN(ji + 1) * ZEPSNM(ji) * ZLAPIN(ji + 1) * A2(pdiv, ld, ji + 1, ir) +
                             ZN(ji - 2) * ZEPSNM(ji - 1) * ZLAPIN(ji - 1) * A2(pdiv, ld, ji - 1, ir);
      }
    }
  } else {
    for (int j = 1; j <= kfield; j++) {
      int ir = 2 * j - 1, ii = ir + 1;
      for (int ji = 2; ji <= ismax + 3 - km; ji++) {
        A2(pu, ld, ji, ir) = -zkm * ZLAPIN(ji) * A2(pdiv, ld, ji, ii) +
                             ZN(ji + 1) * ZEPSNM(ji) * ZLAPIN(ji + 1) * A2(pvor, ld, ji + 1, ir) -
                             ZN(ji - 2) * ZEPSNM(ji - 1) * ZLAPIN(ji - 1) * A2(pvor, ld, ji - 1, ir);
        A2(pu, ld, ji, ii) = +zkm * ZLAPIN(ji) * A2(pdiv, ld, ji, ir) +
                             ZN(ji + 1) * ZEPSNM(ji) * ZLAPIN(ji + 1) * A2(pvor, ld, ji + 1, ii) -
                             ZN(ji - 2) * ZEPSNM(ji - 1) * ZLAPIN(ji - 1) * A2(pvor, ld, ji - 1, ii);
        A2(pv, ld, ji, ir) = -zkm * ZLAPIN(ji) * A2(pvor, ld, ji, ii) -
                             ZN(ji + 1) * ZEPSNM(ji) * ZLAPIN(ji + 1) * A2(pdiv, ld, ji + 1, ir) +
                             ZN(ji - 2) * ZEPSNM(ji - 1) * ZLAPIN(ji - 1) * A2(pdiv, ld, ji - 1, ir);
        A2(pv, ld, ji, ii) = +zkm * ZLAPIN(ji) * A2(pvor, ld, ji, ir) -
                             ZN(ji + 1) * ZEPSNM(ji) * ZLAPIN(ji + 1) * A2(pdiv, ld, ji + 1, ii) +
                             ZN(ji - 2) * ZEPSNM(ji - 1) * ZLAPIN(ji - 1) * A2(pdiv, ld, ji - 1, ii);
      }
    }
  }
  free(zn), free(zlapin), free(zepsnm);
#undef ZLAPIN
}

/* SPNSDE (cpu/internal/spnsde_mod.F90:95-114) */
static void spnsde(const orc_trans *t, int km, int kf_scalars, const double *pepsnm, const double *pf,
                   double *pnsd, int ld) {
  int ismax = t->nsmax, nw = ismax + 6;
  double *zn = xcalloc(nw, 8), *zepsnm = xcalloc(nw, 8);
  for (int jn = km - 1; jn <= ismax + 2; jn++) {
    int ij = ismax + 3 - jn;
    ZN(ij) = t->rn[jn + 1];
    if (jn >= 0) ZEPSNM(ij) = pepsnm[jn];
  }
  ZN(0) = t->rn[ismax + 3 + 1];
  int iskip = (km == 0) ? 2 : 1;
  for (int j = 1; j <= 2 * kf_scalars; j += iskip)
    for (int ji = 2; ji <= ismax + 3 - km; ji++)
      A2(pnsd, ld, ji, j) = -ZN(ji + 1) * ZEPSNM(ji) * A2(pf, ld, ji + 1, j) + ZN(ji - 2) * ZEPSNM(ji - 1) * A2(pf, ld, ji - 1, j);
  free(zn), free(zepsnm);
#undef ZN
#undef ZEPSNM
}

/* Fourier-space buffer of this oracle: four[(lat*(nsmax+1) + m)*2*nf + 2*(f)+{0,1}].
 * With NPROC=1 the reference's FOUBUF offsets (NSTAGT0B/NPNTGTB*) are a private
 * permutation of exactly this (lat,m,field) set (asre1b_mod.F90:87-99,
 * fourier_in_mod.F90:64-76); any bijection is equivalent.                               */
#define FOUR(t_, four, nf, lat, m, c) (four)[((size_t)((lat)-1) * ((t_)->nsmax + 1) + (m)) * 2 * (nf) + (c)]

/* LTINV (cpu/internal/ltinv_mod.F90:139-329) + LEINV (leinv_mod.F90:92-186, DGEMM restated
 * as loops) + ASRE1B (asre1b_mod.F90:83-102) for one m                                 */
static void ltinv(const orc_trans *t, int km, int kf_uv, int kf_scalars, int kf_scders, int kf_out_lt,
                  int lvorgp, int ldivgp, const double *spvor, const double *spdiv, const double *spsc,
                  double *four) {
  int nlei1 = t->nlei1;
  int klei2 = 8 * kf_uv + 2 * kf_scalars + 2 * kf_scders;
  double *zia = xcalloc((size_t)nlei1 * (klei2 > 0 ? klei2 : 1), 8);
  double *zepsnm = xcalloc((size_t)t->ntmax + 3, 8);
  prepsnm(t, km, zepsnm);
#define COL(j) (zia + (size_t)((j)-1) * nlei1)
  int ilast = 4 * kf_uv, ifirst;
  if (kf_uv > 0) {
    int ivorl = 1, idivl = 2 * kf_uv + 1, iul = 4 * kf_uv + 1, ivl = 6 * kf_uv + 1;
    prfi1b(t, km, COL(ivorl), nlei1, spvor, kf_uv, kf_uv);
    prfi1b(t, km, COL(idivl), nlei1, spdiv, kf_uv, kf_uv);
    ilast += 4 * kf_uv;
    vdtuv(t, km, kf_uv, zepsnm, COL(ivorl), COL(idivl), COL(iul), COL(ivl), nlei1);
  }
  if (kf_scalars > 0) {
    ifirst = ilast + 1;
    ilast = ifirst - 1 + 2 * kf_scalars;
    prfi1b(t, km, COL(ifirst), nlei1, spsc, kf_scalars, kf_scalars);
  }
  if (kf_scders > 0) {
    int isl = 2 * (4 * kf_uv) + 1, idl = 2 * (4 * kf_uv + kf_scalars) + 1;
    spnsde(t, km, kf_scalars, zepsnm, COL(isl), COL(idl), nlei1);
  }
  int ista = 1, ifc = 2 * kf_out_lt;
  if (kf_uv > 0 && !lvorgp) ista += 2 * kf_uv;
  if (kf_uv > 0 && !ldivgp) ista += 2 * kf_uv;
  int idglu = IMIN(t->ndgnh, t->ndglu[km]);
  int isl = IMAX(t->ndgnh - t->ndglu[km] + 1, 1);
  /* LEINV */
  int ia = 1 + (t->nsmax - km + 2) % 2, is = 1 + (t->nsmax - km + 1) % 2;
  int ila = (t->nsmax - km + 2) / 2, ils = (t->nsmax - km + 3) / 2;
  int iskip = (km == 0) ? 2 : 1;
  const double *pia = COL(ista);
  const double *rpa = t->rpnma[km], *rps = t->rpnms[km];
  double *lazy_a = NULL, *lazy_s = NULL;
  if (t->lazy) { /* orc_setup_lazy: this wavenumber's panels are built here and dropped at the end */
    supolf_panels(t, (const pol_t *)t->pol, km, &lazy_a, &lazy_s);
    rpa = lazy_a, rps = lazy_s;
  }
  /* DGEMM('N','N') restated in blocks of ORC_NB columns: each panel column is read once per block instead of
   * once per output column.  For every output element the sum still runs over j = 1..ILA in ascending
   * order, one multiply and one add at a time (no contraction), so the values are those of the
   * column-by-column loops.                                                                            */
  const size_t ldz = (size_t)(idglu > 0 ? idglu : 1);
  double *zca = xcalloc(ldz * ORC_NB, 8), *zcs = xcalloc(ldz * ORC_NB, 8);
  for (int jk0 = 1; jk0 <= ifc; jk0 += ORC_NB) {
    int nb = IMIN(ORC_NB, ifc - jk0 + 1);
    int active[ORC_NB];
    double b[ORC_NB];
    for (int c = 0; c < nb; c++) active[c] = ((jk0 + c - 1) % iskip) == 0; /* m=0: imaginary columns are zero */
    for (size_t i = 0; i < ldz * ORC_NB; i++) zca[i] = zcs[i] = 0.0;
    for (int j = 1; j <= ila; j++) {
      const double *col = rpa + (size_t)(j - 1) * idglu;
      for (int c = 0; c < nb; c++) b[c] = A2(pia, nlei1, ia + 1 + (j - 1) * 2, jk0 + c);
      for (int c = 0; c < nb; c++) {
        if (!active[c]) continue;
        double *z = zca + (size_t)c * ldz;
        const double bc = b[c];
        for (int ji = 0; ji < idglu; ji++) z[ji] += col[ji] * bc;
      }
    }
    for (int j = 1; j <= ils; j++) {
      const double *col = rps + (size_t)(j - 1) * idglu;
      for (int c = 0; c < nb; c++) b[c] = A2(pia, nlei1, is + 1 + (j - 1) * 2, jk0 + c);
      for (int c = 0; c < nb; c++) {
        if (!active[c]) continue;
        double *z = zcs + (size_t)c * ldz;
        const double bc = b[c];
        for (int ji = 0; ji < idglu; ji++) z[ji] += col[ji] * bc;
      }
    }
    for (int c = 0; c < nb; c++) {
      const int jk = jk0 + c;
      const double *za = zca + (size_t)c * ldz, *zs = zcs + (size_t)c * ldz;
      for (int ji = 1; ji <= idglu; ji++) {
        int jgl = isl + ji - 1, igls = t->ndgl + 1 - jgl;
        /* ASRE1B: north = A+S, south = S-A */
        FOUR(t, four, kf_out_lt, jgl, km, jk - 1) = za[ji - 1] + zs[ji - 1];
        FOUR(t, four, kf_out_lt, igls, km, jk - 1) = zs[ji - 1] - za[ji - 1];
      }
    }
  }
  free(zca), free(zcs);
  free(lazy_a), free(lazy_s);
  free(zia);
  free(zepsnm);
#undef COL
}

static int inv_counts(int nuv, int nsc, int lscders, int lvorgp, int ldivgp, int luvder, int *kf_scders,
                      int *kf_out_lt, int *kf_fs) {
  /* inv_trans.F90:352-387 */
  if (lvorgp) ldivgp = 1;
  int if_scders = (nsc > 0 && lscders) ? nsc : 0;
  int if_out_lt = 2 * nuv + nsc + if_scders;
  if (nuv > 0 && lvorgp) if_out_lt += nuv;
  if (nuv > 0 && ldivgp) if_out_lt += nuv;
  int if_fs = if_out_lt + if_scders;
  if (nuv > 0 && luvder) if_fs += 2 * nuv;
  int if_gp = 2 * nuv + nsc;
  if (nsc > 0 && lscders) if_gp += 2 * nsc;
  if (nuv > 0 && lvorgp) if_gp += nuv;
  if (nuv > 0 && ldivgp) if_gp += nuv;
  if (nuv > 0 && luvder) if_gp += 2 * nuv;
  *kf_scders = if_scders;
  *kf_out_lt = if_out_lt;
  *kf_fs = if_fs;
  return if_gp;
}

int orc_inv_trans(const orc_trans *t, int nuv, int nsc, const double *spvor, const double *spdiv,
                  const double *spsc, int lscders, int lvorgp, int ldivgp, int luvder, double *gp) {
  if (lvorgp) ldivgp = 1; /* inv_trans.F90:350 */
  if (nuv == 0) luvder = 0;
  if (nsc == 0) lscders = 0;
  int kf_scders, kf_out_lt, kf_fs;
  int kf_gp = inv_counts(nuv, nsc, lscders, lvorgp, ldivgp, luvder, &kf_scders, &kf_out_lt, &kf_fs);
  int nm = t->nsmax + 1;
  double *four = xcalloc((size_t)t->ndgl * nm * 2 * (kf_out_lt > 0 ? kf_out_lt : 1), 8);
  /* LTINV_CTL: loop over m (ltinv_ctl_mod.F90:118-138) */
  const double tt0 = omp_get_wtime();
#pragma omp parallel for schedule(dynamic, 1)
  for (int km = 0; km <= t->nsmax; km++)
    ltinv(t, km, nuv, nsc, kf_scders, kf_out_lt, lvorgp, ldivgp, spvor, spdiv, spsc, four);
  if (getenv("ORC_TIMING")) fprintf(stderr, "orc_inv_trans: Legendre %.3f s\n", omp_get_wtime() - tt0);
  /* FTINV_CTL (ftinv_ctl_mod.F90:173-191): per latitude FOURIER_IN, FSC, FTINV; then TRLTOG
   * (NPROC=1: pure copy ZGTF(field, point) -> PGP(point, field, 1)).                    */
#pragma omp parallel
  {
    int maxn = t->ndlon;
    double *row = xcalloc(((size_t)maxn + 3) * (kf_fs > 0 ? kf_fs : 1), 8); /* [f][NLOEN+3] */
    double *outr = xcalloc((size_t)maxn, 8);
    cplx *work = xcalloc(2 * (size_t)maxn, sizeof(cplx));
#pragma omp for schedule(dynamic, 1)
    for (int jgl = 1; jgl <= t->ndgl; jgl++) {
      int n = t->nloen[jgl - 1], imen = t->nmen[jgl - 1], ldr = maxn + 3;
      memset(row, 0, sizeof(double) * (size_t)ldr * kf_fs);
      /* FOURIER_IN (fourier_in_mod.F90:64-76) */
      for (int jm = 0; jm <= imen; jm++)
        for (int jf = 0; jf < kf_out_lt; jf++) {
          row[(size_t)jf * ldr + 2 * jm] = FOUR(t, four, kf_out_lt, jgl, jm, 2 * jf);
          row[(size_t)jf * ldr + 2 * jm + 1] = FOUR(t, four, kf_out_lt, jgl, jm, 2 * jf + 1);
        }
      /* FSC (fsc_mod.F90:138-187) */
      int ist = 0;
      if (nuv > 0 && lvorgp) ist += nuv;
      if (nuv > 0 && ldivgp) ist += nuv;
      int iuv = ist;
      ist += 2 * nuv;
      int isc = ist;
      ist += nsc;
      int insd = ist;
      ist += kf_scders;
      int iuvd = ist;
      if (luvder) ist += 2 * nuv;
      int iewd = ist;
      double zachte = t->racthe[jgl - 1];
      for (int jf = 0; jf < 2 * nuv; jf++)
        for (int j = 0; j < 2 * (imen + 1); j++) row[(size_t)(iuv + jf) * ldr + j] *= zachte;
      for (int jf = 0; jf < kf_scders; jf++)
        for (int j = 0; j < 2 * (imen + 1); j++) row[(size_t)(insd + jf) * ldr + j] *= zachte;
      if (luvder)
        for (int jm = 0; jm <= imen; jm++) {
          double zmul = zachte * (double)jm;
          for (int jf = 0; jf < 2 * nuv; jf++) {
            row[(size_t)(iuvd + jf) * ldr + 2 * jm] = -row[(size_t)(iuv + jf) * ldr + 2 * jm + 1] * zmul;
            row[(size_t)(iuvd + jf) * ldr + 2 * jm + 1] = row[(size_t)(iuv + jf) * ldr + 2 * jm] * zmul;
          }
        }
      if (kf_scders > 0)
        for (int jm = 0; jm <= imen; jm++) {
          double zmul = zachte * (double)jm;
          for (int jf = 0; jf < nsc; jf++) {
            row[(size_t)(iewd + jf) * ldr + 2 * jm] = -row[(size_t)(isc + jf) * ldr + 2 * jm + 1] * zmul;
            row[(size_t)(iewd + jf) * ldr + 2 * jm + 1] = row[(size_t)(isc + jf) * ldr + 2 * jm] * zmul;
          }
        }
      /* FTINV (ftinv_mod.F90:65-84): tail already zero; c2r of length NLOEN, unscaled */
      fftplan pl = fftplan_make(n);
      for (int jf = 0; jf < kf_fs; jf++) {
        if (n > 1) {
          c2r_plan(&pl, row + (size_t)jf * ldr, work, outr);
        } else {
          outr[0] = row[(size_t)jf * ldr];
        }
        /* TRLTOG: Fourier-space field jf -> grid field (same order, ftinv_ctl_mod.F90:228-262) */
        memcpy(gp + (size_t)jf * t->ngptot + t->gpoff[jgl - 1], outr, sizeof(double) * n);
      }
      fftplan_free(&pl);
    }
    free(row), free(outr), free(work);
  }
  free(four);
  (void)kf_gp;
  return kf_fs;
}

/* VORDIV_TO_UV (cpu/external/vordiv_to_uv.F90:11-178 -> VD2UV_CTL -> VD2UV, cpu/internal/vd2uv_mod.F90:79-120): spectral vorticity /
 * divergence -> spectral U = u cos(theta), V = v cos(theta), coefficients n <= NSMAX only, scaled by 1 / RA.                  */
void orc_vordiv_to_uv(const orc_trans *t, int nuv, const double *spvor, const double *spdiv, double *spu, double *spv) {
  if (nuv <= 0) return;
  int ld = t->nlei1;
#pragma omp parallel for schedule(dynamic, 1)
  for (int km = 0; km <= t->nsmax; km++) {
    double *zia = xcalloc((size_t)ld * 8 * nuv, 8), *zeps = xcalloc(t->ntmax + 3, 8);
    prepsnm(t, km, zeps);
    double *vor = zia, *div = zia + (size_t)ld * 2 * nuv, *pu = zia + (size_t)ld * 4 * nuv, *pv = zia + (size_t)ld * 6 * nuv;
    prfi1b(t, km, vor, ld, spvor, nuv, nuv);
    prfi1b(t, km, div, ld, spdiv, nuv, nuv);
    vdtuv(t, km, nuv, zeps, vor, div, pu, pv, ld);
    int ilcm = t->nsmax + 1 - km, ioff = t->nasm0[km];
    double za_r = 1.0 / t->ra;
    for (int j = 1; j <= ilcm; j++) { /* vd2uv_mod.F90:104-114 */
      int inm = ioff + (ilcm - j) * 2;
      for (int jfld = 1; jfld <= nuv; jfld++) {
        int ir = 2 * (jfld - 1) + 1, ii = ir + 1;
        spu[(size_t)(inm - 1) * nuv + (jfld - 1)] = A2(pu, ld, j + 2, ir) * za_r;
        spu[(size_t)(inm) * nuv + (jfld - 1)] = A2(pu, ld, j + 2, ii) * za_r;
        spv[(size_t)(inm - 1) * nuv + (jfld - 1)] = A2(pv, ld, j + 2, ir) * za_r;
        spv[(size_t)(inm) * nuv + (jfld - 1)] = A2(pv, ld, j + 2, ii) * za_r;
      }
    }
    free(zia), free(zeps);
  }
}

/* GPNORM_TRANS (cpu/external/gpnorm_trans.F90:11-96 -> GPNORM_TRANS_CTL, cpu/internal/gpnorm_trans_ctl_mod.F90:170-210, 436-443)
 * for one task: per latitude the sum over the longitudes in double, times RW(lat) / NLOEN(lat); the average is the sum of these over
 * the latitudes in latitude order; minimum and maximum over all points.  gp[f*ngptot + p].                                       */
void orc_gpnorm(const orc_trans *t, int nfld, const double *gp, double *ave, double *pmin, double *pmax) {
  for (int jf = 0; jf < nfld; jf++) {
    const double *f = gp + (size_t)jf * t->ngptot;
    double a = 0.0, mn = f[0], mx = f[0];
    for (int jgl = 1; jgl <= t->ndgl; jgl++) {
      int n = t->nloen[jgl - 1];
      const double *row = f + t->gpoff[jgl - 1];
      double zave = 0.0;
      for (int jl = 0; jl < n; jl++) {
        zave += row[jl];
        if (row[jl] < mn) mn = row[jl];
        if (row[jl] > mx) mx = row[jl];
      }
      a += zave * t->rw[jgl - 1] / (double)n;
    }
    ave[jf] = a, pmin[jf] = mn, pmax[jf] = mx;
  }
}

/* LTDIR (cpu/internal/ltdir_mod.F90:128-193): PRFI2B + LDFOU2 + LEDIR + UVTVD + UPDSP   */
static void ltdir(const orc_trans *t, int km, int kf_fs, int kf_uv, int kf_scalars, const double *four,
                  double *spvor, double *spdiv, double *spsc) {
  int ifc = 2 * kf_fs;
  int idglu = IMIN(t->ndgnh, t->ndglu[km]);
  int isl = IMAX(t->ndgnh - t->ndglu[km] + 1, 1);
  int nled4 = t->nled4, itmax = t->ntmax, ismax = t->nsmax;
  double *zoa1 = xcalloc((size_t)nled4 * ifc, 8);
  double *zoa2 = xcalloc((size_t)nled4 * (4 * kf_uv > 0 ? 4 * kf_uv : 1), 8);
  double *zepsnm = xcalloc((size_t)itmax + 3, 8);
  int ia = 1 + (itmax - km + 2) % 2, is = 1 + (itmax - km + 1) % 2;
  int ila = (itmax - km + 2) / 2, ils = (itmax - km + 3) / 2;
  int iskip = (km == 0) ? 2 : 1;
  const double *rpa = t->rpnma[km], *rps = t->rpnms[km];
  double *lazy_a = NULL, *lazy_s = NULL;
  if (t->lazy) { /* orc_setup_lazy: this wavenumber's panels are built here and dropped at the end */
    supolf_panels(t, (const pol_t *)t->pol, km, &lazy_a, &lazy_s);
    rpa = lazy_a, rps = lazy_s;
  }
  /* blocks of ORC_NB columns, operands stored [latitude][column] so that the loop over the columns of a
   * block is the vector loop; every dot product still runs over the latitudes in ascending order, one
   * multiply and one add at a time -- the values of the column-by-column loops                       */
  const size_t ldz = (size_t)(idglu > 0 ? idglu : 1);
  double *zba = xcalloc(ldz * ORC_NB, 8), *zbs = xcalloc(ldz * ORC_NB, 8);
  int ncol = 0;
  for (int jk = 1; jk <= ifc; jk += iskip) ncol++;
  for (int c0 = 0; c0 < ncol; c0 += ORC_NB) {
    int nb = IMIN(ORC_NB, ncol - c0);
    for (size_t i = 0; i < ldz * ORC_NB; i++) zba[i] = zbs[i] = 0.0;
    /* PRFI2B (prfi2b_mod.F90:82-94), LDFOU2 (ldfou2_mod.F90:85-96), ZB=PAIA*PW (ledir_mod.F90:118-124) */
    for (int c = 0; c < nb; c++) {
      const int jk = 1 + (c0 + c) * iskip;
      for (int j = 1; j <= idglu; j++) {
        int jgl = isl + j - 1, igls = t->ndgl + 1 - jgl;
        double fn = FOUR(t, four, kf_fs, jgl, km, jk - 1), fs = FOUR(t, four, kf_fs, igls, km, jk - 1);
        double psia = fn + fs, paia = fn - fs;
        if (jk <= 4 * kf_uv) {
          double zacthe = t->racthe[jgl - 1];
          paia *= zacthe;
          psia *= zacthe;
        }
        if (t->sp_mode) { /* float storage and float arithmetic of PRFI2B / LDFOU2 / ZB = PAIA * PW */
          const float ffn = (float)fn, ffs = (float)fs, fw = (float)t->rw[jgl - 1];
          float fps = ffn + ffs, fpa = ffn - ffs;
          if (jk <= 4 * kf_uv) fpa *= (float)t->racthe[jgl - 1], fps *= (float)t->racthe[jgl - 1];
          zba[(size_t)(j - 1) * ORC_NB + c] = (double)(fpa * fw);
          zbs[(size_t)(j - 1) * ORC_NB + c] = (double)(fps * fw);
          continue;
        }
        zba[(size_t)(j - 1) * ORC_NB + c] = paia * t->rw[jgl - 1];
        zbs[(size_t)(j - 1) * ORC_NB + c] = psia * t->rw[jgl - 1];
      }
    }
    /* LEDIR GEMM('T','N') (ledir_mod.F90:130,204) + scatter (ledir_mod.F90:181-187,255-261) */
    if (t->sp_mode) {
      /* the single-precision library (JPRB = JPRM): operands are stored in float -- the Fourier coefficients, ZB =
       * PAIA * PW (ledir_mod.F90:118-124, a float product) and the Legendre matrices -- and the GEMM is SGEMM, except
       * for the mean wavenumber m = 0: "DGEM for the mean to improve mass conservation" (ledir_mod.F90:133-171):
       * operands promoted to double, DGEMM, result rounded to float.  The float sums run in the restated GEMM's
       * order (one multiply, one add at a time); the double ones need no order to be right to a float ulp.      */
      for (int par = 0; par < 2; par++) {
        const int il = par ? ils : ila, i0 = par ? is : ia;
        const double *rp = par ? rps : rpa, *zb0 = par ? zbs : zba;
        for (int j = 1; j <= il; j++) {
          const double *row = rp + (size_t)(j - 1) * idglu;
          for (int c = 0; c < nb; c++) {
            const int jk = 1 + (c0 + c) * iskip;
            float out;
            if (km == 0) {
              double acc = 0.0;
              for (int k = 0; k < idglu; k++) acc += (double)(float)row[k] * (double)(float)zb0[(size_t)k * ORC_NB + c];
              out = (float)acc;
            } else {
              float acc = 0.0f;
              for (int k = 0; k < idglu; k++) acc += (float)row[k] * (float)zb0[(size_t)k * ORC_NB + c];
              out = acc;
            }
            A2(zoa1, nled4, i0 + (j - 1) * 2, jk) = (double)out;
          }
        }
      }
      continue;
    }
    for (int j = 1; j <= ila; j++) {
      double sv[ORC_NB];
      const double *row = rpa + (size_t)(j - 1) * idglu;
      for (int c = 0; c < ORC_NB; c++) sv[c] = 0.0;
      for (int k = 0; k < idglu; k++) {
        const double r = row[k];
        const double *zb = zba + (size_t)k * ORC_NB;
        for (int c = 0; c < ORC_NB; c++) sv[c] += r * zb[c];
      }
      for (int c = 0; c < nb; c++) A2(zoa1, nled4, ia + (j - 1) * 2, 1 + (c0 + c) * iskip) = sv[c];
    }
    for (int j = 1; j <= ils; j++) {
      double sv[ORC_NB];
      const double *row = rps + (size_t)(j - 1) * idglu;
      for (int c = 0; c < ORC_NB; c++) sv[c] = 0.0;
      for (int k = 0; k < idglu; k++) {
        const double r = row[k];
        const double *zb = zbs + (size_t)k * ORC_NB;
        for (int c = 0; c < ORC_NB; c++) sv[c] += r * zb[c];
      }
      for (int c = 0; c < nb; c++) A2(zoa1, nled4, is + (j - 1) * 2, 1 + (c0 + c) * iskip) = sv[c];
    }
  }
  free(zba), free(zbs);
  free(lazy_a), free(lazy_s);
#define COL1(j) (zoa1 + (size_t)((j)-1) * nled4)
#define COL2(j) (zoa2 + (size_t)((j)-1) * nled4)
  if (kf_uv > 0) { /* UVTVD (uvtvd_mod.F90:91-139) */
    prepsnm(t, km, zepsnm);
    double *pu = COL1(1), *pv = COL1(2 * kf_uv + 1), *pvor = COL2(1), *pdiv = COL2(2 * kf_uv + 1);
    double zkm = (double)km;
    int in0 = itmax + 2 - (km - 1); /* NLTN(KM-1) */
    for (int j = 1; j <= 2 * kf_uv; j++) {
      A2(pu, nled4, in0, j) = 0.0;
      A2(pv, nled4, in0, j) = 0.0;
    }
#define ZN(i) t->rn[(i) + 1]
    for (int jn = km; jn <= itmax; jn++) {
      int in = itmax + 2 - jn;
      for (int j = 1; j <= kf_uv; j++) {
        int ir = 2 * j - 1, ii = ir + 1;
        if (km != 0) {
          A2(pvor, nled4, in, ir) = -zkm * A2(pv, nled4, in, ii) - ZN(jn) * zepsnm[jn + 1] * A2(pu, nled4, in - 1, ir) +
                                    ZN(jn + 1) * zepsnm[jn] * A2(pu, nled4, in + 1, ir);
          A2(pvor, nled4, in, ii) = +zkm * A2(pv, nled4, in, ir) - ZN(jn) * zepsnm[jn + 1] * A2(pu, nled4, in - 1, ii) +
                                    ZN(jn + 1) * zepsnm[jn] * A2(pu, nled4, in + 1, ii);
          A2(pdiv, nled4, in, ir) = -zkm * A2(pu, nled4, in, ii) + ZN(jn) * zepsnm[jn + 1] * A2(pv, nled4, in - 1, ir) -
                                    ZN(jn + 1) * zepsnm[jn] * A2(pv, nled4, in + 1, ir);
          A2(pdiv, nled4, in, ii) = +zkm * A2(pu, nled4, in, ir) + ZN(jn) * zepsnm[jn + 1] * A2(pv, nled4, in - 1, ii) -
                                    ZN(jn + 1) * zepsnm[jn] * A2(pv, nled4, in + 1, ii);
        } else {
          A2(pvor, nled4, in, ir) = -ZN(jn) * zepsnm[jn + 1] * A2(pu, nled4, in - 1, ir) + ZN(jn + 1) * zepsnm[jn] * A2(pu, nled4, in + 1, ir);
          A2(pdiv, nled4, in, ir) = ZN(jn) * zepsnm[jn + 1] * A2(pv, nled4, in - 1, ir) - ZN(jn + 1) * zepsnm[jn] * A2(pv, nled4, in + 1, ir);
        }
      }
    }
#undef ZN
  }
  /* UPDSP/UPDSPB (updsp_mod.F90:100-161, updspb_mod.F90:92-149) */
  int iasm0 = t->nasm0[km];
  for (int pass = 0; pass < 3; pass++) {
    int kfield = pass < 2 ? kf_uv : kf_scalars;
    if (kfield == 0) continue;
    const double *poa = pass == 0 ? COL2(1) : pass == 1 ? COL2(2 * kf_uv + 1) : COL1(4 * kf_uv + 1);
    double *psp = pass == 0 ? spvor : pass == 1 ? spdiv : spsc;
    for (int jn = itmax + 2 - ismax; jn <= itmax + 2 - km; jn++) {
      int inm = iasm0 + ((itmax + 2 - jn) - km) * 2;
      for (int jfld = 1; jfld <= kfield; jfld++) {
        int ir = 2 * jfld - 1, ii = ir + 1;
        psp[(size_t)(inm - 1) * kfield + (jfld - 1)] = A2(poa, nled4, jn, ir);
        psp[(size_t)(inm) * kfield + (jfld - 1)] = (km == 0) ? 0.0 : A2(poa, nled4, jn, ii);
      }
    }
    if (km == 0 && pass < 2) /* updsp_mod.F90:113-126 */
      for (int jfld = 1; jfld <= kfield; jfld++) psp[(size_t)(t->nasm0[0] - 1) * kfield + (jfld - 1)] = 0.0;
  }
  free(zoa1), free(zoa2), free(zepsnm);
#undef COL1
#undef COL2
}

void orc_dir_trans(const orc_trans *t, int nuv, int nsc, const double *gp, double *spvor, double *spdiv,
                   double *spsc) {
  int kf_fs = 2 * nuv + nsc; /* dir_trans.F90:301 */
  int nm = t->nsmax + 1;
  double *four = xcalloc((size_t)t->ndgl * nm * 2 * (kf_fs > 0 ? kf_fs : 1), 8);
  /* FTDIR_CTL (ftdir_ctl_mod.F90:182-190): TRGTOL copy, FTDIR, FOURIER_OUT */
#pragma omp parallel
  {
    int maxn = t->ndlon;
    double *outc = xcalloc((size_t)maxn + 3, 8);
    cplx *work = xcalloc(2 * (size_t)maxn, sizeof(cplx));
#pragma omp for schedule(dynamic, 1)
    for (int jgl = 1; jgl <= t->ndgl; jgl++) {
      int n = t->nloen[jgl - 1], imen = t->nmen[jgl - 1];
      fftplan pl = fftplan_make(n);
      for (int jf = 0; jf < kf_fs; jf++) {
        const double *in = gp + (size_t)jf * t->ngptot + t->gpoff[jgl - 1];
        if (n > 1) {
          r2c_plan(&pl, in, work, outc);
          double zsc = 1.0 / (double)n; /* tpm_fftw.F90:317-321 */
          for (int j = 0; j < 2 * (n / 2 + 1); j++) outc[j] *= zsc;
        } else {
          outc[0] = in[0];
          outc[1] = 0.0;
        }
        /* FOURIER_OUT (fourier_out_mod.F90:64-76): keep m = 0..NMEN */
        for (int jm = 0; jm <= imen; jm++) {
          int have = (2 * jm + 1) < 2 * (n / 2 + 1);
          FOUR(t, four, kf_fs, jgl, jm, 2 * jf) = have ? outc[2 * jm] : 0.0;
          FOUR(t, four, kf_fs, jgl, jm, 2 * jf + 1) = have ? outc[2 * jm + 1] : 0.0;
        }
      }
      fftplan_free(&pl);
    }
    free(outc), free(work);
  }
  /* LTDIR_CTL (ltdir_ctl_mod.F90:90-98) */
  const double tt0 = omp_get_wtime();
#pragma omp parallel for schedule(dynamic, 1)
  for (int km = 0; km <= t->nsmax; km++) ltdir(t, km, kf_fs, nuv, nsc, four, spvor, spdiv, spsc);
  if (getenv("ORC_TIMING")) fprintf(stderr, "orc_dir_trans: Legendre %.3f s\n", omp_get_wtime() - tt0);
  free(four);
}

/* SPECNORM: SPNORMD (spnormd_mod.F90:40-57) + SPNORMC (spnormc_mod.F90:49-85) */
void orc_specnorm(const orc_trans *t, int nfld, const double *sp, double *norms) {
  for (int f = 0; f < nfld; f++) {
    double s = 0.0;
    for (int km = 0; km <= t->nsmax; km++) {
      double zsm = 0.0;
      int iasm0 = t->nasm0[km];
      if (km == 0) {
        for (int jn = 0; jn <= t->nsmax; jn++) {
          double v = sp[(size_t)(iasm0 + 2 * jn - 1) * nfld + f];
          zsm += v * v;
        }
      } else {
        for (int jn = 0; jn < 2 * (t->nsmax + 1 - km); jn++) {
          double v = sp[(size_t)(iasm0 + jn - 1) * nfld + f];
          zsm += 2.0 * v * v;
        }
      }
      s += zsm;
    }
    norms[f] = sqrt(s);
  }
}
