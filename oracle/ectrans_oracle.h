/*
 * ectrans_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C + OpenMP) of the ecTrans 1.7.0 CPU hot path
 * (SETUP_TRANS -> INV_TRANS / DIR_TRANS / SPECNORM, single MPI task).  It exists to
 * check the HIP product path; nothing under ectrans_amd/ may include, link or call it.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * Parity pin: tests/golden/tl149 (the reference's own ectrans4py fixtures,
 * /root/reference/tests/test_ectrans4py/data) -- see tests/test_oracle_golden.py.
 * The reference itself is NOT buildable in this image (needs un-vendored fiat + ecbuild
 * + sed-generated sources + FFTW/BLAS), so there is no oracle/_ref.
 *
 * Third-party arithmetic restated: FFTW 3.3.10 r2c/c2r semantics (unnormalised DFT,
 * half-complex layout; call sites tpm_fftw.F90:294-321) and BLAS DGEMM (call sites
 * ledir_mod.F90:130,204, leinv_mod.F90:133,166) as plain loops.
 *
 * Array layouts follow the Fortran API with NPROMA = NGPTOT (one block):
 *   spectral  sp[ispec*nfld + f]      == PSPEC(f+1, ispec+1)          (field fastest)
 *   grid      gp[f*ngptot + p]        == PGP(p+1, f+1, 1)             (point fastest)
 */
#ifndef ECTRANS_ORACLE_H
#define ECTRANS_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_trans orc_trans;

/* SETUP_TRANS (setup_trans.F90:169-428) for one MPI task.
 * belusov != 0  -> LDUSERPNM=.TRUE.  (API default; SUPOL/Belousov, suleg_mod.F90:402-488)
 * belusov == 0  -> LDUSERPNM=.FALSE. (benchmark + transi; SUPOLF per m, suleg_mod.F90:635-662)
 * ra: planet radius (setup_trans0.F90:129 default 6371229.0). */
orc_trans *orc_setup(int nsmax, int ndgl, const int *nloen, int belusov, double ra);
/* Same as orc_setup(..., belusov = 0, ...) but without stored Legendre panels: LTINV / LTDIR rebuild the panels of
 * their wavenumber with the same SUPOLF calls each time.  Identical results; for the TCo1279 / TCo2559 parity
 * tests, whose 6.4 / 51 GiB of panels would otherwise sit in host memory.  orc_rpnma/orc_rpnms return NULL. */
orc_trans *orc_setup_lazy(int nsmax, int ndgl, const int *nloen, double ra);
/* on != 0: orc_dir_trans computes LEDIR the way the reference's SINGLE-precision library does (ledir_mod.F90:118-171):
 * float operands, SGEMM for m >= 1, and for m = 0 "DGEM for the mean to improve mass conservation" -- operands
 * promoted to double, DGEMM, one rounding to float.  Everything else (FFT, UVTVD, setup) stays double: the mode
 * exists to pin the m = 0 behaviour of the fp32 product library, not to emulate libtrans_sp bit for bit. */
void orc_set_sp_mode(orc_trans *t, int on);
void orc_free(orc_trans *t);

/* TRANS_INQ subset (trans_inq.F90) */
int orc_nspec2(const orc_trans *t);
int orc_ngptot(const orc_trans *t);
const double *orc_rmu(const orc_trans *t);   /* ndgl */
const double *orc_rw(const orc_trans *t);    /* ndgl */
const int *orc_nmen(const orc_trans *t);     /* ndgl */
const int *orc_ndglu(const orc_trans *t);    /* nsmax+1 */
const int *orc_nasm0(const orc_trans *t);    /* nsmax+1, 1-based offsets as in Fortran */
/* Legendre panels as built by SULEG: column-major (ndglu(m) x ila|ils), n descending. */
const double *orc_rpnma(const orc_trans *t, int m, int *rows, int *cols);
const double *orc_rpnms(const orc_trans *t, int m, int *rows, int *cols);

/* P_n^km(mu(jgl)), n = km .. nsmax, as the panels hold them (one SUPOLF column): out[n - km]; jgl 1-based. */
void orc_legpol(const orc_trans *t, int km, int jgl, double *out);

/* INV_TRANS (inv_trans.F90:182-611).  Grid field order (inv_trans.h:66-76):
 * [vor][div] u v scalars [NS-ders] [u_EW v_EW] [sc_EW].  Returns number of grid fields. */
int orc_inv_trans(const orc_trans *t, int nuv, int nsc, const double *spvor,
                  const double *spdiv, const double *spsc, int lscders, int lvorgp,
                  int ldivgp, int luvder, double *gp);

/* DIR_TRANS (dir_trans.F90:160-502).  Grid field order: u(nuv) v(nuv) scalars(nsc). */
void orc_dir_trans(const orc_trans *t, int nuv, int nsc, const double *gp, double *spvor,
                   double *spdiv, double *spsc);

/* VORDIV_TO_UV (vordiv_to_uv.F90, vd2uv_mod.F90:79-120): spectral vor / div -> spectral U, V (u cos, v cos; n <= NSMAX; times 1 / RA) */
void orc_vordiv_to_uv(const orc_trans *t, int nuv, const double *spvor, const double *spdiv, double *spu, double *spv);

/* GPNORM_TRANS (gpnorm_trans.F90, gpnorm_trans_ctl_mod.F90): area-weighted average, minimum and maximum of grid fields gp[f*ngptot+p] */
void orc_gpnorm(const orc_trans *t, int nfld, const double *gp, double *ave, double *pmin, double *pmax);

/* SPECNORM (spnormd_mod.F90:49-50 + spnormc_mod.F90): per-field L2 norm. */
void orc_specnorm(const orc_trans *t, int nfld, const double *sp, double *norms);

/* FFTW-semantics helpers exposed for unit tests (naive DFT cross-check). */
void orc_fft_r2c(int n, const double *in, double *out /* 2*(n/2+1) */);
void orc_fft_c2r(int n, const double *in /* 2*(n/2+1) */, double *out);

#ifdef __cplusplus
}
#endif
#endif
