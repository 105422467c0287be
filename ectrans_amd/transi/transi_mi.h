/*
 * transi_mi.h -- transi-style C API (struct Trans_t / DirTrans_t / InvTrans_t / SpecNorm_t and
 * trans_* functions) over libectrans_mi.so, source compatible with the hot-path subset of the
 * reference's C interface (/root/reference/src/transi/transi.h:110-685 functions, :701-1240
 * structs).  Same names, same field names, same array layouts
 * (rgp[ngpblks][nfld][nproma], rsp[nspec2][nfld], transi.h:908-921), same return convention
 * (TRANS_SUCCESS = 0, negative error codes, trans_error_msg()).
 *
 * Not provided (outside SURVEY.md section 8): vordiv_to_UV,
 * LAM, lonlat, rmeanu/rmeanv (lglobal is honoured: one task, global == local).  They return
 * TRANS_NOTIMPL instead of being silently ignored.
 */
#ifndef TRANSI_MI_H
#define TRANSI_MI_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef int _bool;

#define TRANS_SUCCESS 0
#define TRANS_ERROR (-1)
#define TRANS_NOTIMPL (-2)
#define TRANS_MISSING_ARG (-3)
#define TRANS_UNRECOGNIZED_ARG (-4)
#define TRANS_STALE_ARG (-5)

struct Trans_t {
  /* inputs */
  int ndgl;   /* number of latitudes */
  int *nloen; /* [ndgl] points per latitude */
  int nlon;   /* points per latitude of a regular grid (when nloen == NULL) */
  int nsmax;  /* spectral truncation */
  _bool lsplit;
  int llatlon; /* must stay 0 */
  int flt;     /* <= 0: Fast Legendre Transform not requested (the only supported setting) */
  /* Legendre polynomials from / to a file or a memory image (transi.h:719-722; set with trans_set_read,
   * trans_set_write, trans_set_cache; precedence read > write > cache as transi_module.F90:762-866) */
  char *readfp;
  char *writefp;
  const void *cache;
  size_t cachesize;
  /* filled by trans_setup */
  int myproc, nproc;
  int handle; /* KRESOL */
  int nspec, nspec2, nspec2g, nspec2mx, nump;
  int ngptot, ngptotg, ngptotmx;
  /* filled on demand by trans_inquire("name,name,...") */
  int *nmyms;   /* [nump] */
  int *nasm0;   /* [0:nsmax], 1-based addresses as in Fortran */
  int *nvalue;  /* [nspec2] n of every coefficient */
  int *ndglu;   /* [0:nsmax] */
  int *nnmeng;  /* [ndgl] NMEN per latitude */
  double *rmu;  /* [ndgl] sin(latitude) */
  double *rgw;  /* [ndgl] Gaussian weights (sum = 1) */
};

struct DirTrans_t {
  const double *rgp;  /* [ngpblks][2*nvordiv+nscalar][nproma]: u, v, scalars */
  double *rspscalar;  /* [nspec2][nscalar] */
  double *rspvor;     /* [nspec2][nvordiv] */
  double *rspdiv;     /* [nspec2][nvordiv] */
  const double *rmeanu, *rmeanv; /* LAM only: must stay NULL */
  int nproma, nscalar, nvordiv, ngpblks, lglobal;
  struct Trans_t *trans;
  int count; /* a DirTrans_t is single use, as in the reference (transi_module.F90:1661-1666) */
};

struct InvTrans_t {
  const double *rspscalar, *rspvor, *rspdiv;
  const double *rmeanu, *rmeanv;
  double *rgp; /* [ngpblks][nfld][nproma]: [vor div] u v scalars [NS-ders] [u_EW v_EW] [sc_EW] */
  int nproma, nscalar, nvordiv;
  int lscalarders, luvder_EW, lvordivgp;
  int ngpblks, lglobal;
  struct Trans_t *trans;
  int count;
};

/* Adjoints (transi.h:946-1076): the structs repeat the field declarations of DirTrans_t / InvTrans_t,
 * as the reference does, although the data flows the other way: trans_dirtrans_adj READS rsp* and
 * WRITES rgp (DIR_TRANSAD, transi_module.F90), trans_invtrans_adj reads rgp and writes rsp* (INV_TRANSAD;
 * lscalarders / luvder_EW / lvordivgp must be 0). */
struct DirTransAdj_t {
  const double *rgp;
  double *rspscalar, *rspvor, *rspdiv;
  const double *rmeanu, *rmeanv;
  int nproma, nscalar, nvordiv, ngpblks, lglobal;
  struct Trans_t *trans;
  int count;
};
struct InvTransAdj_t {
  const double *rspscalar, *rspvor, *rspdiv;
  const double *rmeanu, *rmeanv;
  double *rgp;
  int nproma, nscalar, nvordiv;
  int lscalarders, luvder_EW, lvordivgp;
  int ngpblks, lglobal;
  struct Trans_t *trans;
  int count;
};
struct DirTransAdj_t new_dirtrans_adj(struct Trans_t *);
int trans_dirtrans_adj(struct DirTransAdj_t *);
struct InvTransAdj_t new_invtrans_adj(struct Trans_t *);
int trans_invtrans_adj(struct InvTransAdj_t *);

struct SpecNorm_t {
  const double *rspec; /* [nspec2][nfld] */
  int nmaster;
  const double *rmet; /* must stay NULL */
  double *rnorm;      /* [nfld] */
  int nfld;
  struct Trans_t *trans;
  int count;
};

const char *trans_error_msg(int errcode);
int trans_use_mpi(_bool);          /* true: only once a transport is attached (ectrans_amd/mpi, ectrans_amd/rccl) */
int trans_set_nprtrv(int);         /* 1 (transi.h:144) */
int trans_set_nprgpew(int);        /* 1 (transi.h:156) */
int trans_set_leq_regions(_bool);  /* accepted, no effect on this decomposition (transi.h:167) */
int trans_set_handles_limit(int);  /* default 100 (transi_module.F90:129-136) */
int trans_set_radius(double);      /* default 6371.22e3 (transi's own default) */
int trans_init(void);
int trans_new(struct Trans_t *);
int trans_set_resol(struct Trans_t *, int ndgl, const int *nloen);
int trans_set_trunc(struct Trans_t *, int nsmax);
int trans_set_read(struct Trans_t *, const char *filepath);          /* transi.h:192 */
int trans_set_write(struct Trans_t *, const char *filepath);         /* transi.h:193 */
int trans_set_cache(struct Trans_t *, const void *cache, size_t cachesize); /* transi.h:194 */
int trans_setup(struct Trans_t *);
int trans_inquire(struct Trans_t *, const char *varlist);
struct DirTrans_t new_dirtrans(struct Trans_t *);
int trans_dirtrans(struct DirTrans_t *);
struct InvTrans_t new_invtrans(struct Trans_t *);
int trans_invtrans(struct InvTrans_t *);
/* trans_distgrid / trans_gathgrid / trans_distspec / trans_gathspec (transi.h:1082-1186): global <->
 * distributed arrays over DIST_GRID / GATH_GRID / DIST_SPEC / GATH_SPEC of the C-ABI, any task count: nfrom / nto name
 * the task (1-based) of every field; rgpg / rspecg hold the fields of THIS task, in field order. */
struct DistGrid_t {
  const double *rgpg; /* [nfld][ngptotg] */
  double *rgp;        /* [ngpblks][nfld][nproma] */
  const int *nfrom;   /* [nfld] */
  int nproma, nfld, ngpblks;
  struct Trans_t *trans;
  int count;
};
struct GathGrid_t {
  double *rgpg;
  const double *rgp;
  const int *nto;
  int nproma, nfld, ngpblks;
  struct Trans_t *trans;
  int count;
};
struct DistSpec_t {
  const double *rspecg; /* [nspec2g][nfld] */
  double *rspec;        /* [nspec2][nfld] */
  const int *nfrom;
  int nfld;
  struct Trans_t *trans;
  int count;
};
struct GathSpec_t {
  double *rspecg;
  const double *rspec;
  const int *nto;
  int nfld;
  struct Trans_t *trans;
  int count;
};
struct DistGrid_t new_distgrid(struct Trans_t *);
int trans_distgrid(struct DistGrid_t *);
struct GathGrid_t new_gathgrid(struct Trans_t *);
int trans_gathgrid(struct GathGrid_t *);
struct DistSpec_t new_distspec(struct Trans_t *);
int trans_distspec(struct DistSpec_t *);
struct GathSpec_t new_gathspec(struct Trans_t *);
int trans_gathspec(struct GathSpec_t *);
/* trans_vordiv_to_UV (src/transi/transi.h:620-648, 1189-1217): spectral vorticity / divergence -> spectral U, V (u cos, v cos), local
 * arrays [ncoeff = nspec2][nfld]; no Trans_t handle, only the truncation */
struct VorDivToUV_t {
  const double *rspvor, *rspdiv;
  double *rspu, *rspv;
  int nfld, nsmax, ncoeff;
  int count;
};
struct VorDivToUV_t new_vordiv_to_UV(void);
int trans_vordiv_to_UV(struct VorDivToUV_t *);
struct SpecNorm_t new_specnorm(struct Trans_t *);
int trans_specnorm(struct SpecNorm_t *);
int trans_delete(struct Trans_t *);
int trans_finalize(void);

#ifdef __cplusplus
}
#endif
#endif
