/* transi_mi.c -- see transi_mi.h.  Thin marshalling onto include/ectrans_mi.h. */
#include "transi_mi.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/ectrans_mi.h"

static int g_limit = 100, g_limit_set = 0;
static double g_radius = 6371.22e3;
static int g_radius_set = 0;
static int g_init = 0;
static char g_msg[1200];

const char *trans_error_msg(int errcode) {
  switch (errcode) {
    case TRANS_SUCCESS: return "Trans: No error";
    case TRANS_NOTIMPL: return "Trans: Not (yet) implemented";
    case TRANS_MISSING_ARG: return "Trans: Required member of the argument structure is missing or not allocated";
    case TRANS_UNRECOGNIZED_ARG: return "Trans: Unrecognized argument";
    case TRANS_STALE_ARG: return "Trans: Passed argument was already used in a previous call";
    default:
      strcpy(g_msg, "Trans: ");
      strncat(g_msg, emi_last_error(), sizeof(g_msg) - 16);
      return g_msg;
  }
}

/* Several tasks: the host attaches a transport first (emi_mpi_attach of ectrans_amd/mpi, emi_rccl_attach of
 * ectrans_amd/rccl: they call emi_init with the task count and number and register the exchange and the host
 * collectives); trans_init then adopts what it finds, as the Fortran shim's SETUP_TRANS0 does.  trans_use_mpi(true) is
 * therefore only accepted once such a transport is attached (the reference: transi.h:177, MPL_INIT inside trans_init). */
int trans_use_mpi(_bool b) {
  int np = 0, me = 0;
  if (!b) return TRANS_SUCCESS;
  return emi_inq_tasks(&np, &me) == 0 ? TRANS_SUCCESS : TRANS_NOTIMPL;
}
/* NPRTRV / NPRGPEW (transi.h:144,156): the decomposition of this library is NPRTRW x 1; LEQ_REGIONS (transi.h:167)
 * has no effect on it (the grid-point distribution equals the latitude bands). */
int trans_set_nprtrv(int n) { return n == 1 ? TRANS_SUCCESS : TRANS_NOTIMPL; }
int trans_set_nprgpew(int n) { return n == 1 ? TRANS_SUCCESS : TRANS_NOTIMPL; }
int trans_set_leq_regions(_bool b) {
  (void)b;
  return TRANS_SUCCESS;
}
int trans_set_handles_limit(int n) {
  if (g_init) return TRANS_ERROR;
  g_limit = n;
  g_limit_set = 1;
  return TRANS_SUCCESS;
}
int trans_set_radius(double r) {
  if (g_init) return TRANS_ERROR;
  g_radius = r;
  g_radius_set = 1;
  return TRANS_SUCCESS;
}

int trans_init(void) {
  if (g_init) return TRANS_SUCCESS;
  int np = 0, me = 0;
  if (emi_inq_tasks(&np, &me) == 0) { /* a transport (or the host) has initialised the library: adopt its tasks */
    /* ... but not silently its handle limit and planet radius: trans_set_handles_limit / trans_set_radius of this host must
     * agree with what the transport passed to emi_init (transi's own default radius is 6371.22e3, transi_module.F90:129-136) */
    int lim = 0;
    double rad = 0.0;
    if (emi_inq_init(&lim, &rad) != 0) return TRANS_ERROR;
    if ((g_limit_set && g_limit > lim) || (g_radius_set && g_radius != rad)) {
      fprintf(stderr, "trans_init: the attached transport initialised the library with kmax_resol %d, radius %.17g; this host asks for %d, %.17g\n",
              lim, rad, g_limit, g_radius);
      return TRANS_ERROR;
    }
    g_init = 1;
    return TRANS_SUCCESS;
  }
  emi_init_t cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.kmax_resol = g_limit;
  cfg.prad = g_radius;
  cfg.nproc = 1;
  cfg.myproc = 1;
  cfg.device = -1;
  if (emi_init(&cfg) != 0) return TRANS_ERROR;
  g_init = 1;
  return TRANS_SUCCESS;
}

int trans_new(struct Trans_t *t) {
  memset(t, 0, sizeof(*t));
  t->nsmax = -1;
  t->lsplit = 1;
  t->flt = -1;
  return TRANS_SUCCESS;
}

int trans_set_resol(struct Trans_t *t, int ndgl, const int *nloen) {
  t->ndgl = ndgl;
  free(t->nloen);
  t->nloen = (int *)malloc(sizeof(int) * (size_t)ndgl);
  if (!t->nloen) return TRANS_ERROR;
  memcpy(t->nloen, nloen, sizeof(int) * (size_t)ndgl);
  return TRANS_SUCCESS;
}

int trans_set_trunc(struct Trans_t *t, int nsmax) {
  t->nsmax = nsmax;
  return TRANS_SUCCESS;
}

static int set_path(char **dst, const char *path) {
  free(*dst);
  *dst = NULL;
  if (!path) return TRANS_MISSING_ARG;
  *dst = (char *)malloc(strlen(path) + 1);
  if (!*dst) return TRANS_ERROR;
  strcpy(*dst, path);
  return TRANS_SUCCESS;
}

int trans_set_read(struct Trans_t *t, const char *filepath) { return set_path(&t->readfp, filepath); }

int trans_set_write(struct Trans_t *t, const char *filepath) { return set_path(&t->writefp, filepath); }

int trans_set_cache(struct Trans_t *t, const void *cache, size_t cachesize) {
  t->cache = cache;
  t->cachesize = cachesize;
  return TRANS_SUCCESS;
}

int trans_setup(struct Trans_t *t) {
  int rc = trans_init();
  if (rc) return rc;
  if (t->ndgl <= 0 || (!t->nloen && t->nlon <= 0)) return TRANS_MISSING_ARG;
  if (t->llatlon || t->flt > 0) return TRANS_NOTIMPL;
  if (t->nsmax < 0) t->nsmax = t->ndgl - 1; /* default: linear truncation on the given latitudes */
  emi_setup_t cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.ksmax = t->nsmax;
  cfg.kdgl = t->ndgl;
  cfg.kloen = t->nloen;
  cfg.kdlon = t->nlon;
  cfg.precision = 8;
  emi_legpol_io_t io;
  memset(&io, 0, sizeof(io));
  if (t->cachesize > 0 && !t->cache) { /* transi_module.F90:737-741 */
    fprintf(stderr, "Cache memory was not allocated\n");
    return TRANS_MISSING_ARG;
  }
  if (t->readfp && t->readfp[0]) {
    io.io = "readf";
    io.fname = t->readfp;
  } else if (t->writefp && t->writefp[0]) {
    io.io = "writef";
    io.fname = t->writefp;
  } else if (t->cachesize > 0) {
    io.io = "membuf";
    io.ptr = t->cache;
    io.len = t->cachesize;
  }
  if (emi_setup_legpol(&cfg, io.io ? &io : NULL, &t->handle) != 0) return TRANS_ERROR;
  t->myproc = t->nproc = 1;
  emi_inq_tasks(&t->nproc, &t->myproc);
  emi_inq_int(t->handle, "nspec2", &t->nspec2);
  t->nspec = t->nspec2 / 2;
  emi_inq_int(t->handle, "nspec2g", &t->nspec2g);
  emi_inq_int(t->handle, "nspec2mx", &t->nspec2mx);
  emi_inq_int(t->handle, "nump", &t->nump);
  emi_inq_int(t->handle, "ngptot", &t->ngptot);
  emi_inq_int(t->handle, "ngptotg", &t->ngptotg);
  emi_inq_int(t->handle, "ngptotmx", &t->ngptotmx);
  return TRANS_SUCCESS;
}

static int inq_one(struct Trans_t *t, const char *v) {
  const int ns = t->nsmax + 1;
  if (!strcmp(v, "rmu") || !strcmp(v, "rgw")) {
    double **dst = !strcmp(v, "rmu") ? &t->rmu : &t->rgw;
    if (!*dst) *dst = (double *)malloc(sizeof(double) * (size_t)t->ndgl);
    return emi_inq_real_array(t->handle, v, *dst, t->ndgl) ? TRANS_ERROR : TRANS_SUCCESS;
  }
  struct {
    const char *name, *emi;
    int **dst;
    int len;
  } ints[] = {{"nasm0", "nasm0", &t->nasm0, ns}, {"nmyms", "myms", &t->nmyms, ns}, {"ndglu", "ndglu", &t->ndglu, ns},
              {"nnmeng", "nmen", &t->nnmeng, t->ndgl}, {"nmeng", "nmen", &t->nnmeng, t->ndgl}};
  for (size_t i = 0; i < sizeof(ints) / sizeof(ints[0]); i++)
    if (!strcmp(v, ints[i].name)) {
      if (!*ints[i].dst) *ints[i].dst = (int *)malloc(sizeof(int) * (size_t)ints[i].len);
      return emi_inq_int_array(t->handle, ints[i].emi, *ints[i].dst, ints[i].len) ? TRANS_ERROR : TRANS_SUCCESS;
    }
  if (!strcmp(v, "nvalue")) {
    if (!t->nvalue) t->nvalue = (int *)malloc(sizeof(int) * (size_t)t->nspec2);
    int k = 0;
    for (int m = 0; m <= t->nsmax; m++)
      for (int n = m; n <= t->nsmax; n++) {
        t->nvalue[k++] = n;
        t->nvalue[k++] = n;
      }
    return TRANS_SUCCESS;
  }
  if (!strcmp(v, "nloen")) return TRANS_SUCCESS; /* input, already there */
  return TRANS_UNRECOGNIZED_ARG;
}

int trans_inquire(struct Trans_t *t, const char *varlist) {
  if (!t || !t->handle) return TRANS_MISSING_ARG;
  char buf[512];
  strncpy(buf, varlist, sizeof(buf) - 1);
  buf[sizeof(buf) - 1] = 0;
  for (char *tok = strtok(buf, ", "); tok; tok = strtok(NULL, ", ")) {
    int rc = inq_one(t, tok);
    if (rc) return rc;
  }
  return TRANS_SUCCESS;
}

/* lglobal: rgp is the global field [nfld][ngptotg]; valid with one task only (transi_module.F90:1514-1527) */
static int check_global(const struct Trans_t *t, int lglobal) {
  if (lglobal && t->nproc != 1) {
    fprintf(stderr, "assert_global: ERROR: Configuration only valid for nproc == 1\n");
    return TRANS_ERROR;
  }
  return TRANS_SUCCESS;
}

/* A transi call returns when its results are there (transi has no stream argument): calls on device-resident arrays are queued
 * asynchronously by the library, so wait for the resolution's last call here.  Host-array calls have already synchronised. */
static int run_and_wait(int rc, int handle) {
  if (rc) return TRANS_ERROR;
  return emi_wait(handle) ? TRANS_ERROR : TRANS_SUCCESS;
}

struct DirTrans_t new_dirtrans(struct Trans_t *t) {
  struct DirTrans_t d;
  memset(&d, 0, sizeof(d));
  d.trans = t;
  return d;
}

int trans_dirtrans(struct DirTrans_t *d) {
  if (d->count++ > 0) return TRANS_STALE_ARG;
  if (!d->trans || !d->rgp) return TRANS_MISSING_ARG;
  if (d->nscalar > 0 && !d->rspscalar) return TRANS_MISSING_ARG;
  if (d->nvordiv > 0 && (!d->rspvor || !d->rspdiv)) return TRANS_MISSING_ARG;
  if (d->rmeanu || d->rmeanv) return TRANS_NOTIMPL; /* LAM only */
  emi_dirtrans_t a;
  memset(&a, 0, sizeof(a));
  a.mem_space = EMI_MEM_AUTO; /* rgp / rsp* in device memory are used in place, host arrays are staged (emi_ptr_space) */
  if (d->nvordiv > 0) {
    a.spvor = d->rspvor;
    a.spdiv = d->rspdiv;
    a.nf_uv = d->nvordiv;
  }
  if (d->nscalar > 0) {
    a.spscalar = d->rspscalar;
    a.nf_scalar = d->nscalar;
  }
  /* lglobal: rgp is the global field [nfld][ngptotg] (transi_module.F90:1721-1728): one task, so global == local and
   * the call is the unblocked one */
  if (check_global(d->trans, d->lglobal)) return TRANS_ERROR;
  a.kproma = (d->nproma > 0 && !d->lglobal) ? d->nproma : d->trans->ngptot;
  a.gp = d->rgp;
  a.gp_nfld = 2 * d->nvordiv + d->nscalar;
  return run_and_wait(emi_dir_trans(d->trans->handle, &a), d->trans->handle);
}

struct InvTrans_t new_invtrans(struct Trans_t *t) {
  struct InvTrans_t v;
  memset(&v, 0, sizeof(v));
  v.trans = t;
  return v;
}

int trans_invtrans(struct InvTrans_t *v) {
  if (v->count++ > 0) return TRANS_STALE_ARG;
  if (!v->trans || !v->rgp) return TRANS_MISSING_ARG;
  if (v->nscalar > 0 && !v->rspscalar) return TRANS_MISSING_ARG;
  if (v->nvordiv > 0 && (!v->rspvor || !v->rspdiv)) return TRANS_MISSING_ARG;
  if (v->rmeanu || v->rmeanv) return TRANS_NOTIMPL; /* LAM only */
  emi_invtrans_t a;
  memset(&a, 0, sizeof(a));
  a.mem_space = EMI_MEM_AUTO; /* rgp / rsp* in device memory are used in place, host arrays are staged (emi_ptr_space) */
  if (v->nvordiv > 0) {
    a.spvor = v->rspvor;
    a.spdiv = v->rspdiv;
    a.nf_uv = v->nvordiv;
  }
  if (v->nscalar > 0) {
    a.spscalar = v->rspscalar;
    a.nf_scalar = v->nscalar;
  }
  a.ldscders = v->lscalarders;
  a.lduvder = v->luvder_EW;
  a.ldvorgp = a.lddivgp = v->lvordivgp;
  if (check_global(v->trans, v->lglobal)) return TRANS_ERROR;
  a.kproma = (v->nproma > 0 && !v->lglobal) ? v->nproma : v->trans->ngptot; /* lglobal: as trans_dirtrans */
  a.gp = v->rgp;
  a.gp_nfld = 2 * v->nvordiv + v->nscalar + (v->lscalarders ? 2 * v->nscalar : 0) + (v->lvordivgp ? 2 * v->nvordiv : 0) +
              (v->luvder_EW ? 2 * v->nvordiv : 0);
  return run_and_wait(emi_inv_trans(v->trans->handle, &a), v->trans->handle);
}

/* ---- adjoints ---- */
struct DirTransAdj_t new_dirtrans_adj(struct Trans_t *t) {
  struct DirTransAdj_t d;
  memset(&d, 0, sizeof(d));
  d.trans = t;
  return d;
}
int trans_dirtrans_adj(struct DirTransAdj_t *d) {
  if (d->count++ > 0) return TRANS_STALE_ARG;
  if (!d->trans || !d->rgp) return TRANS_MISSING_ARG;
  if (d->nscalar > 0 && !d->rspscalar) return TRANS_MISSING_ARG;
  if (d->nvordiv > 0 && (!d->rspvor || !d->rspdiv)) return TRANS_MISSING_ARG;
  if (d->rmeanu || d->rmeanv) return TRANS_NOTIMPL;
  emi_dirtrans_t a;
  memset(&a, 0, sizeof(a));
  a.mem_space = EMI_MEM_AUTO; /* rgp / rsp* in device memory are used in place, host arrays are staged (emi_ptr_space) */
  if (d->nvordiv > 0) a.spvor = d->rspvor, a.spdiv = d->rspdiv, a.nf_uv = d->nvordiv;
  if (d->nscalar > 0) a.spscalar = d->rspscalar, a.nf_scalar = d->nscalar;
  if (check_global(d->trans, d->lglobal)) return TRANS_ERROR;
  a.kproma = (d->nproma > 0 && !d->lglobal) ? d->nproma : d->trans->ngptot;
  a.gp = d->rgp; /* written */
  a.gp_nfld = 2 * d->nvordiv + d->nscalar;
  return run_and_wait(emi_dir_transad(d->trans->handle, &a), d->trans->handle);
}
struct InvTransAdj_t new_invtrans_adj(struct Trans_t *t) {
  struct InvTransAdj_t v;
  memset(&v, 0, sizeof(v));
  v.trans = t;
  return v;
}
int trans_invtrans_adj(struct InvTransAdj_t *v) {
  if (v->count++ > 0) return TRANS_STALE_ARG;
  if (!v->trans || !v->rgp) return TRANS_MISSING_ARG;
  if (v->nscalar > 0 && !v->rspscalar) return TRANS_MISSING_ARG;
  if (v->nvordiv > 0 && (!v->rspvor || !v->rspdiv)) return TRANS_MISSING_ARG;
  if (v->rmeanu || v->rmeanv) return TRANS_NOTIMPL;
  if (check_global(v->trans, v->lglobal)) return TRANS_ERROR;
  emi_invtrans_t a;
  memset(&a, 0, sizeof(a));
  a.mem_space = EMI_MEM_AUTO; /* rgp / rsp* in device memory are used in place, host arrays are staged (emi_ptr_space) */
  if (v->nvordiv > 0) a.spvor = v->rspvor, a.spdiv = v->rspdiv, a.nf_uv = v->nvordiv; /* written */
  if (v->nscalar > 0) a.spscalar = v->rspscalar, a.nf_scalar = v->nscalar;
  a.ldscders = v->lscalarders; /* the extra grid fields are further inputs (inv_transad.h) */
  a.lduvder = v->luvder_EW;
  a.ldvorgp = a.lddivgp = v->lvordivgp;
  a.kproma = (v->nproma > 0 && !v->lglobal) ? v->nproma : v->trans->ngptot;
  a.gp = v->rgp;
  a.gp_nfld = 2 * v->nvordiv + v->nscalar + (v->lscalarders ? 2 * v->nscalar : 0) + (v->lvordivgp ? 2 * v->nvordiv : 0) +
              (v->luvder_EW ? 2 * v->nvordiv : 0);
  return run_and_wait(emi_inv_transad(v->trans->handle, &a), v->trans->handle);
}

/* ---- global <-> distributed arrays: DIST_GRID / GATH_GRID / DIST_SPEC / GATH_SPEC of the C-ABI, any task count
 * (transi.h:499-616; nfrom / nto: the task of every field, 1-based; rgpg / rspecg hold the fields of THIS task) ---- */
struct DistGrid_t new_distgrid(struct Trans_t *t) {
  struct DistGrid_t a;
  memset(&a, 0, sizeof(a));
  a.trans = t;
  a.nproma = t->ngptot;
  a.ngpblks = 1;
  return a;
}
struct GathGrid_t new_gathgrid(struct Trans_t *t) {
  struct GathGrid_t a;
  memset(&a, 0, sizeof(a));
  a.trans = t;
  a.nproma = t->ngptot;
  a.ngpblks = 1;
  return a;
}
struct DistSpec_t new_distspec(struct Trans_t *t) {
  struct DistSpec_t a;
  memset(&a, 0, sizeof(a));
  a.trans = t;
  return a;
}
struct GathSpec_t new_gathspec(struct Trans_t *t) {
  struct GathSpec_t a;
  memset(&a, 0, sizeof(a));
  a.trans = t;
  return a;
}
int trans_distgrid(struct DistGrid_t *a) {
  if (a->count++ > 0) return TRANS_STALE_ARG;
  if (!a->trans || !a->rgp || !a->nfrom || a->nfld <= 0 || a->nproma <= 0) return TRANS_MISSING_ARG;
  const long ng = a->trans->ngptot, np = a->nproma, nb = (ng - 1) / np + 1;
  if (a->ngpblks < nb) return TRANS_ERROR;
  if (emi_dist_grid(a->trans->handle, a->rgpg, a->nfld, a->nfrom, NULL, a->nproma, a->rgp)) return TRANS_ERROR;
  /* the padding of the last NPROMA block (and any further block): zero, as the Fortran DIST_GRID of the shim leaves it -- a
   * malloc'ed rgp must not carry NaNs into a later trans_dirtrans or a checksum */
  for (long b = nb - 1; b < a->ngpblks; b++) {
    const long first = b == nb - 1 ? ng - (nb - 1) * np : 0;
    for (int f = 0; f < a->nfld; f++) memset(a->rgp + ((size_t)b * a->nfld + f) * np + first, 0, sizeof(double) * (size_t)(np - first));
  }
  return TRANS_SUCCESS;
}
int trans_gathgrid(struct GathGrid_t *a) {
  if (a->count++ > 0) return TRANS_STALE_ARG;
  if (!a->trans || !a->rgp || !a->nto || a->nfld <= 0 || a->nproma <= 0) return TRANS_MISSING_ARG;
  return emi_gath_grid(a->trans->handle, a->rgpg, a->nfld, a->nto, a->nproma, a->rgp) ? TRANS_ERROR : TRANS_SUCCESS;
}
/* rspecg is [nspec2g][nfldg] here (Fortran PSPECG(nfldg, nspec2g), transi.h:1140-1186), [nfldg][nspec2g] in the C-ABI */
static int count_mine(const struct Trans_t *t, const int *task, int nfld) {
  int n = 0;
  for (int f = 0; f < nfld; f++) n += task[f] == t->myproc;
  return n;
}
int trans_distspec(struct DistSpec_t *a) {
  if (a->count++ > 0) return TRANS_STALE_ARG;
  if (!a->trans || !a->rspec || !a->nfrom || a->nfld <= 0) return TRANS_MISSING_ARG;
  const int nm = count_mine(a->trans, a->nfrom, a->nfld);
  const size_t ng = (size_t)a->trans->nspec2g;
  double *tmp = NULL;
  if (nm > 0) {
    if (!a->rspecg) return TRANS_MISSING_ARG;
    tmp = (double *)malloc(sizeof(double) * ng * (size_t)nm);
    if (!tmp) return TRANS_ERROR;
    for (size_t i = 0; i < ng; i++)
      for (int f = 0; f < nm; f++) tmp[(size_t)f * ng + i] = a->rspecg[i * (size_t)nm + f];
  }
  const int rc = emi_dist_spec(a->trans->handle, tmp, a->nfld, a->nfrom, NULL, a->rspec);
  free(tmp);
  return rc ? TRANS_ERROR : TRANS_SUCCESS;
}
int trans_gathspec(struct GathSpec_t *a) {
  if (a->count++ > 0) return TRANS_STALE_ARG;
  if (!a->trans || !a->rspec || !a->nto || a->nfld <= 0) return TRANS_MISSING_ARG;
  const int nm = count_mine(a->trans, a->nto, a->nfld);
  const size_t ng = (size_t)a->trans->nspec2g;
  double *tmp = NULL;
  if (nm > 0) {
    if (!a->rspecg) return TRANS_MISSING_ARG;
    tmp = (double *)malloc(sizeof(double) * ng * (size_t)nm);
    if (!tmp) return TRANS_ERROR;
  }
  const int rc = emi_gath_spec(a->trans->handle, tmp, a->nfld, a->nto, a->rspec);
  if (!rc)
    for (size_t i = 0; i < ng; i++)
      for (int f = 0; f < nm; f++) a->rspecg[i * (size_t)nm + f] = tmp[(size_t)f * ng + i];
  free(tmp);
  return rc ? TRANS_ERROR : TRANS_SUCCESS;
}

struct SpecNorm_t new_specnorm(struct Trans_t *t) {
  struct SpecNorm_t s;
  memset(&s, 0, sizeof(s));
  s.trans = t;
  s.nmaster = 1;
  return s;
}

int trans_specnorm(struct SpecNorm_t *s) {
  if (s->count++ > 0) return TRANS_STALE_ARG;
  if (!s->trans || !s->rspec || !s->rnorm || s->nfld <= 0) return TRANS_MISSING_ARG;
  if (s->rmet) return TRANS_NOTIMPL;
  return emi_specnorm(s->trans->handle, EMI_MEM_AUTO, s->rspec, s->nfld, s->rnorm) ? TRANS_ERROR : TRANS_SUCCESS;
}

struct VorDivToUV_t new_vordiv_to_UV(void) {
  struct VorDivToUV_t v;
  memset(&v, 0, sizeof(v));
  return v;
}
/* transi_module.F90:2663-2740: the same argument checks, then VORDIV_TO_UV(RSPVOR, RSPDIV, RSPU, RSPV, NSMAX) */
int trans_vordiv_to_UV(struct VorDivToUV_t *v) {
  if (v->count++ > 0) return TRANS_STALE_ARG;
  if (v->ncoeff == 0 || v->nsmax == 0) return TRANS_MISSING_ARG;
  if (!v->rspvor || !v->rspdiv || !v->rspu || !v->rspv) return TRANS_MISSING_ARG;
  if (!g_init) {
    const int rc = trans_init();
    if (rc) return rc;
  }
  return emi_vordiv_to_uv(v->nsmax, 8, EMI_MEM_AUTO, v->rspvor, v->rspdiv, v->rspu, v->rspv, v->nfld, v->ncoeff) ? TRANS_ERROR : TRANS_SUCCESS;
}

int trans_delete(struct Trans_t *t) {
  int rc = TRANS_SUCCESS;
  if (t->handle) rc = emi_release(t->handle) ? TRANS_ERROR : TRANS_SUCCESS;
  free(t->readfp), free(t->writefp);
  free(t->nloen), free(t->nmyms), free(t->nasm0), free(t->nvalue), free(t->ndglu), free(t->nnmeng), free(t->rmu), free(t->rgw);
  memset(t, 0, sizeof(*t));
  return rc;
}

int trans_finalize(void) {
  g_init = 0;
  return emi_finalize() ? TRANS_ERROR : TRANS_SUCCESS;
}
