/* C caller of the transi-style API, modelled on what the reference's transi_test_program.c checks
 * (tests/transi/transi_test_program.c:66-160: constant fields -> only coefficient 0 is non-zero)
 * plus a wind/scalar round trip.  Exit code 0 = pass. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../ectrans_amd/transi/transi_mi.h"

#define CHECK(x)                                                          \
  do {                                                                    \
    int rc_ = (x);                                                        \
    if (rc_ != TRANS_SUCCESS) {                                           \
      fprintf(stderr, "%s failed: %s\n", #x, trans_error_msg(rc_));       \
      return 1;                                                           \
    }                                                                     \
  } while (0)

int main(void) {
  const int nsmax = 31, ndgl = 2 * (nsmax + 1);
  int *nloen = malloc(sizeof(int) * ndgl);
  for (int i = 0; i <= nsmax; i++) nloen[i] = nloen[ndgl - 1 - i] = 20 + 4 * i;
  CHECK(trans_use_mpi(0));
  struct Trans_t trans;
  CHECK(trans_new(&trans));
  CHECK(trans_set_resol(&trans, ndgl, nloen));
  CHECK(trans_set_trunc(&trans, nsmax));
  CHECK(trans_setup(&trans));
  CHECK(trans_inquire(&trans, "rgw,rmu,nasm0,nvalue,nmyms"));
  double sw = 0;
  for (int j = 0; j < ndgl; j++) sw += trans.rgw[j];
  if (fabs(sw - 1.0) > 1e-10) return 2;
  const int nscalar = 4, nvordiv = 2, nfld = 2 * nvordiv + nscalar;
  double *rgp = malloc(sizeof(double) * nfld * trans.ngptot);
  double *rsc = malloc(sizeof(double) * nscalar * trans.nspec2);
  double *rvor = malloc(sizeof(double) * nvordiv * trans.nspec2);
  double *rdiv = malloc(sizeof(double) * nvordiv * trans.nspec2);
  /* constant fields 1..4 and zero wind */
  for (int f = 0; f < nfld; f++)
    for (int p = 0; p < trans.ngptot; p++) rgp[f * trans.ngptot + p] = f < 2 * nvordiv ? 0.0 : (double)(f - 2 * nvordiv + 1);
  struct DirTrans_t d = new_dirtrans(&trans);
  d.nscalar = nscalar, d.nvordiv = nvordiv, d.rgp = rgp, d.rspscalar = rsc, d.rspvor = rvor, d.rspdiv = rdiv;
  CHECK(trans_dirtrans(&d));
  if (trans_dirtrans(&d) != TRANS_STALE_ARG) return 3;
  for (int f = 0; f < nscalar; f++)
    for (int i = 0; i < trans.nspec2; i++) {
      double want = (i == 0) ? (double)(f + 1) : 0.0;
      if (fabs(rsc[i * nscalar + f] - want) > 1e-12) {
        fprintf(stderr, "coefficient %d of field %d = %g\n", i, f, rsc[i * nscalar + f]);
        return 4;
      }
    }
  /* harmonic (4,19) through inverse + direct */
  for (int i = 0; i < nscalar * trans.nspec2; i++) rsc[i] = 0;
  for (int i = 0; i < nvordiv * trans.nspec2; i++) rvor[i] = rdiv[i] = 0;
  int i419 = trans.nasm0[4] - 1 + 2 * (19 - 4);
  for (int f = 0; f < nscalar; f++) rsc[i419 * nscalar + f] = 1.0;
  for (int f = 0; f < nvordiv; f++) rvor[i419 * nvordiv + f] = rdiv[i419 * nvordiv + f] = 1.0;
  struct InvTrans_t v = new_invtrans(&trans);
  v.nscalar = nscalar, v.nvordiv = nvordiv, v.rspscalar = rsc, v.rspvor = rvor, v.rspdiv = rdiv, v.rgp = rgp;
  CHECK(trans_invtrans(&v));
  struct DirTrans_t d2 = new_dirtrans(&trans);
  d2.nscalar = nscalar, d2.nvordiv = nvordiv, d2.rgp = rgp, d2.rspscalar = rsc, d2.rspvor = rvor, d2.rspdiv = rdiv;
  CHECK(trans_dirtrans(&d2));
  double norm[4];
  struct SpecNorm_t s = new_specnorm(&trans);
  s.rspec = rsc, s.nfld = nscalar, s.rnorm = norm;
  CHECK(trans_specnorm(&s));
  for (int f = 0; f < nscalar; f++)
    if (fabs(norm[f] / sqrt(2.0) - 1.0) > 1e-13 || fabs(rsc[i419 * nscalar + f] - 1.0) > 1e-13) return 5;
  for (int f = 0; f < nvordiv; f++)
    if (fabs(rvor[i419 * nvordiv + f] - 1.0) > 1e-12 || fabs(rdiv[i419 * nvordiv + f] - 1.0) > 1e-12) return 6;
  /* global <-> distributed helpers and lglobal (transi.h:512-616, 929): blocked copy and back */
  {
    const int nf = 2 * nvordiv + nscalar, np = 1000, nb = (trans.ngptot - 1) / np + 1;
    double *rgpg = malloc(sizeof(double) * (size_t)nf * trans.ngptot), *rblk = calloc((size_t)nb * nf * np, sizeof(double));
    int *one = malloc(sizeof(int) * nf);
    for (int f = 0; f < nf; f++) one[f] = 1;
    struct GathGrid_t gg = new_gathgrid(&trans);
    gg.rgpg = rgpg, gg.rgp = rgp, gg.nto = one, gg.nfld = nf;
    CHECK(trans_gathgrid(&gg));
    for (long i = 0; i < (long)nf * trans.ngptot; i++)
      if (rgpg[i] != rgp[i]) return 7; /* unblocked local == global for one task */
    struct DistGrid_t dg = new_distgrid(&trans);
    dg.rgpg = rgpg, dg.rgp = rblk, dg.nfrom = one, dg.nfld = nf, dg.nproma = np, dg.ngpblks = nb;
    CHECK(trans_distgrid(&dg));
    struct DirTrans_t d3 = new_dirtrans(&trans);  /* blocked input */
    d3.nscalar = nscalar, d3.nvordiv = nvordiv, d3.rgp = rblk, d3.nproma = np, d3.ngpblks = nb;
    d3.rspscalar = rsc, d3.rspvor = rvor, d3.rspdiv = rdiv;
    CHECK(trans_dirtrans(&d3));
    for (int f = 0; f < nscalar; f++)
      if (fabs(rsc[i419 * nscalar + f] - 1.0) > 1e-13) return 8;
    struct DirTrans_t d4 = new_dirtrans(&trans);  /* lglobal: the global field as is */
    d4.nscalar = nscalar, d4.nvordiv = nvordiv, d4.rgp = rgpg, d4.lglobal = 1;
    d4.rspscalar = rsc, d4.rspvor = rvor, d4.rspdiv = rdiv;
    CHECK(trans_dirtrans(&d4));
    for (int f = 0; f < nscalar; f++)
      if (fabs(rsc[i419 * nscalar + f] - 1.0) > 1e-13) return 9;
    double *rspg = malloc(sizeof(double) * (size_t)trans.nspec2 * nscalar);
    struct GathSpec_t gs = new_gathspec(&trans);
    gs.rspecg = rspg, gs.rspec = rsc, gs.nto = one, gs.nfld = nscalar;
    CHECK(trans_gathspec(&gs));
    struct DistSpec_t ds = new_distspec(&trans);
    ds.rspecg = rspg, ds.rspec = rsc, ds.nfrom = one, ds.nfld = nscalar;
    CHECK(trans_distspec(&ds));
    if (rspg[i419 * nscalar] != rsc[i419 * nscalar]) return 10;
    one[0] = 2;
    struct GathSpec_t bad = new_gathspec(&trans);
    bad.rspecg = rspg, bad.rspec = rsc, bad.nto = one, bad.nfld = nscalar;
    if (trans_gathspec(&bad) == TRANS_SUCCESS) return 11; /* no task 2 */
    free(rgpg), free(rblk), free(one), free(rspg);
  }
  /* adjoints (transi.h:299-352): <dirtrans(invtrans(x)), y> = <x, invtrans_adj(dirtrans_adj(y))> for the
   * spectral inner product with weights 1 (m = 0) and 2 (m > 0), as tests/trans/test_adjoint.F90 */
  {
    const int ns2 = trans.nspec2, nf = nscalar;
    double *x = malloc(sizeof(double) * ns2 * nf), *y = malloc(sizeof(double) * ns2 * nf), *p = calloc((size_t)ns2 * nf, sizeof(double));
    double *g = malloc(sizeof(double) * (size_t)nf * trans.ngptot), *w = malloc(sizeof(double) * ns2);
    unsigned s = 12345u;
    for (int i = 0; i < ns2 * nf; i++) {
      s = s * 1664525u + 1013904223u;
      x[i] = (double)(s >> 8) / 8388608.0 - 1.0;
      s = s * 1664525u + 1013904223u;
      y[i] = (double)(s >> 8) / 8388608.0 - 1.0;
    }
    for (int i = 0; i < ns2; i++) w[i] = 2.0;
    for (int n = 0; n <= trans.nsmax; n++) w[trans.nasm0[0] - 1 + 2 * n] = 1.0, w[trans.nasm0[0] + 2 * n] = 0.0;
    struct InvTrans_t v2 = new_invtrans(&trans);
    v2.nscalar = nf, v2.rspscalar = x, v2.rgp = g;
    CHECK(trans_invtrans(&v2));
    struct DirTrans_t d5 = new_dirtrans(&trans);
    d5.nscalar = nf, d5.rgp = g, d5.rspscalar = p;
    CHECK(trans_dirtrans(&d5));
    double s1 = 0, s2 = 0;
    for (int i = 0; i < ns2; i++)
      for (int f = 0; f < nf; f++) s1 += w[i] * p[i * nf + f] * y[i * nf + f];
    struct DirTransAdj_t da = new_dirtrans_adj(&trans);
    da.nscalar = nf, da.rspscalar = y, da.rgp = g;
    CHECK(trans_dirtrans_adj(&da));
    struct InvTransAdj_t va = new_invtrans_adj(&trans);
    va.nscalar = nf, va.rspscalar = p, va.rgp = g;
    CHECK(trans_invtrans_adj(&va));
    for (int i = 0; i < ns2; i++)
      for (int f = 0; f < nf; f++) s2 += w[i] * x[i * nf + f] * p[i * nf + f];
    if (fabs(s1 - s2) / fabs(s1) > 2000 * 2.220446049250313e-16) {
      fprintf(stderr, "adjoint test: %g vs %g\n", s1, s2);
      return 12;
    }
    free(x), free(y), free(p), free(g), free(w);
  }
  { /* Legendre polynomials written to a file, then taken from a memory image of it and from the file
     * itself: the sequence of the reference's tests/transi/transi_test_io.c:34-83, plus a check that the
     * transforms of the three set-ups agree exactly */
    const char *tmp = getenv("TMPDIR");
    char path[600];
    snprintf(path, sizeof(path), "%s/emi_transi_legpol.bin", tmp && tmp[0] ? tmp : "/tmp");
    const int ns2 = trans.nspec2, ng = trans.ngptot;
    double *x = calloc(ns2, sizeof(double)), *g0 = malloc(sizeof(double) * ng), *g1 = malloc(sizeof(double) * ng);
    x[trans.nasm0[4] - 1 + 2 * (19 - 4)] = 1.0;
    x[trans.nasm0[0] - 1 + 2 * 3] = -0.5;
    struct InvTrans_t v0 = new_invtrans(&trans);
    v0.nscalar = 1, v0.rspscalar = x, v0.rgp = g0;
    CHECK(trans_invtrans(&v0));
    struct Trans_t tw;
    CHECK(trans_new(&tw));
    CHECK(trans_set_resol(&tw, ndgl, nloen));
    CHECK(trans_set_trunc(&tw, nsmax));
    CHECK(trans_set_write(&tw, path));
    CHECK(trans_setup(&tw));
    CHECK(trans_delete(&tw));
    FILE *f = fopen(path, "rb");
    if (!f) return 20;
    fseek(f, 0, SEEK_END);
    size_t size = (size_t)ftell(f);
    rewind(f);
    void *buffer = malloc(size);
    if (fread(buffer, 1, size, f) != size) return 21;
    fclose(f);
    for (int pass = 0; pass < 2; pass++) {
      struct Trans_t tr;
      CHECK(trans_new(&tr));
      CHECK(trans_set_resol(&tr, ndgl, nloen));
      CHECK(trans_set_trunc(&tr, nsmax));
      if (pass == 0)
        CHECK(trans_set_cache(&tr, buffer, size));
      else
        CHECK(trans_set_read(&tr, path));
      CHECK(trans_setup(&tr));
      struct InvTrans_t v1 = new_invtrans(&tr);
      v1.nscalar = 1, v1.rspscalar = x, v1.rgp = g1;
      CHECK(trans_invtrans(&v1));
      for (int i = 0; i < ng; i++)
        if (g1[i] != g0[i]) {
          fprintf(stderr, "legendre %s: point %d differs\n", pass ? "file" : "cache", i);
          return 22;
        }
      CHECK(trans_delete(&tr));
    }
    { /* a cache of another truncation is refused with the reference's message */
      struct Trans_t tr;
      CHECK(trans_new(&tr));
      CHECK(trans_set_resol(&tr, ndgl, nloen));
      CHECK(trans_set_trunc(&tr, nsmax - 1));
      CHECK(trans_set_cache(&tr, buffer, size));
      int rc = trans_setup(&tr);
      if (rc == TRANS_SUCCESS || !strstr(trans_error_msg(rc), "READ_LEGPOL:WRONG SPECTRAL TRUNCATION")) return 23;
      tr.handle = 0;
      trans_delete(&tr);
      CHECK(trans_new(&tr));
      CHECK(trans_set_resol(&tr, ndgl, nloen));
      CHECK(trans_set_cache(&tr, NULL, size));
      if (trans_setup(&tr) != TRANS_MISSING_ARG) return 24;
      trans_delete(&tr);
    }
    remove(path);
    free(buffer), free(x), free(g0), free(g1);
    printf("legendre file / cache set-ups identical (%zu bytes)\n", size);
  }
  { /* trans_vordiv_to_UV (transi.h:620-648): solid-body rotation, vor = 2 U / (a sqrt 3) P_1^0 => U = 2U/3 P_0 - 2U/(3 sqrt 5) P_2, V = 0
       (transi's planet radius is 6371.22 km) */
    const double U = 30.0, a = 6371.22e3;
    double *vor = calloc((size_t)trans.nspec2, 8), *div = calloc((size_t)trans.nspec2, 8);
    double *pu = malloc(sizeof(double) * trans.nspec2), *pv = malloc(sizeof(double) * trans.nspec2);
    vor[trans.nasm0[0] - 1 + 2] = 2 * U / (a * sqrt(3.0));
    struct VorDivToUV_t vd = new_vordiv_to_UV();
    vd.rspvor = vor, vd.rspdiv = div, vd.rspu = pu, vd.rspv = pv, vd.nfld = 1, vd.ncoeff = trans.nspec2, vd.nsmax = nsmax;
    CHECK(trans_vordiv_to_UV(&vd));
    if (trans_vordiv_to_UV(&vd) != TRANS_STALE_ARG) return 30;
    for (int i = 0; i < trans.nspec2; i++) {
      const double want = i == trans.nasm0[0] - 1 ? 2 * U / 3 : (i == trans.nasm0[0] - 1 + 4 ? -2 * U / (3 * sqrt(5.0)) : 0.0);
      if (fabs(pu[i] - want) > 1e-12 * U || fabs(pv[i]) > 1e-12 * U) {
        fprintf(stderr, "vordiv_to_UV: coefficient %d: %g %g (want %g, 0)\n", i, pu[i], pv[i], want);
        return 31;
      }
    }
    free(vor), free(div), free(pu), free(pv);
    printf("trans_vordiv_to_UV ok\n");
  }
  CHECK(trans_delete(&trans));
  CHECK(trans_finalize());
  printf("TRANSI API OK\n");
  return 0;
}
