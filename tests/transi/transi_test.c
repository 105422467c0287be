/* C caller of the transi-style API, modelled on what the reference's transi_test_program.c checks
 * (tests/transi/transi_test_program.c:66-160: constant fields -> only coefficient 0 is non-zero)
 * plus a wind/scalar round trip.  Exit code 0 = pass. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

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
  CHECK(trans_delete(&trans));
  CHECK(trans_finalize());
  printf("TRANSI API OK\n");
  return 0;
}
