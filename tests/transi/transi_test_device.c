/* C caller of the transi-style API whose rgp / rspscalar / rspvor / rspdiv live in DEVICE memory (hipMalloc): the layer passes
 * EMI_MEM_AUTO, the library classifies the pointers and uses the arrays in place -- the reference GPU back-end's
 * present-or-copyin (trans/gpu/internal/trltog_mod.F90:501-523, ltinv_mod.F90:334-338).  Checks: the harmonic (4,19) of the
 * benchmark through trans_invtrans + trans_dirtrans + trans_specnorm on device arrays; the SAME call on host arrays gives the
 * same bits; a call with arrays in both memories is refused.  Exit code 0 = pass. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "../../ectrans_amd/transi/transi_mi.h"

#define CHECK(x)                                                    \
  do {                                                              \
    int rc_ = (x);                                                  \
    if (rc_ != TRANS_SUCCESS) {                                     \
      fprintf(stderr, "%s failed: %s\n", #x, trans_error_msg(rc_)); \
      return 1;                                                     \
    }                                                               \
  } while (0)
#define HIPOK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));              \
      return 1;                                                                   \
    }                                                                             \
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
  CHECK(trans_inquire(&trans, "nasm0"));
  const int nscalar = 3, nvordiv = 2, nfld = 2 * nvordiv + nscalar;
  const size_t ns = (size_t)trans.nspec2, ng = (size_t)trans.ngptot;
  const size_t bsc = sizeof(double) * nscalar * ns, buv = sizeof(double) * nvordiv * ns, bgp = sizeof(double) * nfld * ng;
  double *hsc = calloc(1, bsc), *hvor = calloc(1, buv), *hdiv = calloc(1, buv), *hgp = calloc(1, bgp);
  const int i419 = trans.nasm0[4] - 1 + 2 * (19 - 4);
  for (int f = 0; f < nscalar; f++) hsc[(size_t)i419 * nscalar + f] = 1.0;
  for (int f = 0; f < nvordiv; f++) hvor[(size_t)i419 * nvordiv + f] = hdiv[(size_t)i419 * nvordiv + f] = 1.0;
  double *dsc, *dvor, *ddiv, *dgp;
  HIPOK(hipMalloc((void **)&dsc, bsc));
  HIPOK(hipMalloc((void **)&dvor, buv));
  HIPOK(hipMalloc((void **)&ddiv, buv));
  HIPOK(hipMalloc((void **)&dgp, bgp));
  HIPOK(hipMemcpy(dsc, hsc, bsc, hipMemcpyHostToDevice));
  HIPOK(hipMemcpy(dvor, hvor, buv, hipMemcpyHostToDevice));
  HIPOK(hipMemcpy(ddiv, hdiv, buv, hipMemcpyHostToDevice));
  HIPOK(hipMemset(dgp, 0, bgp));

  /* a call with arrays in both memories is refused */
  struct InvTrans_t bad = new_invtrans(&trans);
  bad.nscalar = nscalar, bad.nvordiv = nvordiv, bad.rgp = hgp, bad.rspscalar = dsc, bad.rspvor = dvor, bad.rspdiv = ddiv;
  if (trans_invtrans(&bad) == TRANS_SUCCESS) {
    fprintf(stderr, "a call with host and device arrays was not refused\n");
    return 2;
  }

  double n0[3], n1[3];
  struct SpecNorm_t sn = new_specnorm(&trans);
  sn.rspec = dsc, sn.nfld = nscalar, sn.rnorm = n0;
  CHECK(trans_specnorm(&sn));
  for (int it = 0; it < 2; it++) {
    struct InvTrans_t v = new_invtrans(&trans);
    v.nscalar = nscalar, v.nvordiv = nvordiv, v.rgp = dgp, v.rspscalar = dsc, v.rspvor = dvor, v.rspdiv = ddiv;
    CHECK(trans_invtrans(&v));
    struct DirTrans_t d = new_dirtrans(&trans);
    d.nscalar = nscalar, d.nvordiv = nvordiv, d.rgp = dgp, d.rspscalar = dsc, d.rspvor = dvor, d.rspdiv = ddiv;
    CHECK(trans_dirtrans(&d));
    /* the same pair on the host copies: staged through the same kernels */
    struct InvTrans_t vh = new_invtrans(&trans);
    vh.nscalar = nscalar, vh.nvordiv = nvordiv, vh.rgp = hgp, vh.rspscalar = hsc, vh.rspvor = hvor, vh.rspdiv = hdiv;
    CHECK(trans_invtrans(&vh));
    struct DirTrans_t dh = new_dirtrans(&trans);
    dh.nscalar = nscalar, dh.nvordiv = nvordiv, dh.rgp = hgp, dh.rspscalar = hsc, dh.rspvor = hvor, dh.rspdiv = hdiv;
    CHECK(trans_dirtrans(&dh));
  }
  sn = new_specnorm(&trans);
  sn.rspec = dsc, sn.nfld = nscalar, sn.rnorm = n1;
  CHECK(trans_specnorm(&sn));
  for (int f = 0; f < nscalar; f++) {
    if (fabs(n0[f] - sqrt(2.0)) > 1e-14 || fabs(n0[f] / n1[f] - 1.0) > 100 * 2.2e-16) {
      fprintf(stderr, "norm of field %d: %.17g -> %.17g\n", f, n0[f], n1[f]);
      return 3;
    }
  }
  /* device results back: bit-identical to the staged host-array calls */
  double *csc = malloc(bsc), *cvor = malloc(buv), *cgp = malloc(bgp);
  HIPOK(hipMemcpy(csc, dsc, bsc, hipMemcpyDeviceToHost));
  HIPOK(hipMemcpy(cvor, dvor, buv, hipMemcpyDeviceToHost));
  HIPOK(hipMemcpy(cgp, dgp, bgp, hipMemcpyDeviceToHost));
  if (memcmp(csc, hsc, bsc) || memcmp(cvor, hvor, buv) || memcmp(cgp, hgp, bgp)) {
    fprintf(stderr, "device-resident and staged calls differ\n");
    return 4;
  }
  if (fabs(csc[(size_t)i419 * nscalar] - 1.0) > 1e-12) return 5;
  HIPOK(hipFree(dsc));
  HIPOK(hipFree(dvor));
  HIPOK(hipFree(ddiv));
  HIPOK(hipFree(dgp));
  CHECK(trans_delete(&trans));
  CHECK(trans_finalize());
  printf("TRANSI DEVICE ARRAYS OK\n");
  return 0;
}
