/* An MPI host of the C-ABI: every rank is one task of the W-set (several ranks may share a GPU here) and the
 * all-to-all-v between them is ectrans_amd/mpi/emi_mpi_hook.c.  Benchmark semantics
 * (ectrans-benchmark.F90:1390-1415): every field carries Re(m=4, n=19) = 1; two inverse + direct round
 * trips must keep the global spectral norm to 100 eps and return the coefficient; the local sizes must add
 * up to the global ones.  (Numerical agreement of the decomposed transform with the oracle is checked by
 * tests/dist_worker.py, which shares everything but the transport of the exchange with this host.) */
#include <math.h>
#include <mpi.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../ectrans_amd/mpi/emi_mpi_hook.h"
#include "../../include/ectrans_mi.h"

#define CHECK(x)                                                              \
  do {                                                                        \
    if ((x) != 0) {                                                           \
      fprintf(stderr, "rank %d: %s failed: %s\n", rank, #x, emi_last_error()); \
      MPI_Abort(MPI_COMM_WORLD, 1);                                           \
    }                                                                         \
  } while (0)

int main(int argc, char **argv) {
  int rank, size;
  MPI_Init(&argc, &argv);
  MPI_Comm_rank(MPI_COMM_WORLD, &rank);
  MPI_Comm_size(MPI_COMM_WORLD, &size);
  const int N = 63, ndgl = 128, nfld = 300; /* 300 fields: the calls run as pipelined batches */
  CHECK(emi_mpi_attach(MPI_COMM_WORLD, 1, 0, 0.0, -1));
  int *nloen = malloc(sizeof(int) * ndgl);
  for (int i = 0; i < ndgl / 2; i++) nloen[i] = nloen[ndgl - 1 - i] = 20 + 4 * i;
  emi_setup_t cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.ksmax = N, cfg.kdgl = ndgl, cfg.kloen = nloen, cfg.precision = 8;
  int r = 0;
  CHECK(emi_setup(&cfg, &r));
  int nspec2, ngptot, nspec2g, ngptotg, nump;
  CHECK(emi_inq_int(r, "nspec2", &nspec2));
  CHECK(emi_inq_int(r, "ngptot", &ngptot));
  CHECK(emi_inq_int(r, "nspec2g", &nspec2g));
  CHECK(emi_inq_int(r, "ngptotg", &ngptotg));
  CHECK(emi_inq_int(r, "nump", &nump));
  long sums[2] = {nspec2, ngptot}, tot[2];
  MPI_Allreduce(sums, tot, 2, MPI_LONG, MPI_SUM, MPI_COMM_WORLD);
  if (tot[0] != nspec2g || tot[1] != ngptotg) {
    fprintf(stderr, "rank %d: local sizes do not add up (%ld/%d, %ld/%d)\n", rank, tot[0], nspec2g, tot[1], ngptotg);
    MPI_Abort(MPI_COMM_WORLD, 2);
  }
  int *nasm0 = malloc(sizeof(int) * (N + 1));
  CHECK(emi_inq_int_array(r, "nasm0", nasm0, N + 1));
  const int own4 = nasm0[4] > 0; /* -99 for a wavenumber of another task */
  const long i419 = own4 ? nasm0[4] - 1 + 2 * (19 - 4) : -1;
  double *sp = calloc((size_t)nspec2 * nfld, sizeof(double)), *gp = calloc((size_t)ngptot * nfld, sizeof(double));
  if (own4)
    for (int f = 0; f < nfld; f++) sp[i419 * nfld + f] = 1.0;
  double *n0 = malloc(sizeof(double) * nfld), *n1 = malloc(sizeof(double) * nfld);
  CHECK(emi_mpi_specnorm(MPI_COMM_WORLD, r, EMI_MEM_HOST, sp, nfld, n0));
  for (int it = 0; it < 2; it++) {
    emi_invtrans_t a;
    memset(&a, 0, sizeof(a));
    a.mem_space = EMI_MEM_HOST, a.spscalar = sp, a.nf_scalar = nfld, a.gp = gp, a.gp_nfld = nfld, a.kproma = ngptot;
    CHECK(emi_inv_trans(r, &a));
    emi_dirtrans_t d;
    memset(&d, 0, sizeof(d));
    d.mem_space = EMI_MEM_HOST, d.spscalar = sp, d.nf_scalar = nfld, d.gp = gp, d.gp_nfld = nfld, d.kproma = ngptot;
    CHECK(emi_dir_trans(r, &d));
  }
  CHECK(emi_mpi_specnorm(MPI_COMM_WORLD, r, EMI_MEM_HOST, sp, nfld, n1));
  double err = 0;
  for (int f = 0; f < nfld; f++) err = fmax(err, fabs(n0[f] / n1[f] - 1.0));
  if (fabs(n0[0] - sqrt(2.0)) > 1e-14 || err > 100 * 2.220446049250313e-16 || (own4 && fabs(sp[i419 * nfld + 7] - 1.0) > 1e-13)) {
    fprintf(stderr, "rank %d: norm %.17g drift %.3e coefficient %.17g\n", rank, n0[0], err, own4 ? sp[i419 * nfld + 7] : 0.0);
    MPI_Abort(MPI_COMM_WORLD, 3);
  }
  /* the grid field of the harmonic has zonal wavenumber 4: its local piece must not be zero */
  double gmax = 0, gall;
  for (long i = 0; i < (long)ngptot; i++) gmax = fmax(gmax, fabs(gp[i]));
  MPI_Allreduce(&gmax, &gall, 1, MPI_DOUBLE, MPI_MAX, MPI_COMM_WORLD);
  CHECK(emi_release(r));
  CHECK(emi_finalize());
  emi_mpi_detach();
  printf("rank %d/%d: nump %d ngptot %d norm drift %.2e max|gp| %.3f  MPI HOOK OK\n", rank, size, nump, ngptot, err, gall);
  MPI_Finalize();
  return 0;
}
