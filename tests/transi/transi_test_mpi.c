/* The transi-style C layer with several tasks (the reference: transi.h with TRANS_USE_MPI=1, tests/transi/*.c under
 * mpirun): the MPI transport is attached first, trans_init adopts its tasks, and
 *   trans_distspec (global fields on tasks 1 and N) -> trans_invtrans -> trans_gathgrid (to tasks N and 1) ->
 *   trans_distgrid -> trans_dirtrans -> trans_gathspec (back to tasks 1 and N)
 * returns the global spectral fields (1e-10); trans_specnorm gives every task the global norms.  T47 / O48.
 * Run: mpiexec -n N transi_test_mpi   (tests/test_gpu_shims.py) */
#include <math.h>
#include <mpi.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../ectrans_amd/mpi/emi_mpi_hook.h"
#include "../../ectrans_amd/transi/transi_mi.h"

#define CHECK(x)                                                                       \
  do {                                                                                 \
    int rc_ = (x);                                                                     \
    if (rc_ != TRANS_SUCCESS) {                                                        \
      fprintf(stderr, "%s failed: %s\n", #x, trans_error_msg(rc_));                   \
      MPI_Abort(MPI_COMM_WORLD, 1);                                                    \
    }                                                                                  \
  } while (0)
#define FAIL(code)                                                           \
  do {                                                                       \
    fprintf(stderr, "transi_test_mpi: check %d failed on task %d\n", code, rank + 1); \
    MPI_Abort(MPI_COMM_WORLD, code);                                         \
  } while (0)

int main(int argc, char **argv) {
  int rank, size;
  MPI_Init(&argc, &argv);
  MPI_Comm_rank(MPI_COMM_WORLD, &rank);
  MPI_Comm_size(MPI_COMM_WORLD, &size);
  if (trans_use_mpi(1) == TRANS_SUCCESS) FAIL(1); /* no transport attached yet */
  if (emi_mpi_attach(MPI_COMM_WORLD, 4, 0, 6371.22e3, -1) != 0) FAIL(2);
  CHECK(trans_use_mpi(1));
  CHECK(trans_set_nprtrv(1));
  CHECK(trans_init());
  const int N = 47, H = N + 1;
  int nloen[2 * 48];
  for (int i = 0; i < H; i++) nloen[i] = nloen[2 * H - 1 - i] = 20 + 4 * i;
  struct Trans_t trans;
  CHECK(trans_new(&trans));
  CHECK(trans_set_resol(&trans, 2 * H, nloen));
  CHECK(trans_set_trunc(&trans, N));
  CHECK(trans_setup(&trans));
  if (trans.nproc != size || trans.myproc != rank + 1) FAIL(3);
  int loc[2] = {trans.nspec2, trans.ngptot}, tot[2];
  MPI_Allreduce(loc, tot, 2, MPI_INT, MPI_SUM, MPI_COMM_WORLD);
  if (tot[0] != trans.nspec2g || tot[1] != trans.ngptotg || trans.nspec2g != (N + 1) * (N + 2)) FAIL(4);

  /* two scalar fields: the global spectrum of field 0 lives on task 1, of field 1 on the last task */
  const int nf = 2, me = rank + 1;
  const int from[2] = {1, size}, swapped[2] = {size, 1};
  const int nmine = (from[0] == me) + (from[1] == me);
  const size_t ng = (size_t)trans.nspec2g;
  double *specg = nmine ? malloc(sizeof(double) * ng * nmine) : NULL, *back = nmine ? malloc(sizeof(double) * ng * nmine) : NULL;
  for (int k = 0, f = 0; f < nf; f++) {
    if (from[f] != me) continue;
    size_t i = 0;
    for (int m = 0; m <= N; m++)
      for (int n = m; n <= N; n++, i += 2) { /* [nspec2g][nmine] */
        specg[i * nmine + k] = cos(0.3 * m + f) / (1.0 + n);
        specg[(i + 1) * nmine + k] = m ? sin(0.7 * n + f) / (1.0 + n) : 0.0;
      }
    k++;
  }
  double *spec = malloc(sizeof(double) * (size_t)trans.nspec2 * nf);
  struct DistSpec_t ds = new_distspec(&trans);
  ds.rspecg = specg, ds.rspec = spec, ds.nfld = nf, ds.nfrom = from;
  CHECK(trans_distspec(&ds));
  double *gp = malloc(sizeof(double) * (size_t)trans.ngptot * nf);
  struct InvTrans_t iv = new_invtrans(&trans);
  iv.nscalar = nf, iv.rspscalar = spec, iv.rgp = gp;
  CHECK(trans_invtrans(&iv));
  /* grid fields gathered to the OTHER task, scattered back from there */
  const int gmine = (swapped[0] == me) + (swapped[1] == me);
  double *gpg = gmine ? malloc(sizeof(double) * (size_t)trans.ngptotg * gmine) : NULL;
  struct GathGrid_t gg = new_gathgrid(&trans);
  gg.rgpg = gpg, gg.rgp = gp, gg.nfld = nf, gg.nto = swapped;
  CHECK(trans_gathgrid(&gg));
  const int np = 1000, nb = (trans.ngptot - 1) / np + 1;
  double *blk = calloc((size_t)nb * nf * np, sizeof(double));
  struct DistGrid_t dg = new_distgrid(&trans);
  dg.rgpg = gpg, dg.rgp = blk, dg.nfld = nf, dg.nfrom = swapped, dg.nproma = np, dg.ngpblks = nb;
  CHECK(trans_distgrid(&dg));
  double *spec2 = calloc((size_t)trans.nspec2 * nf, sizeof(double));
  struct DirTrans_t dt = new_dirtrans(&trans);
  dt.nscalar = nf, dt.rspscalar = spec2, dt.rgp = blk, dt.nproma = np, dt.ngpblks = nb;
  CHECK(trans_dirtrans(&dt));
  struct GathSpec_t gs = new_gathspec(&trans);
  gs.rspecg = back, gs.rspec = spec2, gs.nfld = nf, gs.nto = from;
  CHECK(trans_gathspec(&gs));
  double worst = 0.0;
  for (size_t i = 0; i < ng * (size_t)nmine; i++) worst = fmax(worst, fabs(back[i] - specg[i]));
  /* octahedral grids drop the (latitude, m > NMEN(latitude)) corner (setup_geom_mod.F90:64-78): a dense spectrum comes back
   * to ~1e-11, the same with one task and with several */
  if (worst > 1e-10) {
    fprintf(stderr, "round trip error %.3e\n", worst);
    FAIL(5);
  }
  /* global norms on every task */
  double norm[2], want[2] = {0.0, 0.0};
  struct SpecNorm_t sn = new_specnorm(&trans);
  sn.rspec = spec, sn.nfld = nf, sn.rnorm = norm;
  CHECK(trans_specnorm(&sn));
  for (int f = 0; f < nf; f++) {
    double s = 0.0;
    for (int m = 0; m <= N; m++)
      for (int n = m; n <= N; n++) {
        const double re = cos(0.3 * m + f) / (1.0 + n), im = m ? sin(0.7 * n + f) / (1.0 + n) : 0.0;
        s += (m ? 2.0 : 1.0) * (re * re + im * im);
      }
    want[f] = sqrt(s);
    if (fabs(norm[f] / want[f] - 1.0) > 1e-13) FAIL(6);
  }
  /* lglobal is a one-task configuration (transi_module.F90:1514-1527) */
  if (size > 1) {
    struct DirTrans_t bad = new_dirtrans(&trans);
    bad.nscalar = nf, bad.rspscalar = spec2, bad.rgp = blk, bad.lglobal = 1;
    if (trans_dirtrans(&bad) == TRANS_SUCCESS) FAIL(7);
  }
  CHECK(trans_delete(&trans));
  CHECK(trans_finalize());
  emi_mpi_detach();
  printf("TRANSI MPI OK task %d of %d (round trip %.1e)\n", me, size, worst);
  MPI_Finalize();
  return 0;
}
