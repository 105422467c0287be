/* A C host of the C-ABI over the native RCCL transport (ectrans_amd/rccl/emi_rccl_hook.c), the RCCL twin of
 * tests/mpi/test_mpi_hook.c.  Every task is one GPU of the W-set; the 128-byte RCCL unique id travels by MPI_Bcast
 * (-DEMI_TEST_WITH_MPI) -- without MPI the program is the single task of a 1-task communicator.  Benchmark semantics
 * (ectrans-benchmark.F90:1390-1415): Re(m=4, n=19) = 1 in every field, two inverse + direct round trips keep the global
 * spectral norm to 100 eps.  The exchange itself is additionally driven directly through the registered hook with a
 * known pattern (every task sends block p = 1000 me + dst to task dst).
 * RCCL refuses two tasks on ONE device ("Duplicate GPU detected"): on a one-GPU box the multi-task run reports
 * "RCCL REFUSED" and exits 0 so that the caller can skip. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef EMI_TEST_WITH_MPI
#include <mpi.h>
#endif

#include "../../ectrans_amd/rccl/emi_rccl_hook.h"
#include "../../include/ectrans_mi.h"

static int rank = 0, size = 1;
static void fail(int code) {
#ifdef EMI_TEST_WITH_MPI
  MPI_Abort(MPI_COMM_WORLD, code);
#endif
  exit(code);
}
#define CHECK(x)                                                                                              \
  do {                                                                                                        \
    if ((x) != 0) {                                                                                           \
      fprintf(stderr, "rank %d: %s failed: %s | %s\n", rank, #x, emi_last_error(), emi_rccl_last_error());    \
      fail(1);                                                                                                \
    }                                                                                                         \
  } while (0)

int main(int argc, char **argv) {
  (void)argc, (void)argv;
#ifdef EMI_TEST_WITH_MPI
  MPI_Init(&argc, &argv);
  MPI_Comm_rank(MPI_COMM_WORLD, &rank);
  MPI_Comm_size(MPI_COMM_WORLD, &size);
#endif
  int ndev = 0;
  hipGetDeviceCount(&ndev);
  if (size > ndev) { /* several tasks per device: RCCL cannot build the communicator */
    printf("RCCL REFUSED rank %d: %d tasks on %d device(s)\n", rank, size, ndev);
#ifdef EMI_TEST_WITH_MPI
    MPI_Finalize();
#endif
    return 0;
  }
  char id[EMI_RCCL_ID_BYTES];
  if (rank == 0) CHECK(emi_rccl_get_unique_id(id));
#ifdef EMI_TEST_WITH_MPI
  MPI_Bcast(id, EMI_RCCL_ID_BYTES, MPI_BYTE, 0, MPI_COMM_WORLD);
#endif
  CHECK(emi_rccl_attach(id, size, rank + 1, 1, 0, 0.0, -1));
  const int N = 63, ndgl = 128, nfld = 300; /* 300 fields: with several tasks the calls run as pipelined batches */
  int *nloen = malloc(sizeof(int) * ndgl);
  for (int i = 0; i < ndgl / 2; i++) nloen[i] = nloen[ndgl - 1 - i] = 20 + 4 * i;
  emi_setup_t cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.ksmax = N, cfg.kdgl = ndgl, cfg.kloen = nloen, cfg.precision = 8;
  int r = 0;
  CHECK(emi_setup(&cfg, &r));
  int nspec2, ngptot;
  CHECK(emi_inq_int(r, "nspec2", &nspec2));
  CHECK(emi_inq_int(r, "ngptot", &ngptot));
  int *nasm0 = malloc(sizeof(int) * (N + 1));
  CHECK(emi_inq_int_array(r, "nasm0", nasm0, N + 1));
  const int own4 = nasm0[4] > 0;
  const long i419 = own4 ? nasm0[4] - 1 + 2 * (19 - 4) : -1;
  /* device-resident fields: the transport never touches host memory */
  double *sp, *gp, *hsp = calloc((size_t)nspec2 * nfld, sizeof(double));
  if (own4)
    for (int f = 0; f < nfld; f++) hsp[i419 * nfld + f] = 1.0;
  if (hipMalloc((void **)&sp, sizeof(double) * (size_t)nspec2 * nfld) != hipSuccess ||
      hipMalloc((void **)&gp, sizeof(double) * (size_t)ngptot * nfld) != hipSuccess)
    fail(4);
  hipMemcpy(sp, hsp, sizeof(double) * (size_t)nspec2 * nfld, hipMemcpyHostToDevice);
  hipMemset(gp, 0, sizeof(double) * (size_t)ngptot * nfld);
  double *n0 = malloc(sizeof(double) * nfld), *n1 = malloc(sizeof(double) * nfld);
  CHECK(emi_rccl_specnorm(r, EMI_MEM_DEVICE, sp, nfld, n0));
  for (int it = 0; it < 2; it++) {
    emi_invtrans_t a;
    memset(&a, 0, sizeof(a));
    a.mem_space = EMI_MEM_DEVICE, a.spscalar = sp, a.nf_scalar = nfld, a.gp = gp, a.gp_nfld = nfld, a.kproma = ngptot;
    CHECK(emi_inv_trans(r, &a));
    emi_dirtrans_t d;
    memset(&d, 0, sizeof(d));
    d.mem_space = EMI_MEM_DEVICE, d.spscalar = sp, d.nf_scalar = nfld, d.gp = gp, d.gp_nfld = nfld, d.kproma = ngptot;
    CHECK(emi_dir_trans(r, &d));
  }
  CHECK(emi_rccl_specnorm(r, EMI_MEM_DEVICE, sp, nfld, n1));
  hipMemcpy(hsp, sp, sizeof(double) * (size_t)nspec2 * nfld, hipMemcpyDeviceToHost);
  double err = 0;
  for (int f = 0; f < nfld; f++) err = fmax(err, fabs(n0[f] / n1[f] - 1.0));
  if (fabs(n0[0] - sqrt(2.0)) > 1e-14 || err > 100 * 2.220446049250313e-16 || (own4 && fabs(hsp[i419 * nfld + 7] - 1.0) > 1e-13)) {
    fprintf(stderr, "rank %d: norm %.17g drift %.3e coefficient %.17g\n", rank, n0[0], err, own4 ? hsp[i419 * nfld + 7] : 0.0);
    fail(3);
  }
  /* the exchange itself with a known pattern: task s sends (1000 s + d) x 64 doubles to task d */
  {
    const int nb = 64;
    double *hs = malloc(sizeof(double) * nb * size), *hr = malloc(sizeof(double) * nb * size), *ds, *dr;
    long long *cnt = malloc(sizeof(long long) * 2 * size), *dsp = cnt + size;
    for (int d = 0; d < size; d++) {
      for (int i = 0; i < nb; i++) hs[d * nb + i] = 1000.0 * rank + d + 1e-3 * i;
      cnt[d] = (long long)nb * 8, dsp[d] = (long long)d * nb * 8;
    }
    hipStream_t st;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess || hipMalloc((void **)&ds, sizeof(double) * nb * size) != hipSuccess ||
        hipMalloc((void **)&dr, sizeof(double) * nb * size) != hipSuccess)
      fail(5);
    hipMemcpyAsync(ds, hs, sizeof(double) * nb * size, hipMemcpyHostToDevice, st);
    CHECK(emi_rccl_alltoallv(NULL, ds, cnt, dsp, dr, cnt, dsp, size, st));
    hipMemcpyAsync(hr, dr, sizeof(double) * nb * size, hipMemcpyDeviceToHost, st);
    hipStreamSynchronize(st);
    for (int s = 0; s < size; s++)
      for (int i = 0; i < nb; i++)
        if (hr[s * nb + i] != 1000.0 * s + rank + 1e-3 * i) {
          fprintf(stderr, "rank %d: block from %d element %d is %.6f\n", rank, s, i, hr[s * nb + i]);
          fail(6);
        }
    hipFree(ds), hipFree(dr), hipStreamDestroy(st);
  }
  CHECK(emi_release(r));
  CHECK(emi_finalize());
  CHECK(emi_rccl_detach());
  printf("RCCL HOOK OK rank %d of %d (drift %.2e)\n", rank, size, err);
#ifdef EMI_TEST_WITH_MPI
  MPI_Finalize();
#endif
  return 0;
}
