/* emi_mpi_hook.h -- the all-to-all-v hook of libectrans_mi (include/ectrans_mi.h, emi_set_alltoallv) on MPI.
 *
 * What fiat's MPL_ALLTOALLV does for the reference at trltom_mod.F90:103-114 / trmtol_mod.F90:108-119:
 * one MPI_Alltoallv of whole blocks of the Fourier buffer per transform direction.  An MPI host (the
 * Fortran shim under IFS, ectrans-benchmark, a C program) calls emi_mpi_attach() once after MPI_Init:
 * it takes task count and task number from the communicator, selects the GPU, calls emi_init() and
 * registers the hook.
 *
 * Two transports, chosen at build time:
 *   default              the device blocks are staged through pinned host buffers (any MPI; tested with
 *                        MPICH 3.3 on the one-GPU box, several ranks sharing the GPU)
 *   -DEMI_MPI_GPU_AWARE  device pointers are handed to MPI_Alltoallv directly (GPU-aware MPI over xGMI)
 */
#ifndef EMI_MPI_HOOK_H
#define EMI_MPI_HOOK_H
#include <mpi.h>
#ifdef __cplusplus
extern "C" {
#endif
/* kmax_resol, kprintlev, prad as SETUP_TRANS0; device < 0: rank modulo the number of visible GPUs */
int emi_mpi_attach(MPI_Comm comm, int kmax_resol, int kprintlev, double prad, int device);
/* sum of per-task partial spectral norms (SPECNORM over tasks, spnormc_mod.F90:49-85) */
int emi_mpi_specnorm(MPI_Comm comm, int kresol, int mem_space, const void *spec, int nfld, double *norms);
void emi_mpi_detach(void);
#ifdef __cplusplus
}
#endif
#endif
