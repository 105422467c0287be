/* C helper of tests/fortran/test_shim_mpi.F90: flang has no MPI module here, so the Fortran host starts MPI and attaches
 * the MPI transport (ectrans_amd/mpi/emi_mpi_hook.c) through these two BIND(C) entries -- what MPL_INIT does for the
 * reference.  A real Fortran host calls emi_mpi_attach with MPI_Comm_c2f / its own communicator. */
#include <mpi.h>

#include "../../ectrans_amd/mpi/emi_mpi_hook.h"
#include "../../include/ectrans_mi.h"

int emi_test_mpi_begin(int *nproc, int *myproc) {
  int rank, size;
  MPI_Init(0, 0);
  MPI_Comm_rank(MPI_COMM_WORLD, &rank);
  MPI_Comm_size(MPI_COMM_WORLD, &size);
  *nproc = size;
  *myproc = rank + 1;
  return emi_mpi_attach(MPI_COMM_WORLD, 2, 0, 0.0, -1);
}
void emi_test_mpi_end(void) {
  emi_mpi_detach();
  MPI_Finalize();
}
/* the same with NPRTRV V-sets: emi_set_nprtrv before the attach (the transport calls emi_init) */
int emi_test_mpi_begin_v(int nprtrv, int *nproc, int *myproc) {
  int rank, size;
  MPI_Init(0, 0);
  MPI_Comm_rank(MPI_COMM_WORLD, &rank);
  MPI_Comm_size(MPI_COMM_WORLD, &size);
  *nproc = size;
  *myproc = rank + 1;
  if (emi_set_nprtrv(nprtrv) != 0) return -1;
  return emi_mpi_attach(MPI_COMM_WORLD, 2, 0, 0.0, -1);
}
