/*
 * emi_rccl_hook.h -- native RCCL transport of the TRLTOM / TRMTOL exchange for non-Python hosts.
 *
 * The reference exchanges the Fourier coefficients between its latitude and wavenumber distributions with
 * MPL_ALLTOALLV (trans/cpu/internal/trltom_mod.F90:96-136, trmtol_mod.F90:101-141; device buffers through GPU-aware MPI in
 * its GPU back-end, trans/gpu/internal/trltom_mod.F90:148-209).  On one MI355X node the tasks are the 8 GPUs and the
 * transport is RCCL over xGMI: this file registers, as libectrans_mi's all-to-all-v hook (emi_set_alltoallv), a grouped
 * ncclSend / ncclRecv exchange of the device-resident row blocks on the stream the library passes -- no host staging,
 * no Python.  One process per GPU; the host only has to carry the 128-byte RCCL unique id from task 1 to the others
 * (MPI_Bcast in a Fortran / C host, a file, a socket).
 */
#ifndef EMI_RCCL_HOOK_H
#define EMI_RCCL_HOOK_H
#ifdef __cplusplus
extern "C" {
#endif

#define EMI_RCCL_ID_BYTES 128 /* sizeof(ncclUniqueId) */

/* Task 1: a fresh unique id into id[EMI_RCCL_ID_BYTES]; the host broadcasts the bytes to the other tasks. */
int emi_rccl_get_unique_id(void *id);

/* SETUP_TRANS0 of task `myproc` (1-based) of `nproc` tasks, one GPU each (NPRTRW = nproc, NPRTRV = NPRGPEW = 1):
 * selects HIP device `device` (< 0: myproc - 1 modulo the visible devices), creates the communicator
 * (ncclCommInitRank), registers the exchange and calls emi_init.  kmax_resol / kprintlev / prad as emi_init_t. */
int emi_rccl_attach(const void *id, int nproc, int myproc, int kmax_resol, int kprintlev, double prad, int device);
/* The same on a communicator the host already owns (ncclComm_t passed as void *). */
int emi_rccl_attach_comm(void *nccl_comm, int nproc, int myproc, int kmax_resol, int kprintlev, double prad, int device);

/* The exchange itself (what emi_rccl_attach registers with emi_set_alltoallv; signature emi_alltoallv_fn of
 * include/ectrans_mi.h): counts and displacements in BYTES per task, device buffers, ordered on `stream`. */
int emi_rccl_alltoallv(void *user, const void *sendbuf, const long long *sendcounts, const long long *sdispls, void *recvbuf,
                       const long long *recvcounts, const long long *rdispls, int nproc, void *stream);

/* SPECNORM over all tasks: emi_specnorm_partial + ncclAllReduce of the per-field sums + square root; every task gets
 * the norms (the reference returns them on the master task only, spnormc_mod.F90:49-85). */
int emi_rccl_specnorm(int kresol, int mem_space, const void *spec, int nfld, double *norms);

/* TRANS_END counterpart of the transport: destroys a communicator this file created. */
int emi_rccl_detach(void);

const char *emi_rccl_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
