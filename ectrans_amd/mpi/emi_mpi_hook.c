/* emi_mpi_hook.c -- see emi_mpi_hook.h */
#include "emi_mpi_hook.h"

#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/ectrans_mi.h"

static MPI_Comm g_comm = MPI_COMM_NULL;
static void *g_hs = NULL, *g_hr = NULL; /* pinned staging buffers */
static size_t g_cap_s = 0, g_cap_r = 0;

static int grow_pinned(void **p, size_t *cap, size_t need) {
  if (need <= *cap) return 0;
  if (*p) hipHostFree(*p);
  *p = NULL;
  *cap = 0;
  if (hipHostMalloc(p, need, hipHostMallocDefault) != hipSuccess) return -1;
  *cap = need;
  return 0;
}

/* emi_alltoallv_fn: counts and displacements in BYTES, buffers are device pointers, ordered on `stream` */
static int hook(void *user, const void *sendbuf, const long long *sc, const long long *sd, void *recvbuf, const long long *rc,
                const long long *rd, int nproc, void *stream) {
  (void)user;
  hipStream_t st = (hipStream_t)stream;
  int *isc = malloc(sizeof(int) * 4 * (size_t)nproc), *isd = isc + nproc, *irc = isd + nproc, *ird = irc + nproc;
  long long stot = 0, rtot = 0;
  int rcode = 0;
  /* MPI counts are ints: exchange in units of 16 bytes (every block is a multiple of a complex fp64 /
   * of two complex fp32 numbers) */
  for (int r = 0; r < nproc; r++) {
    if ((sc[r] | sd[r] | rc[r] | rd[r]) & 15 || sc[r] / 16 > 2147483647LL || sd[r] / 16 > 2147483647LL || rd[r] / 16 > 2147483647LL) {
      free(isc);
      return -1;
    }
    isc[r] = (int)(sc[r] / 16), isd[r] = (int)(sd[r] / 16), irc[r] = (int)(rc[r] / 16), ird[r] = (int)(rd[r] / 16);
    if (sd[r] + sc[r] > stot) stot = sd[r] + sc[r];
    if (rd[r] + rc[r] > rtot) rtot = rd[r] + rc[r];
  }
  MPI_Datatype t16;
  MPI_Type_contiguous(16, MPI_BYTE, &t16);
  MPI_Type_commit(&t16);
  /* With V-sets the exchanges run inside a V-set (TRMTOL / TRLTOM) or inside a W-set (TRLTOG / TRGTOL), and two V-sets need not make
   * the same number of calls (their field counts differ, one may hold none): a collective over all tasks cannot take that, so
   * the blocks travel point to point between the tasks that really exchange one */
  int nprtrv = 1;
  (void)emi_inq_vsets(NULL, &nprtrv, NULL, NULL);
#define EMI_MPI_EXCHANGE(sb_, rb_)                                                                                          \
  do {                                                                                                                      \
    if (nprtrv <= 1) {                                                                                                      \
      if (MPI_Alltoallv(sb_, isc, isd, t16, rb_, irc, ird, t16, g_comm) != MPI_SUCCESS) rcode = -1;                         \
    } else {                                                                                                                \
      int me_ = 0, nreq_ = 0;                                                                                               \
      MPI_Comm_rank(g_comm, &me_);                                                                                          \
      MPI_Request *rq_ = malloc(sizeof(MPI_Request) * 2 * (size_t)nproc);                                                   \
      for (int r = 0; r < nproc; r++)                                                                                       \
        if (r != me_ && irc[r] > 0 && MPI_Irecv((char *)(rb_) + rd[r], irc[r], t16, r, 77, g_comm, &rq_[nreq_++]) != MPI_SUCCESS) rcode = -1; \
      for (int r = 0; r < nproc; r++)                                                                                       \
        if (r != me_ && isc[r] > 0 && MPI_Isend((const char *)(sb_) + sd[r], isc[r], t16, r, 77, g_comm, &rq_[nreq_++]) != MPI_SUCCESS) rcode = -1; \
      if (isc[me_] > 0) memcpy_own = 1;                                                                                     \
      if (MPI_Waitall(nreq_, rq_, MPI_STATUSES_IGNORE) != MPI_SUCCESS) rcode = -1;                                          \
      free(rq_);                                                                                                            \
    }                                                                                                                       \
  } while (0)
  int memcpy_own = 0;
#ifdef EMI_MPI_GPU_AWARE
  if (hipStreamSynchronize(st) != hipSuccess) rcode = -1; /* producer kernels done; MPI reads the device blocks */
  if (!rcode) EMI_MPI_EXCHANGE(sendbuf, recvbuf);
  if (!rcode && memcpy_own) {
    int me = 0;
    MPI_Comm_rank(g_comm, &me);
    if (hipMemcpy((char *)recvbuf + rd[me], (const char *)sendbuf + sd[me], (size_t)sc[me], hipMemcpyDeviceToDevice) != hipSuccess) rcode = -1;
  }
#else
  if (grow_pinned(&g_hs, &g_cap_s, (size_t)stot) || grow_pinned(&g_hr, &g_cap_r, (size_t)rtot)) rcode = -1;
  if (!rcode && stot && hipMemcpyAsync(g_hs, sendbuf, (size_t)stot, hipMemcpyDeviceToHost, st) != hipSuccess) rcode = -1;
  if (!rcode && hipStreamSynchronize(st) != hipSuccess) rcode = -1;
  if (!rcode) EMI_MPI_EXCHANGE(g_hs, g_hr);
  if (!rcode && memcpy_own) {
    int me = 0;
    MPI_Comm_rank(g_comm, &me);
    memcpy(g_hr + rd[me], g_hs + sd[me], (size_t)sc[me]);
  }
  if (!rcode && rtot && hipMemcpyAsync(recvbuf, g_hr, (size_t)rtot, hipMemcpyHostToDevice, st) != hipSuccess) rcode = -1;
  if (!rcode && hipStreamSynchronize(st) != hipSuccess) rcode = -1; /* the staging buffer is reused by the next call */
#endif
#undef EMI_MPI_EXCHANGE
  MPI_Type_free(&t16);
  free(isc);
  return rcode;
}

/* host collectives of DIST_x / GATH_x (emi_set_host_collectives): bytes over MPI in chunks of at most 1 GiB / int counts */
static int hc_bcast(void *user, void *buf, long long bytes, int root) {
  (void)user;
  for (long long off = 0; off < bytes; off += (1LL << 30)) {
    const long long n = bytes - off < (1LL << 30) ? bytes - off : (1LL << 30);
    if (MPI_Bcast((char *)buf + off, (int)n, MPI_BYTE, root, g_comm) != MPI_SUCCESS) return -1;
  }
  return 0;
}
static int hc_allgatherv(void *user, const void *sendbuf, long long sendbytes, void *recvbuf, const long long *recvbytes, const long long *displs,
                         int nproc) {
  (void)user;
  int me = 0;
  MPI_Comm_rank(g_comm, &me);
  /* one broadcast per task: counts of any size, no int overflow of displacements */
  for (int r = 0; r < nproc; r++) {
    if (r == me) memcpy((char *)recvbuf + displs[r], sendbuf, (size_t)sendbytes);
    if (hc_bcast(NULL, (char *)recvbuf + displs[r], recvbytes[r], r)) return -1;
  }
  return 0;
}

int emi_mpi_attach(MPI_Comm comm, int kmax_resol, int kprintlev, double prad, int device) {
  int rank = 0, size = 1, ndev = 0;
  MPI_Comm_dup(comm, &g_comm);
  MPI_Comm_rank(g_comm, &rank);
  MPI_Comm_size(g_comm, &size);
  if (device < 0) {
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return EMI_ERR_RUNTIME;
    device = rank % ndev;
  }
  emi_init_t cfg;
  memset(&cfg, 0, sizeof(cfg)); /* nprtrv = 0: emi_set_nprtrv's value, else 1 */
  cfg.kmax_resol = kmax_resol;
  cfg.kprintlev = kprintlev;
  cfg.prad = prad;
  cfg.nproc = size;
  cfg.myproc = rank + 1;
  cfg.device = device;
  int rc = emi_set_alltoallv(size > 1 ? hook : NULL, NULL);
  if (rc) return rc;
  rc = emi_set_host_collectives(hc_bcast, hc_allgatherv, NULL);
  if (rc) return rc;
  return emi_init(&cfg);
}

int emi_mpi_specnorm(MPI_Comm comm, int kresol, int mem_space, const void *spec, int nfld, double *norms) {
  int rc = emi_specnorm_partial(kresol, mem_space, spec, nfld, norms);
  if (rc) return rc;
  if (MPI_Allreduce(MPI_IN_PLACE, norms, nfld, MPI_DOUBLE, MPI_SUM, comm) != MPI_SUCCESS) return EMI_ERR_RUNTIME;
  for (int f = 0; f < nfld; f++) norms[f] = sqrt(norms[f]);
  return EMI_SUCCESS;
}

void emi_mpi_detach(void) {
  if (g_hs) hipHostFree(g_hs);
  if (g_hr) hipHostFree(g_hr);
  g_hs = g_hr = NULL;
  g_cap_s = g_cap_r = 0;
  if (g_comm != MPI_COMM_NULL) MPI_Comm_free(&g_comm);
  g_comm = MPI_COMM_NULL;
}
