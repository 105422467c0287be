/* emi_rccl_hook.c -- see emi_rccl_hook.h */
#include "emi_rccl_hook.h"

#include <hip/hip_runtime_api.h>
#include <math.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/ectrans_mi.h"

static ncclComm_t g_comm = NULL;
static int g_owned = 0, g_nproc = 1, g_me = 0;
static char g_err[512] = "";

#define RCCL_TRY(expr)                                                                                   \
  do {                                                                                                   \
    ncclResult_t r_ = (expr);                                                                            \
    if (r_ != ncclSuccess) {                                                                             \
      snprintf(g_err, sizeof(g_err), "%s failed: %s (%s:%d)", #expr, ncclGetErrorString(r_), __FILE__, __LINE__); \
      return EMI_ERR_RUNTIME;                                                                            \
    }                                                                                                    \
  } while (0)
#define HIP_TRY(expr)                                                                                    \
  do {                                                                                                   \
    hipError_t e_ = (expr);                                                                              \
    if (e_ != hipSuccess) {                                                                              \
      snprintf(g_err, sizeof(g_err), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return EMI_ERR_RUNTIME;                                                                            \
    }                                                                                                    \
  } while (0)

const char *emi_rccl_last_error(void) { return g_err; }

/* emi_alltoallv_fn: counts and displacements in BYTES, device pointers, ordered on `stream`.  One grouped set of
 * sends and receives: xGMI is fully connected, every block travels over its own link; the block a task keeps for
 * itself is a device-to-device copy on the same stream. */
int emi_rccl_alltoallv(void *user, const void *sendbuf, const long long *sc, const long long *sd, void *recvbuf, const long long *rc,
                       const long long *rd, int nproc, void *stream) {
  (void)user;
  hipStream_t st = (hipStream_t)stream;
  if (nproc != g_nproc || !g_comm) {
    snprintf(g_err, sizeof(g_err), "all-to-all-v over %d tasks on a communicator of %d", nproc, g_nproc);
    return -1;
  }
  if (sc[g_me] != rc[g_me]) {
    snprintf(g_err, sizeof(g_err), "own block: %lld bytes to send, %lld to receive", sc[g_me], rc[g_me]);
    return -1;
  }
  if (sc[g_me] > 0)
    HIP_TRY(hipMemcpyAsync((char *)recvbuf + rd[g_me], (const char *)sendbuf + sd[g_me], (size_t)sc[g_me], hipMemcpyDeviceToDevice, st));
  RCCL_TRY(ncclGroupStart());
  ncclResult_t bad = ncclSuccess;  /* a failed send / recv must not leave the group open: close it, then report */
  for (int r = 0; r < nproc && bad == ncclSuccess; r++) {
    if (r == g_me) continue;
    if (sc[r] > 0) bad = ncclSend((const char *)sendbuf + sd[r], (size_t)sc[r], ncclChar, r, g_comm, st);
    if (bad == ncclSuccess && rc[r] > 0) bad = ncclRecv((char *)recvbuf + rd[r], (size_t)rc[r], ncclChar, r, g_comm, st);
  }
  const ncclResult_t endr = ncclGroupEnd();
  if (bad != ncclSuccess || endr != ncclSuccess) {
    snprintf(g_err, sizeof(g_err), "grouped ncclSend / ncclRecv exchange failed: %s", ncclGetErrorString(bad != ncclSuccess ? bad : endr));
    return EMI_ERR_RUNTIME;
  }
  return 0;
}

/* host collectives of DIST_x / GATH_x (emi_set_host_collectives) over RCCL: the bytes are staged through device memory */
static int hc_bcast(void *user, void *buf, long long bytes, int root) {
  (void)user;
  if (g_nproc == 1 || bytes == 0) return 0;
  /* chunks of at most 256 MiB through one staging buffer, released on every path */
  const size_t chunk = (size_t)bytes < ((size_t)256 << 20) ? (size_t)bytes : ((size_t)256 << 20);
  void *d = NULL;
  HIP_TRY(hipMalloc(&d, chunk));
  int rc = 0;
  for (size_t off = 0; off < (size_t)bytes && !rc; off += chunk) {
    const size_t nb = (size_t)bytes - off < chunk ? (size_t)bytes - off : chunk;
    hipError_t e = hipSuccess;
    ncclResult_t r = ncclSuccess;
    if (root == g_me) e = hipMemcpy(d, (char *)buf + off, nb, hipMemcpyHostToDevice);
    if (e == hipSuccess) r = ncclBroadcast(d, d, nb, ncclChar, root, g_comm, (hipStream_t)0);
    if (e == hipSuccess && r == ncclSuccess) e = hipStreamSynchronize((hipStream_t)0);
    if (e == hipSuccess && r == ncclSuccess && root != g_me) e = hipMemcpy((char *)buf + off, d, nb, hipMemcpyDeviceToHost);
    if (e != hipSuccess || r != ncclSuccess) {
      snprintf(g_err, sizeof(g_err), "broadcast of %lld host bytes failed: %s", bytes, e != hipSuccess ? hipGetErrorString(e) : ncclGetErrorString(r));
      rc = -1;
    }
  }
  (void)hipFree(d);
  return rc;
}
static int hc_allgatherv(void *user, const void *sendbuf, long long sendbytes, void *recvbuf, const long long *recvbytes, const long long *displs,
                         int nproc) {
  (void)user;
  for (int r = 0; r < nproc; r++) {
    if (r == g_me) memcpy((char *)recvbuf + displs[r], sendbuf, (size_t)sendbytes);
    if (hc_bcast(NULL, (char *)recvbuf + displs[r], recvbytes[r], r)) return -1;
  }
  return 0;
}

int emi_rccl_get_unique_id(void *id) {
  ncclUniqueId u;
  if (!id) return EMI_ERR_ARG;
  if (sizeof(u) != EMI_RCCL_ID_BYTES) {
    snprintf(g_err, sizeof(g_err), "sizeof(ncclUniqueId) = %zu, expected %d", sizeof(u), EMI_RCCL_ID_BYTES);
    return EMI_ERR_RUNTIME;
  }
  RCCL_TRY(ncclGetUniqueId(&u));
  memcpy(id, &u, sizeof(u));
  return EMI_SUCCESS;
}

static int attach_common(int nproc, int myproc, int kmax_resol, int kprintlev, double prad, int device) {
  emi_init_t cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.kmax_resol = kmax_resol;
  cfg.kprintlev = kprintlev;
  cfg.prad = prad;
  cfg.nproc = nproc;
  cfg.myproc = myproc;
  cfg.device = device;
  g_nproc = nproc;
  g_me = myproc - 1;
  int rc = emi_set_alltoallv(nproc > 1 ? emi_rccl_alltoallv : NULL, NULL);
  if (rc) return rc;
  rc = emi_set_host_collectives(hc_bcast, hc_allgatherv, NULL);
  if (rc) return rc;
  return emi_init(&cfg);
}

static int pick_device(int myproc, int device, int *out) {
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (ndev <= 0) {
    snprintf(g_err, sizeof(g_err), "no HIP device visible");
    return EMI_ERR_RUNTIME;
  }
  *out = device >= 0 ? device : (myproc - 1) % ndev;
  HIP_TRY(hipSetDevice(*out));
  return 0;
}

int emi_rccl_attach(const void *id, int nproc, int myproc, int kmax_resol, int kprintlev, double prad, int device) {
  if (!id || nproc < 1 || myproc < 1 || myproc > nproc) {
    snprintf(g_err, sizeof(g_err), "emi_rccl_attach: bad arguments (nproc %d, myproc %d)", nproc, myproc);
    return EMI_ERR_ARG;
  }
  int dev = 0;
  if (pick_device(myproc, device, &dev)) return EMI_ERR_RUNTIME;
  ncclUniqueId u;
  memcpy(&u, id, sizeof(u));
  RCCL_TRY(ncclCommInitRank(&g_comm, nproc, u, myproc - 1));
  g_owned = 1;
  const int rc = attach_common(nproc, myproc, kmax_resol, kprintlev, prad, dev);
  if (rc) { /* emi_init refused: do not keep a communicator nobody will destroy */
    (void)ncclCommDestroy(g_comm);
    g_comm = NULL;
    g_owned = 0;
    (void)emi_set_alltoallv(NULL, NULL);
  }
  return rc;
}

int emi_rccl_attach_comm(void *nccl_comm, int nproc, int myproc, int kmax_resol, int kprintlev, double prad, int device) {
  if (!nccl_comm || nproc < 1 || myproc < 1 || myproc > nproc) return EMI_ERR_ARG;
  int dev = 0;
  if (pick_device(myproc, device, &dev)) return EMI_ERR_RUNTIME;
  g_comm = (ncclComm_t)nccl_comm;
  g_owned = 0;
  return attach_common(nproc, myproc, kmax_resol, kprintlev, prad, dev);
}

int emi_rccl_specnorm(int kresol, int mem_space, const void *spec, int nfld, double *norms) {
  int rc = emi_specnorm_partial(kresol, mem_space, spec, nfld, norms);
  if (rc) return rc;
  if (g_nproc > 1) {
    double *d = NULL;
    HIP_TRY(hipMalloc((void **)&d, sizeof(double) * (size_t)nfld));
    hipError_t e = hipMemcpy(d, norms, sizeof(double) * (size_t)nfld, hipMemcpyHostToDevice);
    ncclResult_t r = ncclSuccess;
    if (e == hipSuccess) r = ncclAllReduce(d, d, (size_t)nfld, ncclDouble, ncclSum, g_comm, (hipStream_t)0);
    if (e == hipSuccess && r == ncclSuccess) e = hipStreamSynchronize((hipStream_t)0);
    if (e == hipSuccess && r == ncclSuccess) e = hipMemcpy(norms, d, sizeof(double) * (size_t)nfld, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess || r != ncclSuccess) {
      snprintf(g_err, sizeof(g_err), "SPECNORM all-reduce failed: %s", e != hipSuccess ? hipGetErrorString(e) : ncclGetErrorString(r));
      return EMI_ERR_RUNTIME;
    }
  }
  for (int f = 0; f < nfld; f++) norms[f] = sqrt(norms[f]);
  return EMI_SUCCESS;
}

int emi_rccl_detach(void) {
  if (g_comm && g_owned) RCCL_TRY(ncclCommDestroy(g_comm));
  g_comm = NULL;
  g_owned = 0;
  g_nproc = 1;
  g_me = 0;
  return EMI_SUCCESS;
}
