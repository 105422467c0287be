/*
 * ectrans_mi.h -- C-ABI of the MI355X-native spherical-harmonic transform (libectrans_mi.so).
 *
 * This is the drop-in boundary for the hot path SETUP_TRANS -> INV_TRANS / DIR_TRANS of
 * ecTrans 1.7.0.  Every entry point names the reference interface it replaces
 * (paths relative to /root/reference/src).  The Fortran shim (ectrans_amd/fortran) and the
 * transi-compatible C layer (ectrans_amd/transi) bind to exactly these symbols; see
 * INTEGRATION.md for the reference-side stubs.
 *
 * Conventions
 *  - plain pointers + sizes, no C++ / torch types; all functions return 0 on success and a
 *    negative code on error, with the text available from emi_last_error().  (The reference
 *    aborts the process through ABORT_TRANS, trans/common/internal/abort_trans_mod.F90:13-37;
 *    the Fortran shim turns a negative code into the same abort.)
 *  - array layouts are the Fortran ones, column-major:
 *      spectral  PSPEC(nfld, nspec2)               -> p[ispec*nfld + f]
 *      3-D spec  PSPSC3A(nlev, nspec2, nvar)       -> p[(v*nspec2 + ispec)*nlev + l]
 *      grid      PGP(nproma, nfld, ngpblks)        -> p[(blk*nfld + f)*nproma + i]
 *      grid 4-D  PGPUV(nproma, nlev, nvar, ngpblks)-> p[((blk*nvar + v)*nlev + l)*nproma + i]
 *  - pointers are HOST pointers when mem_space == EMI_MEM_HOST (staged over PCIe) and DEVICE (HBM)
 *    pointers when mem_space == EMI_MEM_DEVICE (used in place, zero-copy; what bench.py times).
 *    mem_space == EMI_MEM_AUTO: the library classifies every array of the call (hipPointerGetAttributes):
 *    all in device memory -> used in place, all in host memory -> staged, a mixture -> EMI_ERR_ARG (with several tasks
 *    and registered host collectives every task of the call fails together; without them the caller must make sure
 *    the outcome is the same on every task, or the others wait in the exchange).  This is
 *    what the Fortran shim and the transi layer pass, i.e. the reference GPU back-end's present-or-copyin
 *    treatment of its caller's arrays (trans/gpu/internal/trltog_mod.F90:501-523, trgtol_mod.F90:444-448,
 *    ltinv_mod.F90:334-338, updsp_mod.F90:96-97): a Fortran / C caller whose fields already live on the
 *    device (hipMalloc + C_F_POINTER, OpenMP use_device_addr, OpenACC host_data) gets the device-resident rate.
 *  - not thread-safe / not re-entrant, exactly like the reference (module-global state,
 *    trans/cpu/internal/tpm_trans.F90:28-56).
 */
#ifndef ECTRANS_MI_H
#define ECTRANS_MI_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EMI_MEM_HOST 0
#define EMI_MEM_DEVICE 1
#define EMI_MEM_AUTO 2
/* Where p lives: EMI_MEM_DEVICE for device (or managed) memory of any visible GPU, EMI_MEM_HOST for everything else (pageable,
 * pinned or registered host memory: it is staged).  The classification behind EMI_MEM_AUTO.                           */
int emi_ptr_space(const void *p);
/* Blocks the host until every call queued so far on resolution kresol has finished on the device (kresol <= 0: on every
 * resolution).  Calls with device-resident arrays return without waiting for their kernels (stream semantics); a host that has
 * no stream to synchronise -- the Fortran shim, the transi layer -- calls this before it returns to its caller.        */
int emi_wait(int kresol);

#define EMI_SUCCESS 0
#define EMI_ERR_ARG (-1)      /* what ABORT_TRANS would have reported about the arguments */
#define EMI_ERR_STATE (-2)    /* setup order / unknown resolution handle */
#define EMI_ERR_RUNTIME (-3)  /* HIP runtime failure */
#define EMI_ERR_UNSUPPORTED (-4)

/* ---- SETUP_TRANS0 (trans/common/external/setup_trans0.F90:13-229) ------------------- */
typedef struct {
  int kmax_resol;  /* KMAX_RESOL (default 1 when <= 0)                                    */
  int kprintlev;   /* KPRINTLEV                                                           */
  double prad;     /* PRAD, planet radius; <= 0 selects the reference default 6371229.0   */
  int nproc;       /* number of ranks sharing the zonal-wavenumber/latitude distribution  */
  int myproc;      /* 1-based rank (MYPROC)                                               */
  int device;      /* HIP device ordinal, -1: keep current                                */
  int nprtrv;      /* NPRTRV (sump_trans0_mod.F90:49): V-sets; 0 or 1: one.  nproc = NPRTRW x NPRTRV tasks, task
                    * myproc is (MYSETW, MYSETV) = ((myproc-1) / NPRTRV + 1, mod(myproc-1, NPRTRV) + 1) as PE2SET
                    * (pe2set_mod.F90:111-112).  0 also takes a value set earlier with emi_set_nprtrv (for hosts whose
                    * transport calls emi_init: emi_mpi_attach, emi_rccl_attach).                                    */
} emi_init_t;
int emi_set_nprtrv(int nprtrv);
int emi_init(const emi_init_t *cfg);

/* ---- SETUP_TRANS (trans/cpu/external/setup_trans.F90:11-434) ------------------------ */
typedef struct {
  int ksmax;         /* KSMAX spectral truncation                                          */
  int kdgl;          /* KDGL number of Gaussian latitudes (even)                           */
  const int *kloen;  /* KLOEN(kdgl) points per latitude, NULL: full grid with 2*KDGL       */
  int kdlon;         /* KDLON (used only when kloen == NULL and > 0)                       */
  int precision;     /* bytes per real of every data array of this resolution: 8 (or 0) =
                      * fp64, the reference's libtrans_dp (JPRB=JPRD); 4 = fp32, its
                      * libtrans_sp (JPRB=JPRM).  Setup arithmetic is double in both.       */
  /* options the reference GPU backend also refuses (gpu/external/setup_trans.F90:309,442):
   * a non-zero value returns EMI_ERR_UNSUPPORTED */
  int lduseflt, ldll, ldstretch;
  int lduserpnm;     /* LDUSERPNM: 1 = Belousov's generator for the Legendre polynomials (supol_mod.F90; the
                      * default of the Fortran API), 0 = the per-wavenumber recurrence SUPOLF (what the
                      * benchmark and transi pass; computed on the device).  The two agree to ~1e-12.       */
} emi_setup_t;
int emi_setup(const emi_setup_t *cfg, int *kresol);

/* SETUP_TRANS with its Legendre-polynomial I/O arguments CDIO_LEGPOL, CDLEGPOLFNAME, KLEGPOLPTR,
 * KLEGPOLPTR_LEN (setup_trans.F90:63-71, 360-384; one task only, as there).  The byte format is the
 * reference's (write_legpol_mod.F90:66-158, read_legpol_mod.F90:78-215), so files and memory segments
 * are interchangeable with libtrans_dp's: 'LEGPOL  ', NSMAX, NDGNH, (NLOEN, NMEN) per northern
 * latitude, then per wavenumber RPNMA(ndglu, (nsmax-m+2)/2) and RPNMS(ndglu, (nsmax-m+3)/2) as
 * 8-byte reals.  "readf"/"membuf": the panels are taken from the file / segment (label, truncation,
 * latitude count, NLOEN and NMEN are checked with the reference's messages); "writef": the panels this
 * setup computed are written out.  io == NULL or io->io == NULL: plain emi_setup.              */
typedef struct {
  const char *io;    /* CDIO_LEGPOL: "readf" | "writef" | "membuf" (or upper case)              */
  const char *fname; /* CDLEGPOLFNAME (readf, writef)                                          */
  const void *ptr;   /* KLEGPOLPTR (membuf): host memory holding a file image                  */
  size_t len;        /* KLEGPOLPTR_LEN, bytes                                                  */
} emi_legpol_io_t;
int emi_setup_legpol(const emi_setup_t *cfg, const emi_legpol_io_t *io, int *kresol);

/* ---- TRANS_INQ (trans/cpu/external/trans_inq.F90:11-529), subset used by callers ----- */
/* integer scalars: "nspec2" "nspec2g" "nspec2mx" "ngptot" "ngptotg" "ngptotmx" "nump" "ndgl" "nsmax"
 * "ndlon" "nproc" "myproc" "nfrstlat" "nlstlat"                                           */
int emi_inq_int(int kresol, const char *name, int *value);
/* integer arrays: "nloen"(ndgl) "nmen"(ndgl) "ndglu"(nsmax+1) "nasm0"(nsmax+1, 1-based as
 * D%NASM0, -99 for wavenumbers of other tasks) "myms"(nump) "procm"(nsmax+1) "latlo"(nproc+1) */
int emi_inq_int_array(int kresol, const char *name, int *out, int len);
/* real arrays: "rmu"/"pmu"(ndgl) "rgw"/"pgw"(ndgl)                                        */
int emi_inq_real_array(int kresol, const char *name, double *out, int len);
/* Legendre panel of zonal wavenumber m as the reference stores it, S%FA(m)%RPNMA/RPNMS
 * (trans/cpu/internal/suleg_mod.F90:609-615,891-897): column-major (ndglu x ncols), n
 * descending.  out == NULL only returns the sizes.                                        */
int emi_inq_legendre(int kresol, int m, int symmetric, double *out, int *nrows, int *ncols);

/* Extents (UBOUND) of the caller's arrays, so that the library can make the reference's own extent checks
 * (inv_trans.F90:476-600, dir_trans.F90:370-491: 'SEC. DIMENSION OF PGPUV INCONSISTENT', 'THIRD DIMENSION OF PGPUV
 * TOO SMALL', 'SEC. DIMENSION OF PGP2 INCONSISTENT', ...) before any kernel touches them.  Arrays that are absent
 * have zero extents.  The leading extent of every grid array must EQUAL NPROMA at this boundary (the reference
 * accepts a larger one; the Fortran shim then passes a packed copy); a PGPUV with more variables than the call
 * produces (third extent > IF_UV_PAR) is addressed with its real extent.  emi_invtrans_t.ext / emi_dirtrans_t.ext
 * == NULL: the caller vouches for the sizes (device-resident callers that computed them from TRANS_INQ).       */
typedef struct {
  int sp_dim2;   /* smallest NSPEC2-extent over the spectral arrays present (dimension 2 of all of them)      */
  int gp[3];     /* UBOUND(PGP)   = nproma, fields, ngpblks                                                     */
  int gpuv[4];   /* UBOUND(PGPUV) = nproma, levels, variables, ngpblks                                          */
  int gp3a[4];   /* UBOUND(PGP3A)                                                                               */
  int gp3b[4];   /* UBOUND(PGP3B)                                                                               */
  int gp2[3];    /* UBOUND(PGP2)  = nproma, fields, ngpblks                                                     */
} emi_extents_t;

/* ---- KVSETUV / KVSETSC / KVSETSC2 / KVSETSC3A / KVSETSC3B (inv_trans.h:84-101, dir_trans.h:75-92) -------------------
 * With NPRTRV > 1 the fields are dealt to the V-sets in spectral (and Fourier) space: kvset*[f] (1..NPRTRV) names the V-set
 * of GLOBAL field f, the spectral arrays of a call hold only the fields of this task's V-set (in global order), the grid
 * arrays hold ALL fields (n*_g of them; levels for the 3-D arrays) on this task's grid points (inv_trans.F90:212-300).  Every
 * group that is present needs its array; NULL block = one V-set.                                                     */
typedef struct {
  const int *kvsetuv;   int nuv_g;    /* KVSETUV(nuv_g)                                            */
  const int *kvsetsc;   int nsc_g;    /* KVSETSC(nsc_g)                                            */
  const int *kvsetsc2;  int nsc2_g;   /* KVSETSC2(nsc2_g)                                          */
  const int *kvsetsc3a; int nsc3a_g;  /* KVSETSC3A(nsc3a_g): V-set of every LEVEL of PSPSC3A       */
  const int *kvsetsc3b; int nsc3b_g;
  /* Variables in PGP3A / PGP3B: their third extent (/ 3 with LDSCDERS).  The number of variables itself (the reference's
   * IF_SC3A_G3 = UBOUND(PSPSC3A,3), inv_trans.F90:277) is sc3a_nvar / sc3b_nvar of the call wherever a task names it -- a task whose V-set owns
   * no level passes it with sc3a_nlev = 0, as the reference's zero-level array -- and must be the same on every task.  The grid arrays must hold
   * exactly that many: any other value fails with 'THIRD DIMENSION OF PGP3A INCONSISTENT' (inv_trans.F90:557, :587, dir_trans.F90:451, :481).
   * 0: not stated (the call's count is used); a task that names no spectral array (sc3a_nvar = 0) takes the count from here.               */
  int nvar3a_g, nvar3b_g;
} emi_vsets_t;

/* ---- INV_TRANS (trans/include/ectrans/inv_trans.h:12-163) --------------------------- */
typedef struct {
  int mem_space;
  /* spectral inputs (NULL = absent) */
  const void *spvor, *spdiv;  /* PSPVOR/PSPDIV(nf_uv, nspec2)                              */
  int nf_uv;
  const void *spscalar;       /* PSPSCALAR(nf_scalar, nspec2)                              */
  int nf_scalar;
  const void *spsc3a;         /* PSPSC3A(sc3a_nlev, nspec2, sc3a_nvar)                     */
  int sc3a_nlev, sc3a_nvar;
  const void *spsc3b;
  int sc3b_nlev, sc3b_nvar;
  const void *spsc2;          /* PSPSC2(nf_sc2, nspec2)                                    */
  int nf_sc2;
  int ldscders, ldvorgp, lddivgp, lduvder; /* LDSCDERS, LDVORGP, LDDIVGP, LDUVDER          */
  int kproma;                 /* KPROMA, <= 0: NGPTOT                                      */
  /* grid outputs */
  void *gp;                   /* PGP(nproma, gp_nfld, ngpblks)                             */
  int gp_nfld;                /* second extent of PGP (>= IF_GP)                           */
  void *gpuv;                 /* PGPUV(nproma, nf_uv, nvar_uv, ngpblks)                    */
  void *gp3a, *gp3b;          /* PGP3A(nproma, nlev, nvar[*3], ngpblks)                    */
  void *gp2;                  /* PGP2(nproma, nf_sc2[*3], ngpblks)                         */
  void *stream;               /* hipStream_t, NULL: default stream.  Calls on one resolution handle are
                               * serialised by the library whatever their streams (they share its work
                               * buffers): a call waits, on the device, for the previous call of that handle */
  const emi_extents_t *ext;   /* extents of the arrays above, NULL: unchecked                */
  const emi_vsets_t *vsets;   /* NPRTRV > 1: the V-set of every global field (nf_uv ... then count LOCAL fields) */
} emi_invtrans_t;
int emi_inv_trans(int kresol, const emi_invtrans_t *args);

/* ---- DIR_TRANS (trans/include/ectrans/dir_trans.h:12-140) --------------------------- */
typedef struct {
  int mem_space;
  void *spvor, *spdiv;
  int nf_uv;
  void *spscalar;
  int nf_scalar;
  void *spsc3a;
  int sc3a_nlev, sc3a_nvar;
  void *spsc3b;
  int sc3b_nlev, sc3b_nvar;
  void *spsc2;
  int nf_sc2;
  int kproma;
  const void *gp;    /* PGP(nproma, gp_nfld, ngpblks): u(nf_uv) v(nf_uv) scalars           */
  int gp_nfld;
  const void *gpuv;  /* PGPUV(nproma, nf_uv, 2, ngpblks)                                   */
  const void *gp3a, *gp3b, *gp2;
  void *stream;
  const emi_extents_t *ext;
  const emi_vsets_t *vsets;
} emi_dirtrans_t;
int emi_dir_trans(int kresol, const emi_dirtrans_t *args);

/* ---- INV_TRANSAD / DIR_TRANSAD (trans/include/ectrans/inv_transad.h:12, dir_transad.h:12) ------
 * Adjoints with respect to the inner products of the reference's own adjoint tests
 * (tests/trans/test_invtrans_adjoint.F90:243-315): plain sum over grid points; in spectral space the
 * SPECNORM weights (1 for m = 0, 2 for m > 0, imaginary parts of m = 0 excluded).  Same argument
 * blocks as the transforms they are the adjoints of, with the intents swapped: emi_inv_transad READS
 * the gp* arrays and WRITES the sp* arrays (it overwrites them; the reference adds to them, callers
 * zero them first -- test_invtrans_adjoint.F90:192-198), emi_dir_transad reads sp*, writes gp*.
 * emi_inv_transad takes ldscders / ldvorgp / lddivgp / lduvder like INV_TRANSAD (inv_transad.h): the extra grid fields
 * (same places as in INV_TRANS's output) are further inputs whose contributions are added to the spectral fields they
 * derive from (spnsdead_mod.F90, fscad_mod.F90, vdtuvad_mod.F90).                                                    */
int emi_inv_transad(int kresol, const emi_invtrans_t *args);
int emi_dir_transad(int kresol, const emi_dirtrans_t *args);

/* ---- SPECNORM (trans/include/ectrans/specnorm.h:12) --------------------------------- */
int emi_specnorm(int kresol, int mem_space, const void *spec, int nfld, double *norms /* host */);

/* SPECNORM with KVSET (NPRTRV > 1): spec holds the nfld fields of this task's V-set, norms_g (host) receives the norms of all
 * nfld_g fields, kvset[f] = V-set of global field f.  Collective over all tasks (host collectives).                   */
int emi_specnorm_kvset(int kresol, int mem_space, const void *spec, int nfld, const int *kvset, int nfld_g, double *norms_g);
/* NPRTRW, NPRTRV, MYSETW, MYSETV (sump_trans0_mod.F90:49, pe2set_mod.F90:111-112) of the initialised library                */
int emi_inq_vsets(int *nprtrw, int *nprtrv, int *mysetw, int *mysetv);

/* ---- TRLTOM / TRMTOL (trans/cpu/internal/trltom_mod.F90:96-136, trmtol_mod.F90:101-141) ---------
 * With nproc > 1 every task owns the zonal wavenumbers of its W-set (zig-zag, suwavedi_mod.F90:118-137)
 * and a contiguous latitude band; each transform needs ONE all-to-all-v of whole blocks of the
 * device-resident Fourier buffer.  The host supplies it as a hook (RCCL through torch.distributed,
 * an MPI_Alltoallv on GPU-aware MPI, ...): counts and displacements are in BYTES, one entry per
 * task; buffers are device pointers; the call must be ordered after the work already queued on
 * `stream` and must complete (or be stream-ordered) before work queued on it afterwards.         */
typedef int (*emi_alltoallv_fn)(void *user, const void *sendbuf, const long long *sendcounts, const long long *sdispls,
                                void *recvbuf, const long long *recvcounts, const long long *rdispls, int nproc, void *stream);
int emi_set_alltoallv(emi_alltoallv_fn fn, void *user);
/* This task's share of the SPECNORM sums (spnormd_mod.F90); sum over tasks, then sqrt.          */
int emi_specnorm_partial(int kresol, int mem_space, const void *spec, int nfld, double *sumsq /* host */);

/* ---- DIST_SPEC / GATH_SPEC / DIST_GRID / GATH_GRID (trans/include/ectrans/dist_spec.h, gath_spec.h, dist_grid.h,
 * gath_grid.h; trans/cpu/internal/dist_spec_control_mod.F90, gath_spec_control_mod.F90, dist_grid_ctl_mod.F90,
 * gath_grid_ctl_mod.F90) -- global fields on single tasks <-> the distributed arrays of the transforms.  HOST arrays
 * of the resolution's precision, not on the transform path.  Global spectral order: m = 0..N, n = m..N, (re, im)
 * (dist_spec_control_mod.F90:158-161); global grid order: latitudes north to south = the tasks' bands in task order.
 *   specg / gpg : the global fields this task is the source (kfrom[f] == myproc) or the target (kto[f] == myproc) of,
 *                 one after the other in field order: [n_mine][nspec2g] / [n_mine][ngptotg]  (= PGPG(ngptotg, n_mine))
 *   spec        : PSPEC(nfld, nspec2);   gp : PGP(nproma, nfld, ngpblks), elements past NGPTOT are left alone
 *   kfrom / kto : task (1-based) of every field; ksort (or NULL): field f lands in slot ksort[f] (1-based) of spec / gp
 * With several tasks the host supplies two collectives over its tasks (an MPI or RCCL transport registers them, see
 * ectrans_amd/mpi and ectrans_amd/rccl): a broadcast and an all-gather-v of host bytes.                              */
typedef int (*emi_bcast_fn)(void *user, void *buf, long long bytes, int root /* 0-based */);
typedef int (*emi_allgatherv_fn)(void *user, const void *sendbuf, long long sendbytes, void *recvbuf, const long long *recvbytes,
                                 const long long *displs, int nproc);
int emi_set_host_collectives(emi_bcast_fn bcast, emi_allgatherv_fn allgatherv, void *user);
int emi_dist_spec(int kresol, const void *specg, int nfld, const int *kfrom, const int *ksort, void *spec);
int emi_gath_spec(int kresol, void *specg, int nfld, const int *kto, const void *spec);
int emi_dist_grid(int kresol, const void *gpg, int nfld, const int *kfrom, const int *ksort, int kproma, void *gp);
int emi_gath_grid(int kresol, void *gpg, int nfld, const int *kto, int kproma, const void *gp);
/* What emi_init was given (EMI_ERR_STATE before it): a Fortran host whose transport attached first (emi_mpi_attach,
 * emi_rccl_attach) learns its task number from here in SETUP_TRANS0.                                                 */
int emi_inq_tasks(int *nproc, int *myproc);
/* KMAX_RESOL and PRAD the library was initialised with (setup_trans0.F90:113-129): SETUP_TRANS0 / trans_init of a host whose
 * transport called emi_init first compare their own arguments with these and abort on a mismatch instead of silently
 * computing with the transport's planet radius.                                                                      */
int emi_inq_init(int *kmax_resol, double *prad);

/* ---- TRANS_RELEASE / TRANS_END (trans/cpu/external/trans_release.F90, trans_end.F90) -- */
int emi_release(int kresol);
int emi_finalize(void);
/* Frees the idle device staging buffers that host-array calls (mem_space = EMI_MEM_HOST) keep between calls; also done by
 * emi_release and emi_finalize.  EMI_STAGE_POOL=0 in the environment keeps no buffers at all.                          */
int emi_trim_cache(void);

const char *emi_last_error(void);

/* ---- GPNORM_TRANS (trans/include/ectrans/gpnorm_trans.h:12, cpu/internal/gpnorm_trans_ctl_mod.F90) ---------------------------
 * Area-weighted average, minimum and maximum of the first `kfields` fields of PGP(nproma, gp_nfld, ngpblks) on this task's grid
 * points: per latitude the sum over the longitudes (in double) x RW(lat) / NLOEN(lat), summed over ALL latitudes of the sphere
 * in latitude order (every task gets the result; the reference fills it on task 1).  ave_only != 0 (LDAVE_ONLY): pmin / pmax hold
 * the caller's local extrema on entry and are only reduced over the tasks.  kproma <= 0: NGPTOT.                              */
int emi_gpnorm(int kresol, int mem_space, const void *gp, int gp_nfld, int kfields, int kproma, double *ave, double *pmin, double *pmax,
               int ave_only);

/* ---- VORDIV_TO_UV (trans/include/ectrans/vordiv_to_uv.h:12, cpu/internal/vd2uv_mod.F90:79-120) --------------------------------
 * Spectral vorticity / divergence PSPVOR / PSPDIV(nfld, nspec2) -> spectral U = u cos(theta), V = v cos(theta) in PSPU / PSPV, for
 * the zonal wavenumbers of this task's W-set and total wavenumbers n <= ksmax.  Needs emi_init only (the reference builds and
 * releases a spectral-only resolution inside the call); precision = bytes per real of the four arrays (8 or 4).
 * nspec2 = second extent of the four arrays (UBOUND(PSPVOR,2), transi's ncoeff): the call needs (and touches) the spectral
 * coefficients of this task's wavenumbers at truncation ksmax and fails with EMI_ERR_ARG when the arrays are shorter.        */
int emi_vordiv_to_uv(int ksmax, int precision, int mem_space, const void *spvor, const void *spdiv, void *spu, void *spv, int nfld, int nspec2);

/* Broadcast of host bytes from task `root` (1-based) over the attached transport (emi_set_host_collectives): for the layers above,
 * where the reference sends a small array from its master task (GPNORM_TRANSAD, gpnorm_trans_ctlad_mod.F90:108-113).  */
int emi_bcast_host(void *buf, long long bytes, int root);

/* ---- measurement hooks (no reference counterpart; used by bench.py / profiles) -------- */
/* Algorithmic work of one call on this resolution: Legendre flops and Fourier/grid bytes
 * for `nfields` Fourier-space fields.                                      */
int emi_work_model(int kresol, int nfields, double *legendre_flops_per_direction,
                   double *fft_flops_per_direction, double *fourier_bytes_per_direction);
/* Per-phase device time (ms) of the last emi_inv_trans/emi_dir_trans call when
 * EMI_PROFILE=1: [0] pack/unpack spectral, [1] Legendre MFMA, [2] FFT.  Returns 0.        */
int emi_last_phase_ms(double *ms3);
/* Number of Legendre/FFT/pack phase intervals (one per field batch) behind those sums.      */
int emi_last_phase_launches(int *l3);
/* The exchanges (TRMTOL / TRLTOM; with V-sets also TRLTOG / TRGTOL) of the calls since emi_set_profile: device time between an event in
 * front of and one behind the all-to-all-v hook on the stream the hook was given (ms), their number, the bytes this task sent to
 * its peers.  With the field batches of a call pipelined the exchange of one batch runs beside the kernels of its neighbours, so
 * this time overlaps the phase times above (bench.py: exchange_ms_per_step, overlap_frac).  NULL pointers are skipped.          */
int emi_last_exchange(double *ms, int *calls, double *bytes_sent);
/* Kernel launches of the FFT phases since emi_set_profile (one FTINV / FTDIR of a field batch = one launch per length class). */
int emi_last_fft_launches(long long *kernels);
/* HIP-event phase timers (default: env EMI_PROFILE): 0 off; 1 per call (emi_last_phase_ms = the last call);
 * 2 accumulated over all calls since this emi_set_profile(2) (nothing is resolved, hence nothing
 * synchronises, between the calls of a timed loop; up to 4096 intervals).                   */
int emi_set_profile(int on);
/* 1 when the roctx ranges are live: INV_TRANS / DIR_TRANS / LTINV_CTL / LTDIR_CTL / FTDIR_CTL / FTINV_CTL / the two
 * transpositions / SETUP_TRANS / SULEG, with the GSTATS labels of the reference harness (gpu/internal/tpm_stats.F90:33-55 maps
 * GSTATS to NVTX ranges; src/programs/ectrans-benchmark.F90:1681-1697), around the host-side enqueue of each phase.
 * librocprofiler-sdk-roctx is opened at run time; EMI_ROCTX=0 switches the ranges off.       */
int emi_roctx_active(void);
/* Upper bound on Fourier-space fields per batch (0: from free HBM).                         */
int emi_set_max_batch(int max_fields);

/* ---- checksum dumps of the benchmark harness (src/programs/ectrans-benchmark.F90:1455-1600) ----------
 * CRC-64 of `bytes` bytes continued from *crc (the harness carries one value through the fields of an array).
 * The reference calls fiat's `crc64`, which is not vendored under /root/reference; this is CRC-64/ECMA-182
 * (polynomial 0x42F0E1EBA9EA3693, no reflection, no final xor).  What the reference's test checks is that the
 * dump of one decomposition is byte-identical to the dump of another (tests/compare_checksums.py:11-60), which
 * holds for any CRC; host memory only.                                                                    */
int emi_crc64(const void *data, size_t bytes, unsigned long long *crc);

#ifdef __cplusplus
}
#endif
#endif
