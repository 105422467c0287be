// emi_types.h -- device-visible descriptors shared by host code and both kernel precisions.
#pragma once
#include "emi_rt.h"

// ------------------------------------------------------------------------------------------
// device-visible descriptors
// ------------------------------------------------------------------------------------------
struct EmiGeomDev {
  // Everything is indexed by LOCAL zonal-wavenumber number ml (0..nump-1, actual wavenumber
  // mval[ml]) and LOCAL latitude number (0..nlat-1): with one task local == global; with several
  // tasks each task owns the wavenumbers of its W-set (suwavedi_mod.F90:118-137) and a contiguous
  // latitude band.
  int nsmax, nump, nlat, ngptot;
  int m0_wide;  // fp32 library: Legendre transforms of zonal wavenumber 0 accumulate in double (ledir_mod.F90:133-171)
  const int *mval;    // [nump] actual zonal wavenumber
  const int *nmen, *gpoff;  // [nlat]
  const int *nasm0;   // [nump] 0-based index of Re(m, n=m) in the (local) user spectral dimension
  const int *fbase;   // [nlat+1]  Fourier rows (lat, m<=NMEN) before the local latitude
  const int *fftrow;  // [fbase[nlat]] row of (lat, m) in the FFT-side Fourier buffer; NULL: one task, row = fbase[lat] + m
  const int *lbase;   // [nump+1] start of wavenumber ml in legN/legS
  const int *legN, *legS;  // row of (ml, j-th northern latitude with m<=NMEN) / its southern mirror
                           // in the Legendre-side Fourier buffer
  const int *wbase;   // [nump+1] packed-spectral rows before ml (padded to 16)
  const int *wrows;   // [nump] padded row count (multiple of 16)
  const int *rowm;    // [wbase[nump]] row -> ml
  const int *ebase;   // [nump] index of eps(n=m) in eps[] (n = m..N+2)
  const double *eps;  // REPSNM
  const double *lapin;  // RLAPIN(n) at [n+1], n=-1..N+2
  const double *rw, *racthe;   // [nlat]
  const void *P;               // Legendre panels (real_t of the library precision)
  const long long *offS, *offA;  // [nump] element offsets of the even/odd (n-m) panels
  const int *ldp;              // [nump] padded latitude count (multiple of 64)
  const void *PT;              // transposed panels for the direct transform: [par][lat j][k], k fastest
  const long long *offTS, *offTA;  // [nump]
  const int *ldk;              // [nump] padded k count (multiple of 64)
  const int *lattile_pref;     // [nump+1] prefix of ceil(ndglu/64)
  const int *ktile_pref;       // [nump+1] k_leg_dir's row tiles: 2 floor(nk/128) one-parity tiles + 0 | 1 | 2 for the rest (nk = wrows/2)
  const double *specw;         // [nspec2 local] SPECNORM weight of every spectral entry (0, 1 or 2)
};

// k_vd2uv (VORDIV_TO_UV): spectral tables of one task, no resolution handle
struct Vd2uvDev {
  const int *pairm;     // [nspec2 / 2] local wavenumber number of every (Re, Im) pair
  const int *mval;      // [nump]
  const int *nasm0;     // [nump] 0-based
  const int *ebase;     // [nump] index of eps(n = m) in eps[] (n = m .. N+2)
  const double *eps, *lapin;  // REPSNM; RLAPIN(n) at [n + 1]
  int nsmax, nspec2, nfld;
  double rra;           // 1 / RA
};

struct LegPolDev {  // SETUP_TRANS on the device: inputs of k_legpol
  const double *mu;        // [ndgnh] Gaussian latitudes (sin) of the northern hemisphere, pole first
  const double *dcl, *ddl;  // [nump][nmax+1] recurrence coefficients of SUPOLF (supolf_mod.F90:79-83), index n
  const double *zfac;      // [nump] sqrt(2m-1) prod_{j<m} sqrt((2j-1)/(2j))
  const int *blk;          // [nblocks][2]: local wavenumber, parity << 16 | latitude tile (64 latitudes)
  int ndgnh, nmax;
};

enum { SPK_COPY = 0, SPK_U = 1, SPK_V = 2, SPK_NSD = 3, SPK_U_AD = 4, SPK_V_AD = 5 };  // *_AD: adjoint of UVTVD (DIR_TRANSAD)
struct SpecSrc {  // one Legendre-space input field of the inverse transform
  const void *a, *b;  // real_t arrays; element (ispec) of field = a[ispec*sa + ia]
  int sa, ia, sb, ib;
  int kind, pad_;
};
enum { SPO_COPY = 0, SPO_VOR = 1, SPO_DIV = 2, SPO_VOR_AD = 3, SPO_DIV_AD = 4, SPO_SC_AD = 5 };  // *_AD: INV_TRANSAD
struct SpecDst {  // one spectral output field of the direct transform
  void *dst;  // real_t array
  int stride, idx;
  int kind, src0, src1;  // src*: field index in W (U and V for vor/div; the scalar for SPO_COPY / SPO_SC_AD)
  // INV_TRANSAD with LDSCDERS / LDUVDER / LDVORGP / LDDIVGP (-1: absent).  SPO_VOR_AD / SPO_DIV_AD: src2, src3 = the
  // adjoint inputs of the east-west derivatives of U and V, src4 = the grid vorticity / divergence input.
  // SPO_SC_AD: src1 = the north-south derivative input (adjoint of SPNSDE, spnsdead_mod.F90), src2 = the east-west one.
  int src2, src3, src4;
};
struct FuseDst {  // k_leg_dir epilogue: where the coefficients of one W field go (dst == NULL: to W, for k_postpack_dir)
  void *dst;      // real_t array; element (ispec) of the field = dst[ispec * stride + idx]
  int stride, idx;
};
enum { GM_PLAIN = 0, GM_ACOS = 1, GM_EWDER = 2, GM_EWDER_UV = 3 };
struct GridFld {  // one Fourier-space field <-> one user grid field
  void *base;     // real_t array base; element (p) = base[((p/nproma)*nf_arr + fidx)*nproma + p%nproma]
  int nf_arr, fidx;
  int mode, src;  // src: field index inside FB (inverse only)
};

struct FftPlanDev {
  int n;      // row length (NLOEN)
  int sz;     // logical complex transform size: n/2 (n even) or n (cmode)
  int S;      // LDS work size per field (complex): sz, or Bluestein length L
  int cmode;  // 1: odd n, complex transform of the real row
  int blue;   // 1: Bluestein
  int nfac;
  int fac[14];
  int tw_off, perm_off, rtw_off, chirp_off, bhat_off;
  int ptw_off[14];  // per pass (DIT order): table [(t-1)*lenp + j] = exp(-2 pi i j t/(lenp*R))
  int fbk;  // fields per workgroup
  int lds_class;
  int mr;   // 1: direct mixed-radix kernels k_fft_*_mr: fac = A, B, C (B, C may be 1), ptw_off[0 / 1] the twiddles after pass 1 / 2, perm = LDS position of coefficient k
  int r16;  // > 0: register-resident kernels k_fft_*_r16<r16>, S = 256 r16; ptw_off[0]: the 7 x 256 digit twiddles, bhat in [k2][16 k0 + k1] order
  int split;  // 2: k_fft_*_r16p<r16> -- the row as two convolutions of half-length sz / 2 (chirp and filter tables of that length); else 0
};
struct FftTabDev {
  const void *tw;              // real2 tables of the library precision: e^{-2 pi i k/S}
  const void *ptw;             // per-pass twiddles, coalesced layout
  const unsigned short *perm;  // DIT input position of natural index
  const void *rtw;             // e^{-2 pi i k/n}, k=0..sz
  const void *chirp;           // e^{-i pi k^2/sz}
  const void *bhat;            // DFT_L of the chirp filter in DIT order, laid out [t][q] for the fused middle pass (q = butterfly, t = element)
  const void *tw256;           // [15][16]: exp(-2 pi i c k1 / 256), k1 = 1..15 (k_fft_*_r16)
  const FftPlanDev *plans;
  const int *planid;           // [ndgl]
};

// Everything a workgroup of the long-row FFT kernels needs to know about its latitude, in ONE 64-byte record per (launch group,
// latitude): the kernels used to walk lats[li] -> planid[lat] -> plans[..] -> table offsets and nmen / fbase / gpoff / racthe / rw[lat],
// four dependent scalar loads in front of the first vector load of a workgroup whose whole life is ~25 microseconds (round 4's clock
// stamps: ~12 % of it).  One s_load_dwordx16 now.
struct FftRowDev {
  int lat, planid, n, sz;
  int nmen, fb0, gpoff, chirp_off;
  int rtw_off, bhat_off, ptw_off0;
  int mr_abc;  // k_fft_*_mr: A | B << 8 | C << 16 | fields per workgroup << 24 (the second twiddle table follows the first: ptw_off0 + A)
  double racthe, rw;
};
// block -> (latitude, field chunk) through a per-class prefix table
struct FftLaunchDev {
  const int *lats;  // latitudes of this launch group (same workgroup size and fields per workgroup)
  int nlat;
  int nchunk;       // field chunks per latitude: block b works on latitude b / nchunk, chunk b % nchunk
  long long nblocks;
  int adj;  // adjoint transforms: k_fft_dir* drop the Gaussian weight and 1/NLOEN (INV_TRANSAD, ftinvad_mod.F90:77-83),
            // k_fft_inv* apply them (DIR_TRANSAD, ledirad_mod.F90:151,183 + ftdirad_mod.F90:84-89)
  const FftRowDev *rows;  // [nlat] the records of lats[] (k_fft_*_r16, k_fft_*_hot, k_fft_*_mr)
};


// ---- tile / LDS constants shared by both precisions
#define LG_THREADS 256
#define LG_BN 128
#define LG_LDA 80
#define LG_LDB 144
#define LG_LDS_BYTES ((2 * 8 * LG_LDA + 2 * 8 * LG_LDB) * 8)  // k_leg_inv, 8-row stages; sized for fp64 (fp32 uses half)
#define LG_LDS_BYTES_DIR (2 * LG_LDS_BYTES)                  // k_leg_dir, 16-row stages of its two-parity tile (the one-parity tile needs 2 x 16 x LG_LDB x 8)
#define FPAD(i) ((i) ^ (((i) >> 3) & 15))
#define FFT_LDS_ELEMS(S) (((S) + 15) & ~15)
// First radices R1 of the register-resident kernels (work length 256 R1).  R1 = 18 and 20 (work lengths 4608, 5120) were built and
// measured in round 3 and lost to the in-place LDS kernels (k_fft_inv_r16<18>, 20 bytes of spills: 18.0 ms against 15.0 ms for the
// rows of TCo1279; R1 = 20: 21.2 against 15.8): their first / last butterflies hold 72 / 80 data registers of the 128 a wave
// may use at four waves per SIMD, and 16 R1 threads make five-wave workgroups of which only three fit a CU.
#define EMI_R16_LIST(X) X(8) X(10) X(12) X(16)
// ... and of the split kernels k_fft_*_r16p<R1> (round 6): rows of 512 R1' < NLOEN <= 512 R1 as two convolutions of work length 256 R1
// (R1' = the entry before; 8 is not in the list because rows up to 4096 points fit one convolution of the list above).  fp64 TCo1279: R1 = 10
// carries the rows of 4100 .. 5120 points, R1 = 12 the eight longest; fp32 TCo2559: 10, 12 and 16 carry 4100 .. 8192.
#define EMI_R16S_LIST(X) X(10) X(12) X(16)
// radices of the direct mixed-radix kernels k_fft_*_mr (emi_mr_body.h)
#ifndef EMI_MR_RADICES
#define EMI_MR_RADICES(X) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(19) X(23)
// ... and of the fp32 library only: the long rows of TCo2559 (half-lengths above 16^3) need a radix above 16, and the composite ones
// give 16.3 instead of 10.9 % of that grid a direct plan (FFT phase 584 -> 568 ms).  In the fp64 kernels their butterflies spill and
// the TCo1279 rows they would add (1.5 % of the grid) cost what they cost on the convolution kernels -- measured, left out.
#define EMI_MR_RADICES_F32(X) X(18) X(20) X(21)
#endif

// Work lengths with a specialised FFT kernel (k_fft_*_hot<pc>): X(pc, S, nfac, factors[5], fields per
// workgroup).  The factor lists are what emi::factorize_smooth yields for S and the field count what the
// 40-KiB rule gives in fp64 (both checked when a plan is matched).  1-8: the rows that carry TCo1279;
// 9-12: the longer rows of TCo2559 (fp32: up to 10240 points fit the LDS); 13-21: the short rows, several
// fields per workgroup (most of TCo399); 22-27: 3072, 4608, 5120, 640, 576, 384 with their last two factors (2 3, 3 3, 2 5) merged into
// one composite radix (6, 9, 10: one LDS round trip fewer; the plain five-factor plans 3, 5, 6, 16, 17, 19 they replaced are gone).
// (Measured and rejected, round 2: nine intermediate work lengths 64 R3 R4 -- 1600, 1728, 1920, 2304, 2880, 3200, 3456,
// 3840, 5184 as 8, 8, R3, R4 -- cut the summed work length of TCo1279's long rows by 4.5 %, but their third pass is
// not wave-local and their lanes fill worse: FFT phase 197.9 ms against 197.4 ms without them.)
#define EMI_HOT_PLAN_LIST(X)          \
  X(1, 2048, 4, 8, 8, 8, 4, 1, 1)     \
  X(2, 2560, 4, 8, 8, 8, 5, 1, 1)     \
  X(4, 4096, 4, 8, 8, 8, 8, 1, 1)     \
  X(7, 1536, 4, 8, 8, 8, 3, 1, 1)     \
  X(8, 1280, 4, 8, 8, 4, 5, 1, 2)     \
  X(9, 6144, 5, 8, 8, 8, 4, 3, 1)     \
  X(10, 7680, 5, 8, 8, 8, 3, 5, 1)    \
  X(11, 8192, 5, 8, 8, 8, 8, 2, 1)    \
  X(12, 10240, 5, 8, 8, 8, 4, 5, 1)   \
  X(13, 1024, 4, 8, 8, 8, 2, 1, 2)    \
  X(14, 960, 4, 8, 8, 3, 5, 1, 2)     \
  X(15, 768, 4, 8, 8, 4, 3, 1, 2)     \
  X(18, 512, 3, 8, 8, 8, 1, 1, 4)     \
  X(20, 320, 3, 8, 8, 5, 1, 1, 8)     \
  X(21, 256, 3, 8, 8, 4, 1, 1, 8)     \
  X(22, 3072, 4, 8, 8, 8, 6, 1, 1)    \
  X(23, 4608, 4, 8, 8, 8, 9, 1, 1)    \
  X(24, 5120, 4, 8, 8, 8, 10, 1, 1)   \
  X(25, 640, 3, 8, 8, 10, 1, 1, 4)    \
  X(26, 576, 3, 8, 8, 9, 1, 1, 4)     \
  X(27, 384, 3, 8, 8, 6, 1, 1, 4)
