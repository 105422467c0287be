// emi_kernels.h -- device kernels of the MI355X spherical-harmonic transform.
//
// Pipeline (one field batch; see DESIGN.md §3 for layouts):
//   INV:  k_prepack_inv  (PRFI1B+VDTUV+SPNSDE)  user spectral -> W  [m][r=n-m][f][c]
//         k_leg_inv      (LEINV+ASRE1B, fp64 MFMA)          W -> FB [lat][m][f][c]
//         k_fft_inv      (FOURIER_IN+FSC+FTINV+TRLTOG)      FB -> user grid arrays
//   DIR:  k_fft_dir      (TRGTOL+FTDIR+FOURIER_OUT, w/NLOEN & 1/(a cos) folded in)  grid -> FB
//         k_leg_dir      (PRFI2B+LEDIR, fp64 MFMA)          FB -> W
//         k_postpack_dir (UVTVD+UPDSP)                      W -> user spectral
// Reference routines named in () are the CPU routines whose results each kernel reproduces
// (cpu/internal/*_mod.F90); none of the reference's GPU code is used.
#pragma once
#include "emi_rt.h"

#include "emi_types.h"
#include <type_traits>
#include "emi_mr_tables.h"
#if defined(EMI_MR_STAMP) && !defined(EMI_CPU_EMU)
__device__ unsigned long long emi_mr_stamp[16];
#endif

// fp64 library (the _dp build of the reference) and fp32 library (_sp): same kernel source, the
// matrix-core instruction and the accumulator row map differ
// (v_mfma_f64_16x16x4_f64: row = (lane>>4) + 4*i ; v_mfma_f32_16x16x4_f32: row = 4*(lane>>4) + i).
namespace emi_f64 {
#define EMI_REAL double
#define EMI_REAL2 d2
#define EMI_ACC4 v4d
#define EMI_MFMA emi_mfma_f64
#define EMI_ACC_ROW(l, i) (((l) >> 4) + 4 * (i))
#define EMI_FFT_WAVES 4
#define EMI_MR_EXTRA(X)
// k_leg_inv: waves per SIMD, and whether the mean wavenumber in double has a kernel of its own (k_leg_inv_wide)
#define EMI_LEG_INV_WAVES 2
#define EMI_LEG_DIR_WAVES 2
#define EMI_LEG_WIDE_KERNEL 0
#include "emi_kernels_body.h"
#undef EMI_MR_EXTRA
#undef EMI_REAL
#undef EMI_REAL2
#undef EMI_ACC4
#undef EMI_MFMA
#undef EMI_ACC_ROW
#undef EMI_FFT_WAVES
#undef EMI_LEG_INV_WAVES
#undef EMI_LEG_DIR_WAVES
#undef EMI_LEG_WIDE_KERNEL
}  // namespace emi_f64

namespace emi_f32 {
#define EMI_REAL float
#define EMI_REAL2 f2
#define EMI_ACC4 v4f
#define EMI_MFMA emi_mfma_f32
#define EMI_ACC_ROW(l, i) (4 * ((l) >> 4) + (i))
#define EMI_FFT_WAVES 4
#ifdef EMI_MR_RADICES_F32
#define EMI_MR_EXTRA(X) EMI_MR_RADICES_F32(X)
#else
#define EMI_MR_EXTRA(X)
#endif
// fp32: the accumulators of k_leg_inv are 64 registers, not 128 -- without the double-precision tiles of the mean wavenumber in the same
// kernel (LegAcc<true>: 128 registers of accumulators again) it fits three waves per SIMD in 136 registers: 47.0 -> 45.0 ms at TCo1279
// (four waves: 128 registers + 32 bytes of scratch, 45.0).  k_leg_dir likewise (154 registers), with 8-byte loader vectors: emi_kernels_body.h, LG_LS.
#define EMI_LEG_INV_WAVES 3
#define EMI_LEG_DIR_WAVES 3
#define EMI_LEG_WIDE_KERNEL 1
#include "emi_kernels_body.h"
#undef EMI_MR_EXTRA
#undef EMI_REAL
#undef EMI_REAL2
#undef EMI_ACC4
#undef EMI_MFMA
#undef EMI_ACC_ROW
#undef EMI_FFT_WAVES
#undef EMI_LEG_INV_WAVES
#undef EMI_LEG_DIR_WAVES
#undef EMI_LEG_WIDE_KERNEL
}  // namespace emi_f32
