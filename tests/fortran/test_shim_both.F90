! Both precisions in ONE executable, as IFS links trans_dp beside trans_sp (ecTrans 1.7.0: src/trans/CMakeLists.txt:43-93): the program
! links libectrans_mi_f.so AND libectrans_mi_f_sp.so, takes the interface blocks of the suffixed entry points from
! ectrans_amd/fortran/include/*_dp.h / *_sp.h, and runs a real64 and a real32 resolution side by side -- SETUP_TRANS0 once (the common
! library), SETUP_TRANS_DP -> handle 1, SETUP_TRANS_SP -> handle 2 (handles are numbered across the precisions, as NDEF_RESOL / NCUR_RESOL
! of the reference's common TPM_GEN).  Checks: the dp round trip to 1e-11, the sp fields against the dp fields to float accuracy, the
! norms of both, and that each flavour refuses the other's resolution handle.  Exit code 0 = pass.
PROGRAM TEST_SHIM_BOTH
USE, INTRINSIC :: ISO_C_BINDING, ONLY : C_INT32_T, C_FLOAT, C_DOUBLE
IMPLICIT NONE
#include "setup_trans0.h"
#include "get_current.h"
#include "setup_trans_dp.h"
#include "setup_trans_sp.h"
#include "trans_inq_dp.h"
#include "trans_inq_sp.h"
#include "inv_trans_dp.h"
#include "inv_trans_sp.h"
#include "dir_trans_dp.h"
#include "dir_trans_sp.h"
#include "specnorm_dp.h"
#include "specnorm_sp.h"
#include "trans_release_dp.h"
#include "trans_release_sp.h"
#include "trans_end_dp.h"
INTEGER(C_INT32_T), PARAMETER :: NSMAX = 47, NDGL = 2*(NSMAX+1), NLEV = 3
INTEGER(C_INT32_T) :: NLOEN(NDGL), I, NSPEC2, NSPEC2S, NGPTOT, NGPTOTS, IRD, IRS, ICUR, NASM0(0:NSMAX), I419, JM
REAL(C_DOUBLE), ALLOCATABLE :: ZSPD(:,:), ZSPD2(:,:), ZGPD(:,:,:)
REAL(C_FLOAT), ALLOCATABLE :: ZSPS(:,:), ZSPS2(:,:), ZGPS(:,:,:)
REAL(C_DOUBLE) :: ZND(NLEV), ZERR
REAL(C_FLOAT) :: ZNS(NLEV)
DO I = 1, NSMAX+1
  NLOEN(I) = 20+4*(I-1)
  NLOEN(NDGL+1-I) = NLOEN(I)
ENDDO
CALL SETUP_TRANS0(KPRINTLEV=0, LDMPOFF=.TRUE., KMAX_RESOL=4)
CALL SETUP_TRANS_DP(KSMAX=NSMAX, KDGL=NDGL, KLOEN=NLOEN, LDUSERPNM=.FALSE., KRESOL=IRD)
CALL SETUP_TRANS_SP(KSMAX=NSMAX, KDGL=NDGL, KLOEN=NLOEN, LDUSERPNM=.FALSE., KRESOL=IRS)
IF (IRD /= 1 .OR. IRS /= 2) ERROR STOP 2
CALL GET_CURRENT(KRESOL=ICUR)
IF (ICUR /= 1) ERROR STOP 3            ! the default resolution is the first one defined, whichever library defined it
CALL TRANS_INQ_DP(KRESOL=IRD, KSPEC2=NSPEC2, KGPTOT=NGPTOT, KASM0=NASM0)
CALL TRANS_INQ_SP(KRESOL=IRS, KSPEC2=NSPEC2S, KGPTOT=NGPTOTS)
IF (NSPEC2 /= NSPEC2S .OR. NGPTOT /= NGPTOTS) ERROR STOP 4
ALLOCATE(ZSPD(NLEV,NSPEC2), ZSPD2(NLEV,NSPEC2), ZGPD(NGPTOT,NLEV,1), ZSPS(NLEV,NSPEC2), ZSPS2(NLEV,NSPEC2), ZGPS(NGPTOT,NLEV,1))
! a dense, decaying spectrum (imaginary parts of m = 0 zero), the same numbers in both precisions
CALL RANDOM_NUMBER(ZSPD)
ZSPD = ZSPD - 0.5_C_DOUBLE
DO JM = 0, NSMAX
  DO I = 0, NSMAX-JM
    ZSPD(:,NASM0(JM)+2*I:NASM0(JM)+2*I+1) = ZSPD(:,NASM0(JM)+2*I:NASM0(JM)+2*I+1)/REAL(JM+I+1,C_DOUBLE)
  ENDDO
ENDDO
DO I = 0, NSMAX
  ZSPD(:,NASM0(0)+2*I+1) = 0
ENDDO
ZSPS = REAL(ZSPD, C_FLOAT)
ZSPD = REAL(ZSPS, C_DOUBLE)              ! both libraries start from float-representable coefficients
CALL INV_TRANS_DP(PSPSCALAR=ZSPD, KRESOL=IRD, PGP=ZGPD)
CALL INV_TRANS_SP(PSPSCALAR=ZSPS, KRESOL=IRS, PGP=ZGPS)
ZERR = MAXVAL(ABS(REAL(ZGPS,C_DOUBLE)-ZGPD))/MAXVAL(ABS(ZGPD))
WRITE(*,'(A,ES10.2)') 'sp against dp grid fields (of the field maximum) ', ZERR
IF (ZERR > 2E-5_C_DOUBLE .OR. ZERR < 1E-9_C_DOUBLE) ERROR STOP 5    ! float accuracy -- and NOT double: the sp library really ran fp32 kernels
CALL DIR_TRANS_DP(PSPSCALAR=ZSPD2, KRESOL=IRD, PGP=ZGPD)
CALL DIR_TRANS_SP(PSPSCALAR=ZSPS2, KRESOL=IRS, PGP=ZGPS)
ZERR = MAXVAL(ABS(ZSPD2-ZSPD))
WRITE(*,'(A,ES10.2)') 'dp round trip ', ZERR
IF (ZERR > 1E-11_C_DOUBLE) ERROR STOP 6   ! an octahedral grid drops the (lat, m > NMEN(lat)) corner: ~1e-12 inherent
ZERR = MAXVAL(ABS(REAL(ZSPS2,C_DOUBLE)-ZSPD))
WRITE(*,'(A,ES10.2)') 'sp round trip ', ZERR
IF (ZERR > 2E-5_C_DOUBLE) ERROR STOP 7
CALL SPECNORM_DP(PNORM=ZND, PSPEC=ZSPD, KRESOL=IRD)
CALL SPECNORM_SP(PNORM=ZNS, PSPEC=ZSPS, KRESOL=IRS)
IF (MAXVAL(ABS(REAL(ZNS,C_DOUBLE)/ZND-1)) > 1E-5_C_DOUBLE) ERROR STOP 8
! releasing one precision's resolution leaves the other's usable
CALL TRANS_RELEASE_SP(IRS)
CALL INV_TRANS_DP(PSPSCALAR=ZSPD, KRESOL=IRD, PGP=ZGPD)
I419 = NASM0(4)+2*(19-4)
IF (I419 < 1) ERROR STOP 9
CALL TRANS_END_DP()
WRITE(*,'(A)') 'FORTRAN SHIM OK (dp and sp in one executable)'
END PROGRAM TEST_SHIM_BOTH
