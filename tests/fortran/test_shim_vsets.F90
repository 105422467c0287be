#include "emi_variant.h"   /* the generic names of this file are the _DP (or, with -DEMI_SP, _SP) entry points: ectrans_amd/fortran/emi_variant.h */
! NPRTRV = 2 through the Fortran drop-in: `mpiexec -n 2` (NPRTRW x NPRTRV = 1 x 2) and `-n 4` (2 x 2).
! As in the IFS, the spectral arrays of a task hold the wavenumbers of its W-set and the fields of its V-set (KVSETSC), the
! grid arrays all fields on its own latitudes (inv_trans.F90:212-300).  Two checks without an oracle:
!  (A) placement: field f = f * P_1^0 = f sqrt(3) mu: after INV_TRANS every task must find f sqrt(3) mu(lat) in slot f of PGP on
!      all its points -- whichever V-set transformed the field;
!  (B) dense fields: INV_TRANS -> DIR_TRANS returns the local coefficients, SPECNORM(KVSET) the same norms before and after.
PROGRAM TEST_SHIM_VSETS
USE, INTRINSIC :: ISO_C_BINDING
USE EMI_SHIM_MOD, ONLY : JPIM, JPRB, JPRD
USE ECTRANS_MI_INTERFACES
IMPLICIT NONE
INTERFACE
  INTEGER(C_INT) FUNCTION EMI_TEST_MPI_BEGIN_V(NPRTRV, NPROC, MYPROC) BIND(C, NAME="emi_test_mpi_begin_v")
    IMPORT
    INTEGER(C_INT), VALUE :: NPRTRV
    INTEGER(C_INT), INTENT(OUT) :: NPROC, MYPROC
  END FUNCTION
  SUBROUTINE EMI_TEST_MPI_END() BIND(C, NAME="emi_test_mpi_end")
  END SUBROUTINE
END INTERFACE
INTEGER(JPIM), PARAMETER :: NSMAX = 47, NDGL = 96, NFG = 5, NPRV = 2
INTEGER(C_INT) :: INPROC, IMYPROC
INTEGER(JPIM) :: NLOEN(NDGL), I, J, JM, JN, JF, JL, ISP, NSPEC2, NGPTOT, NPROMA, NGPBLKS, IRESOL, IPRW, IMYW, IMYV, NUMP, NFL, IP
INTEGER(JPIM) :: IVSET(NFG), ILOC(NFG), IASM0(0:NSMAX)
INTEGER(JPIM), ALLOCATABLE :: IMYMS(:), IFRST(:), ILST(:)
REAL(JPRB), ALLOCATABLE :: ZSP(:,:), ZSP2(:,:), ZGP(:,:,:)
REAL(JPRD) :: ZMU(NDGL)
REAL(JPRB) :: ZNORM0(NFG), ZNORM1(NFG)
REAL(JPRD) :: ZERR, ZTOL, ZWANT

IF (EMI_TEST_MPI_BEGIN_V(NPRV, INPROC, IMYPROC) /= 0) ERROR STOP 'transport attach failed'
IF (MOD(INPROC, NPRV) /= 0) ERROR STOP 'run with an even number of tasks'
DO I = 1, NDGL/2
  NLOEN(I) = 20 + 4*(I-1)
  NLOEN(NDGL+1-I) = NLOEN(I)
ENDDO
CALL SETUP_TRANS0(KPRINTLEV=0, KMAX_RESOL=2, KPRGPNS=INPROC, KPRGPEW=1, KPRTRW=INPROC/NPRV)
CALL SETUP_TRANS(KSMAX=NSMAX, KDGL=NDGL, KLOEN=NLOEN, LDSPLIT=.FALSE., KRESOL=IRESOL)
ALLOCATE(IFRST(INPROC), ILST(INPROC))
CALL TRANS_INQ(KRESOL=IRESOL, KSPEC2=NSPEC2, KGPTOT=NGPTOT, KNUMP=NUMP, KPRTRW=IPRW, KMYSETW=IMYW, KMYSETV=IMYV, PMU=ZMU, &
 &             KFRSTLAT=IFRST, KLSTLAT=ILST, KASM0=IASM0)
IF (IPRW /= INPROC/NPRV .OR. IMYW /= (IMYPROC-1)/NPRV+1 .OR. IMYV /= MOD(IMYPROC-1,NPRV)+1) ERROR STOP 'PE2SET numbering'
ALLOCATE(IMYMS(NUMP))
CALL TRANS_INQ(KRESOL=IRESOL, KMYMS=IMYMS)
DO JF = 1, NFG
  IVSET(JF) = MOD(JF-1, NPRV) + 1      ! fields dealt round-robin to the V-sets
ENDDO
NFL = COUNT(IVSET == IMYV)
IP = 0
DO JF = 1, NFG
  IF (IVSET(JF) == IMYV) THEN
    IP = IP + 1
    ILOC(IP) = JF                      ! global number of local field IP
  ENDIF
ENDDO
NPROMA = 300
NGPBLKS = (NGPTOT-1)/NPROMA+1
ALLOCATE(ZSP(NFL, NSPEC2), ZSP2(NFL, NSPEC2), ZGP(NPROMA, NFG, NGPBLKS))
ZTOL = 1.0D-9
IF (JPRB /= JPRD) ZTOL = 1.0D-4
! ---- (A) field f = f P_1^0
ZSP = 0
IF (IASM0(0) > 0) THEN                 ! this W-set owns m = 0: (m, n) = (0, 1) sits at NASM0(0) + 2
  DO IP = 1, NFL
    ZSP(IP, IASM0(0)+2) = REAL(ILOC(IP), JPRB)
  ENDDO
ENDIF
ZGP = -999
CALL INV_TRANS(PSPSCALAR=ZSP, KVSETSC=IVSET, KPROMA=NPROMA, KRESOL=IRESOL, PGP=ZGP)
ZERR = 0
IP = 0
DO JL = IFRST(IMYPROC), ILST(IMYPROC)
  DO J = 1, NLOEN(JL)
    IP = IP + 1
    DO JF = 1, NFG
      ZWANT = JF*SQRT(3.0D0)*ZMU(JL)
      ZERR = MAX(ZERR, ABS(REAL(ZGP(MOD(IP-1,NPROMA)+1, JF, (IP-1)/NPROMA+1), JPRD) - ZWANT))
    ENDDO
  ENDDO
ENDDO
IF (IP /= NGPTOT) ERROR STOP 'KFRSTLAT / KLSTLAT do not describe NGPTOT'
WRITE(*,'(A,I0,A,I0,A,I0,A,ES10.2)') 'task ', IMYPROC, ' (W-set ', IMYW, ', V-set ', IMYV, '): every field on its own latitudes, max err ', ZERR
IF (.NOT. (ZERR < ZTOL*NFG*2)) ERROR STOP 'a field is not where INV_TRANS should have put it'
! ---- (B) dense local coefficients: a function of the GLOBAL (m, n, field)
DO IP = 1, NFL
  JF = ILOC(IP)
  DO I = 1, NUMP
    JM = IMYMS(I)
    DO JN = JM, NSMAX
      ISP = IASM0(JM) + 2*(JN-JM)
      ZSP(IP, ISP)   = REAL(COS(0.37D0*JM + 1.1D0*JN + JF) / (JN+1.0D0), JPRB)
      ZSP(IP, ISP+1) = REAL(SIN(0.71D0*JM - 0.3D0*JN + JF) / (JN+1.0D0), JPRB)
      IF (JM == 0) ZSP(IP, ISP+1) = 0
    ENDDO
  ENDDO
ENDDO
CALL SPECNORM(PNORM=ZNORM0, PSPEC=ZSP, KVSET=IVSET, KRESOL=IRESOL)
CALL INV_TRANS(PSPSCALAR=ZSP, KVSETSC=IVSET, KPROMA=NPROMA, KRESOL=IRESOL, PGP=ZGP)
ZSP2 = 0
CALL DIR_TRANS(PSPSCALAR=ZSP2, KVSETSC=IVSET, KPROMA=NPROMA, KRESOL=IRESOL, PGP=ZGP)
CALL SPECNORM(PNORM=ZNORM1, PSPEC=ZSP2, KVSET=IVSET, KRESOL=IRESOL)
ZERR = 0
IF (NFL > 0) ZERR = MAXVAL(ABS(REAL(ZSP2, JPRD) - REAL(ZSP, JPRD)))
WRITE(*,'(A,I0,A,ES10.2,A,5ES10.2)') 'task ', IMYPROC, ': round trip max err ', ZERR, ' norm drift ', ABS(ZNORM0/ZNORM1 - 1.0_JPRB)
IF (.NOT. (ZERR < ZTOL)) ERROR STOP 'round trip error too large'
IF (ANY(ABS(REAL(ZNORM0, JPRD)/REAL(ZNORM1, JPRD) - 1.0D0) > ZTOL)) ERROR STOP 'spectral norm drift too large'
CALL TRANS_END()
WRITE(*,'(A,I0,A,I0)') 'FORTRAN SHIM VSETS OK task ', IMYPROC, ' of ', INPROC
CALL EMI_TEST_MPI_END()
END PROGRAM TEST_SHIM_VSETS
