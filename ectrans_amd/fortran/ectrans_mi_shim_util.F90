! ectrans_mi_shim_util.F90 -- the utility routines of the reference's public interface (round 4), so that every header of
! /root/reference/src/trans/include/ectrans/ that belongs to the spherical-harmonic package has a counterpart in the drop-in:
!   GPNORM_TRANS  (gpnorm_trans.h)   grid-point norms            -> emi_gpnorm       (k_gpnorm)
!   VORDIV_TO_UV  (vordiv_to_uv.h)   spectral vor / div -> U, V  -> emi_vordiv_to_uv (k_vd2uv)
!   TRANS_PNM     (trans_pnm.h)      Legendre polynomials of one zonal wavenumber -> emi_inq_legendre
!   GET_CURRENT   (get_current.h)    the default resolution handle
!   INI_SPEC_DIST (ini_spec_dist.h)  spectral distribution of a truncation over KPRTRW W-sets (integer bookkeeping only)
! As in ectrans_mi_shim.F90 the routines only marshal arguments; arithmetic happens in the HIP library.

! ---------------------------------------------------------------------------------------------
! GPNORM_TRANS (cpu/external/gpnorm_trans.F90:11-96): PAVE = area-weighted average, PMIN / PMAX of the first KFIELDS fields of
! PGP(NPROMA, fields, NGPBLKS).  The results are valid on every task (the reference: task 1).
! ---------------------------------------------------------------------------------------------
SUBROUTINE GPNORM_TRANS(PGP,KFIELDS,KPROMA,PAVE,PMIN,PMAX,LDAVE_ONLY,KRESOL)
USE EMI_SHIM_MOD
IMPLICIT NONE
REAL(KIND=JPRB)   ,INTENT(IN)    :: PGP(:,:,:)
REAL(KIND=JPRB)   ,INTENT(OUT)   :: PAVE(:)
REAL(KIND=JPRB)   ,INTENT(INOUT) :: PMIN(:)
REAL(KIND=JPRB)   ,INTENT(INOUT) :: PMAX(:)
INTEGER(KIND=JPIM),INTENT(IN)    :: KFIELDS
INTEGER(KIND=JPIM),INTENT(IN)    :: KPROMA
LOGICAL           ,INTENT(IN)    :: LDAVE_ONLY
INTEGER(KIND=JPIM),OPTIONAL, INTENT(IN)  :: KRESOL
INTEGER(C_INT) :: IRESOL, INGPTOT
INTEGER(JPIM) :: IGPBLKS
REAL(KIND=JPRB), ALLOCATABLE, TARGET :: ZGP(:,:,:)
REAL(C_DOUBLE), ALLOCATABLE :: ZAVE(:), ZMIN(:), ZMAX(:)
TYPE(C_PTR) :: P
IRESOL = NDEF_RESOL
IF (PRESENT(KRESOL)) IRESOL = KRESOL
CALL CHK(EMI_INQ_INT(IRESOL, 'ngptot'//C_NULL_CHAR, INGPTOT), 'GPNORM_TRANS')
IGPBLKS = (INGPTOT-1)/KPROMA+1
! the reference's extent checks (gpnorm_trans_ctl_mod.F90:103-114)
IF (SIZE(PGP,1) < KPROMA)  CALL SHIM_ABORT('GPNORM_TRANS_CTL:FIRST DIMENSION OF PGP TOO SMALL ')
IF (SIZE(PGP,2) < KFIELDS) CALL SHIM_ABORT('GPNORM_TRANS_CTL:SECOND DIMENSION OF PGP TOO SMALL ')
IF (SIZE(PGP,3) < IGPBLKS) CALL SHIM_ABORT('GPNORM_TRANS_CTL:THIRD DIMENSION OF PGP TOO SMALL ')
IF (KFIELDS <= 0) RETURN
IF (IS_CONTIGUOUS(PGP) .AND. SIZE(PGP,1) == KPROMA) THEN
  P = LOC_R(PGP)
  ALLOCATE(ZAVE(KFIELDS), ZMIN(KFIELDS), ZMAX(KFIELDS))
  IF (LDAVE_ONLY) THEN
    ZMIN = REAL(PMIN(1:KFIELDS), C_DOUBLE); ZMAX = REAL(PMAX(1:KFIELDS), C_DOUBLE)
  ENDIF
  CALL CHK(EMI_GPNORM(IRESOL, 0_C_INT, P, INT(SIZE(PGP,2),C_INT), INT(KFIELDS,C_INT), INT(KPROMA,C_INT), ZAVE, ZMIN, ZMAX, &
   & MERGE(1_C_INT,0_C_INT,LDAVE_ONLY)), 'GPNORM_TRANS')
ELSE   ! a packed copy with leading extent NPROMA
  ALLOCATE(ZGP(KPROMA,KFIELDS,IGPBLKS))
  ZGP = PGP(1:KPROMA,1:KFIELDS,1:IGPBLKS)
  ALLOCATE(ZAVE(KFIELDS), ZMIN(KFIELDS), ZMAX(KFIELDS))
  IF (LDAVE_ONLY) THEN
    ZMIN = REAL(PMIN(1:KFIELDS), C_DOUBLE); ZMAX = REAL(PMAX(1:KFIELDS), C_DOUBLE)
  ENDIF
  CALL CHK(EMI_GPNORM(IRESOL, 0_C_INT, C_LOC(ZGP), INT(KFIELDS,C_INT), INT(KFIELDS,C_INT), INT(KPROMA,C_INT), ZAVE, ZMIN, ZMAX, &
   & MERGE(1_C_INT,0_C_INT,LDAVE_ONLY)), 'GPNORM_TRANS')
ENDIF
PAVE(1:KFIELDS) = REAL(ZAVE, JPRB)
PMIN(1:KFIELDS) = REAL(ZMIN, JPRB)
PMAX(1:KFIELDS) = REAL(ZMAX, JPRB)
END SUBROUTINE GPNORM_TRANS

! ---------------------------------------------------------------------------------------------
! VORDIV_TO_UV (cpu/external/vordiv_to_uv.F90:11-178): PSPU / PSPV = U, V = (u, v) cos(theta) in spectral space from vorticity /
! divergence, truncation KSMAX.  SETUP_TRANS0 is called on the caller's behalf if it has not been (as the reference does).
! ---------------------------------------------------------------------------------------------
SUBROUTINE VORDIV_TO_UV(PSPVOR,PSPDIV,PSPU,PSPV,KSMAX,KVSETUV)
USE EMI_SHIM_MOD
USE ECTRANS_MI_INTERFACES, ONLY : SETUP_TRANS0, TRANS_END
IMPLICIT NONE
REAL(KIND=JPRB), INTENT(IN) :: PSPVOR(:,:)
REAL(KIND=JPRB), INTENT(IN) :: PSPDIV(:,:)
REAL(KIND=JPRB), INTENT(OUT) :: PSPU(:,:)
REAL(KIND=JPRB), INTENT(OUT) :: PSPV(:,:)
INTEGER(KIND=JPIM) , INTENT(IN) :: KSMAX
INTEGER(KIND=JPIM) ,OPTIONAL, INTENT(IN) :: KVSETUV(:)
INTEGER(C_INT) :: IW, IV, IMW, IMV, INPROC, IMYPROC
INTEGER(JPIM) :: IF_UV, J, INS2
LOGICAL :: LLTMP0
REAL(KIND=JPRB), ALLOCATABLE, TARGET :: ZVOR(:,:), ZDIV(:,:), ZU(:,:), ZV(:,:)
LLTMP0 = EMI_INQ_TASKS(INPROC, IMYPROC) /= 0    ! MSETUP0 == 0 (vordiv_to_uv.F90:103-109)
IF (LLTMP0) CALL SETUP_TRANS0()
CALL CHK(EMI_INQ_VSETS(IW, IV, IMW, IMV), 'VORDIV_TO_UV')
IF (PRESENT(KVSETUV)) THEN
  IF_UV = 0
  DO J=1,SIZE(KVSETUV)
    IF (KVSETUV(J) > IV .OR. KVSETUV(J) < 1) CALL SHIM_ABORT('VORDIV_TO_UV:KVSETUV TOO LONG OR CONTAINS VALUES OUTSIDE RANGE')
    IF (KVSETUV(J) == IMV) IF_UV = IF_UV+1
  ENDDO
ELSE
  IF_UV = SIZE(PSPVOR,1)
  IF (IV > 1 .AND. IF_UV > 0) CALL SHIM_ABORT('VORDIV_TO_UV: SPECIFY VERTICAL SPECTRAL DISTRIBUTION!')
ENDIF
IF (IF_UV > 0) THEN
  IF (SIZE(PSPVOR,1) < IF_UV) CALL SHIM_ABORT('VORDIV_TO_UV : PSPVOR TOO SHORT')
  IF (SIZE(PSPDIV,1) < IF_UV) CALL SHIM_ABORT('VORDIV_TO_UV : PSPDIV TOO SHORT')
  IF (SIZE(PSPU,1) < IF_UV)   CALL SHIM_ABORT('VORDIV_TO_UV : PSPU TOO SHORT')
  IF (SIZE(PSPV,1) < IF_UV)   CALL SHIM_ABORT('VORDIV_TO_UV : PSPV TOO SHORT')
  INS2 = SIZE(PSPVOR,2)
  IF (IS_CONTIGUOUS(PSPVOR) .AND. IS_CONTIGUOUS(PSPDIV) .AND. IS_CONTIGUOUS(PSPU) .AND. IS_CONTIGUOUS(PSPV) .AND. &
   &  SIZE(PSPVOR,1) == IF_UV .AND. SIZE(PSPDIV,1) == IF_UV .AND. SIZE(PSPU,1) == IF_UV .AND. SIZE(PSPV,1) == IF_UV) THEN
    CALL CHK(EMI_VORDIV_TO_UV(INT(KSMAX,C_INT), INT(STORAGE_SIZE(1.0_JPRB)/8,C_INT), 0_C_INT, LOC_R(PSPVOR), LOC_R(PSPDIV), &
     & LOC_R(PSPU), LOC_R(PSPV), INT(IF_UV,C_INT)), 'VORDIV_TO_UV')
  ELSE
    ALLOCATE(ZVOR(IF_UV,INS2), ZDIV(IF_UV,INS2), ZU(IF_UV,INS2), ZV(IF_UV,INS2))
    ZVOR = PSPVOR(1:IF_UV,:); ZDIV = PSPDIV(1:IF_UV,:)
    CALL CHK(EMI_VORDIV_TO_UV(INT(KSMAX,C_INT), INT(STORAGE_SIZE(1.0_JPRB)/8,C_INT), 0_C_INT, C_LOC(ZVOR), C_LOC(ZDIV), &
     & C_LOC(ZU), C_LOC(ZV), INT(IF_UV,C_INT)), 'VORDIV_TO_UV')
    PSPU(1:IF_UV,1:INS2) = ZU; PSPV(1:IF_UV,1:INS2) = ZV
  ENDIF
ENDIF
IF (LLTMP0) CALL TRANS_END()
END SUBROUTINE VORDIV_TO_UV

! ---------------------------------------------------------------------------------------------
! TRANS_PNM (cpu/external/trans_pnm.F90:11-198): the Legendre polynomials of zonal wavenumber KM in the layout of the reference:
! PRPNM(latitude, column) (LDTRANSPOSE: (column, latitude)), latitudes ISL .. NDGNH of the northern hemisphere, antisymmetric
! (odd n - m) and symmetric columns interleaved, n descending (IA, IA+2, ... and IS, IS+2, ...).  The values are the panels the
! transforms use (SUPOLF at set-up; LDCHEAP selects a cheaper recurrence in the reference, the same values to rounding).
! ---------------------------------------------------------------------------------------------
SUBROUTINE TRANS_PNM(KRESOL,KM,PRPNM,LDTRANSPOSE,LDCHEAP)
USE EMI_SHIM_MOD
IMPLICIT NONE
INTEGER(KIND=JPIM) ,OPTIONAL, INTENT(IN)  :: KRESOL
INTEGER(KIND=JPIM) ,INTENT(IN)  :: KM
REAL(KIND=JPRB)    ,OPTIONAL, INTENT(OUT) :: PRPNM(:,:)
LOGICAL, OPTIONAL, INTENT(IN) :: LDTRANSPOSE
LOGICAL, OPTIONAL, INTENT(IN) :: LDCHEAP
INTEGER(C_INT) :: IRESOL, INSMAX, INDGL, INR, INC
INTEGER(JPIM) :: ILA, ILS, IA, IS, ISL, INDGNH, INLEI3, JGL, JI, JS
LOGICAL :: LLT
REAL(C_DOUBLE), ALLOCATABLE, TARGET :: ZPAN(:,:)
IF (.NOT.PRESENT(PRPNM)) RETURN
IRESOL = NDEF_RESOL
IF (PRESENT(KRESOL)) IRESOL = KRESOL
LLT = .FALSE.
IF (PRESENT(LDTRANSPOSE)) LLT = LDTRANSPOSE
CALL CHK(EMI_INQ_INT(IRESOL, 'nsmax'//C_NULL_CHAR, INSMAX), 'TRANS_PNM')
CALL CHK(EMI_INQ_INT(IRESOL, 'ndgl'//C_NULL_CHAR, INDGL), 'TRANS_PNM')
INDGNH = INDGL/2
INLEI3 = INDGNH+MOD(INDGNH+2,2)                         ! R%NLEI3 (setup_dims_mod.F90)
IF (LLT) THEN
  IF (SIZE(PRPNM,2) < INLEI3)       CALL SHIM_ABORT('TRANS_PNM : FIRST DIM. OF PRPNM TOO SMALL')
  IF (SIZE(PRPNM,1) < INSMAX-KM+3)  CALL SHIM_ABORT('TRANS_PNM : SECOND DIM. OF PRPNM TOO SMALL')
  PRPNM(:,INLEI3) = 0.0_JPRB
ELSE
  IF (SIZE(PRPNM,1) < INLEI3)       CALL SHIM_ABORT('TRANS_PNM : FIRST DIM. OF PRPNM TOO SMALL')
  IF (SIZE(PRPNM,2) < INSMAX-KM+3)  CALL SHIM_ABORT('TRANS_PNM : SECOND DIM. OF PRPNM TOO SMALL')
  PRPNM(INLEI3,:) = 0.0_JPRB
ENDIF
ILA = (INSMAX-KM+2)/2
ILS = (INSMAX-KM+3)/2
IA  = 1+MOD(INSMAX-KM+2,2)
IS  = 1+MOD(INSMAX-KM+1,2)
DO JS = 0, 1          ! antisymmetric, symmetric panel: (rows = latitudes ISL .. NDGNH, columns = n descending)
  CALL CHK(EMI_INQ_LEGENDRE(IRESOL, INT(KM,C_INT), INT(JS,C_INT), C_NULL_PTR, INR, INC), 'TRANS_PNM')
  IF (INR*INC == 0) CYCLE
  ALLOCATE(ZPAN(INR,INC))
  CALL CHK(EMI_INQ_LEGENDRE(IRESOL, INT(KM,C_INT), INT(JS,C_INT), C_LOC(ZPAN), INR, INC), 'TRANS_PNM')
  ISL = MAX(INDGNH-INR+1,1)
  DO JGL = 1, INR
    DO JI = 1, MERGE(ILS, ILA, JS == 1)
      IF (JI > INC) CYCLE
      IF (LLT) THEN
        PRPNM(MERGE(IS,IA,JS==1)+(JI-1)*2, ISL+JGL-1) = REAL(ZPAN(JGL,JI), JPRB)
      ELSE
        PRPNM(ISL+JGL-1, MERGE(IS,IA,JS==1)+(JI-1)*2) = REAL(ZPAN(JGL,JI), JPRB)
      ENDIF
    ENDDO
  ENDDO
  DEALLOCATE(ZPAN)
ENDDO
END SUBROUTINE TRANS_PNM

! ---------------------------------------------------------------------------------------------
! GET_CURRENT (common/external/get_current.F90:11-66): the current (default) resolution handle; LDLAM is always false here
! (the limited-area package is out of scope).
! ---------------------------------------------------------------------------------------------
SUBROUTINE GET_CURRENT(KRESOL,LDLAM)
USE EMI_SHIM_MOD
IMPLICIT NONE
INTEGER(KIND=JPIM)  ,OPTIONAL,INTENT(OUT)  :: KRESOL
LOGICAL             ,OPTIONAL,INTENT(OUT)  :: LDLAM
IF (PRESENT(KRESOL)) KRESOL = NDEF_RESOL
IF (PRESENT(LDLAM))  LDLAM = .FALSE.
END SUBROUTINE GET_CURRENT

! ---------------------------------------------------------------------------------------------
! INI_SPEC_DIST (common/external/ini_spec_dist.F90:11-98 = SUWAVEDI, common/internal/suwavedi_mod.F90:13-183): how a triangular
! truncation KSMAX is dealt to KPRTRW W-sets -- integer bookkeeping, no state.  Wavenumbers go to the sets 1, 2, .., KPRTRW,
! KPRTRW, .., 2, 1, 1, 2, ... (the zig-zag that balances the Legendre work); everything else is counting:
!   KPROCM(m) set of wavenumber m | KUMPP(a) wavenumbers of set a | KMYMS the wavenumbers of set KMYSETW, ascending
!   KASM0(m) position of (m, n = m) in that set's spectral arrays (-99: not mine) | KSPEC / KSPEC2 complex / real coefficients of the set
!   KSPEC2MX the largest KSPEC2 | KSPOLEGL sum of (KTMAX + 2 - m) over the set | KPOSSP(a) start of set a in a gathered array
!   KPTRMS(a) start of set a in KALLMS | KALLMS all wavenumbers, set by set
! ---------------------------------------------------------------------------------------------
SUBROUTINE INI_SPEC_DIST(KSMAX,KTMAX,KPRTRW,KMYSETW,KASM0,KSPOLEGL,KPROCM,&
                    &KUMPP,KSPEC,KSPEC2,KSPEC2MX,KPOSSP,KMYMS,KPTRMS,KALLMS)
USE EMI_SHIM_MOD
IMPLICIT NONE
INTEGER(KIND=JPIM),INTENT(IN)  :: KSMAX
INTEGER(KIND=JPIM),INTENT(IN)  :: KTMAX
INTEGER(KIND=JPIM),INTENT(IN)  :: KPRTRW
INTEGER(KIND=JPIM),INTENT(IN)  :: KMYSETW
INTEGER(KIND=JPIM),OPTIONAL,INTENT(OUT) :: KSPEC
INTEGER(KIND=JPIM),OPTIONAL,INTENT(OUT) :: KSPEC2
INTEGER(KIND=JPIM),OPTIONAL,INTENT(OUT) :: KSPEC2MX
INTEGER(KIND=JPIM),OPTIONAL,INTENT(OUT) :: KSPOLEGL
INTEGER(KIND=JPIM),OPTIONAL,INTENT(OUT) :: KASM0(0:KSMAX)
INTEGER(KIND=JPIM),OPTIONAL,INTENT(OUT) :: KPROCM(0:KSMAX)
INTEGER(KIND=JPIM),OPTIONAL,INTENT(OUT) :: KUMPP(KPRTRW)
INTEGER(KIND=JPIM),OPTIONAL,INTENT(OUT) :: KPOSSP(KPRTRW+1)
INTEGER(KIND=JPIM),OPTIONAL,INTENT(OUT) :: KMYMS(KSMAX+1)
INTEGER(KIND=JPIM),OPTIONAL,INTENT(OUT) :: KPTRMS(KPRTRW)
INTEGER(KIND=JPIM),OPTIONAL,INTENT(OUT) :: KALLMS(KSMAX+1)
INTEGER(JPIM) :: IOWNER(0:KSMAX), ICOEF(KPRTRW), INUM(KPRTRW), IFILL(KPRTRW), ISTART(KPRTRW)
INTEGER(JPIM) :: JM, JA, ISET, ISTEP, IMINE, IPOS
! owner of every wavenumber: a pointer that bounces between set 1 and set KPRTRW
ISET = 0; ISTEP = 1
DO JM = 0, KSMAX
  ISET = ISET+ISTEP
  IF (ISET > KPRTRW) THEN
    ISET = KPRTRW; ISTEP = -1
  ELSEIF (ISET < 1) THEN
    ISET = 1; ISTEP = 1
  ENDIF
  IOWNER(JM) = ISET
ENDDO
ICOEF(:) = 0; INUM(:) = 0
DO JM = 0, KSMAX
  ICOEF(IOWNER(JM)) = ICOEF(IOWNER(JM))+KSMAX-JM+1
  INUM(IOWNER(JM))  = INUM(IOWNER(JM))+1
ENDDO
ISTART(1) = 1
DO JA = 2, KPRTRW
  ISTART(JA) = ISTART(JA-1)+INUM(JA-1)
ENDDO
IF (PRESENT(KPROCM)) KPROCM(:) = IOWNER(:)
IF (PRESENT(KUMPP))  KUMPP(:)  = INUM(:)
IF (PRESENT(KPTRMS)) KPTRMS(:) = ISTART(:)
IF (PRESENT(KSPEC))    KSPEC    = ICOEF(KMYSETW)
IF (PRESENT(KSPEC2))   KSPEC2   = 2*ICOEF(KMYSETW)
IF (PRESENT(KSPEC2MX)) KSPEC2MX = 2*MAXVAL(ICOEF)
IF (PRESENT(KPOSSP)) THEN
  KPOSSP(1) = 1
  DO JA = 1, KPRTRW
    KPOSSP(JA+1) = KPOSSP(JA)+2*ICOEF(JA)
  ENDDO
ENDIF
IF (PRESENT(KALLMS)) THEN
  IFILL(:) = 0
  DO JM = 0, KSMAX
    KALLMS(ISTART(IOWNER(JM))+IFILL(IOWNER(JM))) = JM
    IFILL(IOWNER(JM)) = IFILL(IOWNER(JM))+1
  ENDDO
ENDIF
! this set's own wavenumbers
IF (PRESENT(KASM0)) KASM0(:) = -99
IMINE = 0; IPOS = 1
IF (PRESENT(KSPOLEGL)) KSPOLEGL = 0
DO JM = 0, KSMAX
  IF (IOWNER(JM) /= KMYSETW) CYCLE
  IMINE = IMINE+1
  IF (PRESENT(KMYMS)) KMYMS(IMINE) = JM
  IF (PRESENT(KASM0)) KASM0(JM) = IPOS
  IPOS = IPOS+2*(KSMAX-JM+1)
  IF (PRESENT(KSPOLEGL)) KSPOLEGL = KSPOLEGL+KTMAX+2-JM
ENDDO
END SUBROUTINE INI_SPEC_DIST
