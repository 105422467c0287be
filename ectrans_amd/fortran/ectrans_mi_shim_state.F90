! ectrans_mi_shim_state.F90 -- the one piece of Fortran-side state of the drop-in, compiled ONCE (libectrans_mi_f_common.so) and shared by
! the dp and the sp library: the default resolution (the reference keeps NDEF_RESOL / NCUR_RESOL in TPM_GEN of ectrans_common,
! common/internal/tpm_gen.F90; resolution handles are numbered across both precisions there too, and here -- libectrans_mi.so holds them).
MODULE EMI_SHIM_STATE_MOD
USE, INTRINSIC :: ISO_C_BINDING
IMPLICIT NONE
INTEGER(C_INT32_T), SAVE :: NDEF_RESOL = 1   ! default resolution = first defined (set_resol_mod.F90)
END MODULE EMI_SHIM_STATE_MOD
