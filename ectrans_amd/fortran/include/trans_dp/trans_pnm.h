! Automatically generated interface header for backward compatibility of generic symbols !
#if defined(trans_pnm)
#undef trans_pnm
#endif
#if defined(TRANS_PNM)
#undef TRANS_PNM
#endif
#include "../trans_pnm_dp.h"
#define trans_pnm TRANS_PNM_DP
#define TRANS_PNM TRANS_PNM_DP
