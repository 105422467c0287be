! Automatically generated interface header for backward compatibility of generic symbols !
#if defined(gpnorm_trans)
#undef gpnorm_trans
#endif
#if defined(GPNORM_TRANS)
#undef GPNORM_TRANS
#endif
#include "../gpnorm_trans_dp.h"
#define gpnorm_trans GPNORM_TRANS_DP
#define GPNORM_TRANS GPNORM_TRANS_DP
