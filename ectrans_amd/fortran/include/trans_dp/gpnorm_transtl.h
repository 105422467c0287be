! Automatically generated interface header for backward compatibility of generic symbols !
#if defined(gpnorm_transtl)
#undef gpnorm_transtl
#endif
#if defined(GPNORM_TRANSTL)
#undef GPNORM_TRANSTL
#endif
#include "../gpnorm_transtl_dp.h"
#define gpnorm_transtl GPNORM_TRANSTL_DP
#define GPNORM_TRANSTL GPNORM_TRANSTL_DP
