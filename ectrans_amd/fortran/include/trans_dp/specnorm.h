! Automatically generated interface header for backward compatibility of generic symbols !
#if defined(specnorm)
#undef specnorm
#endif
#if defined(SPECNORM)
#undef SPECNORM
#endif
#include "../specnorm_dp.h"
#define specnorm SPECNORM_DP
#define SPECNORM SPECNORM_DP
