! Automatically generated interface header for backward compatibility of generic symbols !
#if defined(specnorm)
#undef specnorm
#endif
#if defined(SPECNORM)
#undef SPECNORM
#endif
#include "../specnorm_sp.h"
#define specnorm SPECNORM_SP
#define SPECNORM SPECNORM_SP
