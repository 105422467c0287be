! Automatically generated interface header for backward compatibility of generic symbols !
#if defined(gath_spec)
#undef gath_spec
#endif
#if defined(GATH_SPEC)
#undef GATH_SPEC
#endif
#include "../gath_spec_sp.h"
#define gath_spec GATH_SPEC_SP
#define GATH_SPEC GATH_SPEC_SP
