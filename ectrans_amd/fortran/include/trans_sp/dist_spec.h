! Automatically generated interface header for backward compatibility of generic symbols !
#if defined(dist_spec)
#undef dist_spec
#endif
#if defined(DIST_SPEC)
#undef DIST_SPEC
#endif
#include "../dist_spec_sp.h"
#define dist_spec DIST_SPEC_SP
#define DIST_SPEC DIST_SPEC_SP
