! Automatically generated interface header for backward compatibility of generic symbols !
#if defined(trans_release)
#undef trans_release
#endif
#if defined(TRANS_RELEASE)
#undef TRANS_RELEASE
#endif
#include "../trans_release_sp.h"
#define trans_release TRANS_RELEASE_SP
#define TRANS_RELEASE TRANS_RELEASE_SP
