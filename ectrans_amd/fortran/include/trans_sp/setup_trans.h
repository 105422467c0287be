! Automatically generated interface header for backward compatibility of generic symbols !
#if defined(setup_trans)
#undef setup_trans
#endif
#if defined(SETUP_TRANS)
#undef SETUP_TRANS
#endif
#include "../setup_trans_sp.h"
#define setup_trans SETUP_TRANS_SP
#define SETUP_TRANS SETUP_TRANS_SP
