! Automatically generated interface header for backward compatibility of generic symbols !
#if defined(inv_trans)
#undef inv_trans
#endif
#if defined(INV_TRANS)
#undef INV_TRANS
#endif
#include "../inv_trans_sp.h"
#define inv_trans INV_TRANS_SP
#define INV_TRANS INV_TRANS_SP
