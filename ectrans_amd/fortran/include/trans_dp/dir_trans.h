! Automatically generated interface header for backward compatibility of generic symbols !
#if defined(dir_trans)
#undef dir_trans
#endif
#if defined(DIR_TRANS)
#undef DIR_TRANS
#endif
#include "../dir_trans_dp.h"
#define dir_trans DIR_TRANS_DP
#define DIR_TRANS DIR_TRANS_DP
