! Automatically generated interface header for backward compatibility of generic symbols !
#if defined(dir_transad)
#undef dir_transad
#endif
#if defined(DIR_TRANSAD)
#undef DIR_TRANSAD
#endif
#include "../dir_transad_dp.h"
#define dir_transad DIR_TRANSAD_DP
#define DIR_TRANSAD DIR_TRANSAD_DP
