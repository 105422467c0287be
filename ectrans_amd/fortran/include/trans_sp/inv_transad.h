! Automatically generated interface header for backward compatibility of generic symbols !
#if defined(inv_transad)
#undef inv_transad
#endif
#if defined(INV_TRANSAD)
#undef INV_TRANSAD
#endif
#include "../inv_transad_sp.h"
#define inv_transad INV_TRANSAD_SP
#define INV_TRANSAD INV_TRANSAD_SP
