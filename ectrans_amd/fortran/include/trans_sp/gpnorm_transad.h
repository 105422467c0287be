! Automatically generated interface header for backward compatibility of generic symbols !
#if defined(gpnorm_transad)
#undef gpnorm_transad
#endif
#if defined(GPNORM_TRANSAD)
#undef GPNORM_TRANSAD
#endif
#include "../gpnorm_transad_sp.h"
#define gpnorm_transad GPNORM_TRANSAD_SP
#define GPNORM_TRANSAD GPNORM_TRANSAD_SP
