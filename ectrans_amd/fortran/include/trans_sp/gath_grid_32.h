! Automatically generated interface header for backward compatibility of generic symbols !
#if defined(gath_grid_32)
#undef gath_grid_32
#endif
#if defined(GATH_GRID_32)
#undef GATH_GRID_32
#endif
#include "../gath_grid_32_sp.h"
#define gath_grid_32 GATH_GRID_32_SP
#define GATH_GRID_32 GATH_GRID_32_SP
