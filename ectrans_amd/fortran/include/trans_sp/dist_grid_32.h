! Automatically generated interface header for backward compatibility of generic symbols !
#if defined(dist_grid_32)
#undef dist_grid_32
#endif
#if defined(DIST_GRID_32)
#undef DIST_GRID_32
#endif
#include "../dist_grid_32_sp.h"
#define dist_grid_32 DIST_GRID_32_SP
#define DIST_GRID_32 DIST_GRID_32_SP
