! Automatically generated interface header for backward compatibility of generic symbols !
#if defined(dist_grid)
#undef dist_grid
#endif
#if defined(DIST_GRID)
#undef DIST_GRID
#endif
#include "../dist_grid_dp.h"
#define dist_grid DIST_GRID_DP
#define DIST_GRID DIST_GRID_DP
