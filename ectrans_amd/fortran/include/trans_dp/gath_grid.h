! Automatically generated interface header for backward compatibility of generic symbols !
#if defined(gath_grid)
#undef gath_grid
#endif
#if defined(GATH_GRID)
#undef GATH_GRID
#endif
#include "../gath_grid_dp.h"
#define gath_grid GATH_GRID_DP
#define GATH_GRID GATH_GRID_DP
