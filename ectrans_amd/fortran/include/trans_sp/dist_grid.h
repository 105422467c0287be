! Automatically generated interface header for backward compatibility of generic symbols !
#if defined(dist_grid)
#undef dist_grid
#endif
#if defined(DIST_GRID)
#undef DIST_GRID
#endif
#include "../dist_grid_sp.h"
#define dist_grid DIST_GRID_SP
#define DIST_GRID DIST_GRID_SP
