! Automatically generated interface header for backward compatibility of generic symbols !
#if defined(vordiv_to_uv)
#undef vordiv_to_uv
#endif
#if defined(VORDIV_TO_UV)
#undef VORDIV_TO_UV
#endif
#include "../vordiv_to_uv_dp.h"
#define vordiv_to_uv VORDIV_TO_UV_DP
#define VORDIV_TO_UV VORDIV_TO_UV_DP
