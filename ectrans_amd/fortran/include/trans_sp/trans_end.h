! Automatically generated interface header for backward compatibility of generic symbols !
#if defined(trans_end)
#undef trans_end
#endif
#if defined(TRANS_END)
#undef TRANS_END
#endif
#include "../trans_end_sp.h"
#define trans_end TRANS_END_SP
#define TRANS_END TRANS_END_SP
