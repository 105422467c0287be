! Automatically generated interface header for backward compatibility of generic symbols !
#if defined(trans_inq)
#undef trans_inq
#endif
#if defined(TRANS_INQ)
#undef TRANS_INQ
#endif
#include "../trans_inq_sp.h"
#define trans_inq TRANS_INQ_SP
#define TRANS_INQ TRANS_INQ_SP
