#include "../setup_trans0.h"
