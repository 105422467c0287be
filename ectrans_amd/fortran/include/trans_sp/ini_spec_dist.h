#include "../ini_spec_dist.h"
