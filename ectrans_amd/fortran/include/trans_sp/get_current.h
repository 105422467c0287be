#include "../get_current.h"
