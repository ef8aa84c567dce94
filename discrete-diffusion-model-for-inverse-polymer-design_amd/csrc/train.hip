// training-step kernels (backward passes, loss, optimizer) live here
#include "bd_common.h"
