/* TEST INFRASTRUCTURE ONLY -- see cuda.h in this directory. */
#include "cuda.h"
