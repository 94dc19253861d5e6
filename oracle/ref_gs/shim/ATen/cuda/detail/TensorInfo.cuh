/* TEST INFRASTRUCTURE ONLY: see sr_aten_shim.h */
#include "../../../sr_aten_shim.h"
