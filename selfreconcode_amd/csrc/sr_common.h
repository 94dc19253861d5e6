// Shared helpers for the gfx950 kernels of libselfrecon_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/selfrecon_hip.h"

#define SR_WAVE 64

static inline int sr_launch_status() {
  return hipGetLastError() == hipSuccess ? SR_OK : SR_ELAUNCH;
}

static inline int64_t sr_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Memory-bound kernels: cap the grid at 256 CUs x 8 blocks and grid-stride the rest.
static inline int sr_stream_grid(int64_t work_items, int block) {
  int64_t g = sr_cdiv(work_items, block);
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}
