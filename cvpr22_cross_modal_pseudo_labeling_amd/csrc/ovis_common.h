// Shared host-side helpers for the libovis_hip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ovis_hip.h"

#define OVIS_HIP_TRY(expr)                         \
  do {                                             \
    hipError_t _e = (expr);                        \
    if (_e != hipSuccess) return (int)_e;          \
  } while (0)

// Launch-error check that does not synchronise: picks up invalid-configuration errors.
#define OVIS_LAUNCH_CHECK()                        \
  do {                                             \
    hipError_t _e = hipGetLastError();             \
    if (_e != hipSuccess) return (int)_e;          \
  } while (0)

static inline int ovis_ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// MI355X: 256 CUs in 8 XCDs. Used only to size grids, never for correctness.
#define OVIS_NUM_CU 256
#define OVIS_WAVE 64
