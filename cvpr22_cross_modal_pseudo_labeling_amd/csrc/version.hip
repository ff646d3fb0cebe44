#include "ovis_common.h"
extern "C" const char* ovis_version(void) { return "ovis_hip 0.1 gfx950"; }
