#include "common.h"
#include "../../include/slender_hip.h"
extern "C" const char* sod_version(void) { return "slender_hip 0.1 (gfx950)"; }
