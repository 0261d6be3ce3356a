// Library identification for the C ABI (include/gpp.h).
#include "gpp.h"

extern "C" const char* gpp_version(void) { return "gpp-hip 0.1.0 gfx950"; }
