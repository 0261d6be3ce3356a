// Library identification for the C ABI (include/gpp.h): "gpp-hip <version> gfx950 src:<hash of csrc/ + include/gpp.h>".
// The hash (Makefile) changes with every source edit: measurements committed under profiles/ record it, and bench.py
// refuses to report a counter-derived figure that was collected with another build.
#include "gpp.h"

#ifndef GPP_SRC_HASH
#define GPP_SRC_HASH "unknown"
#endif

extern "C" const char* gpp_version(void) { return "gpp-hip 0.2.0 gfx950 src:" GPP_SRC_HASH; }
