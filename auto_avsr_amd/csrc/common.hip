// common.hip -- status plumbing of the C ABI (see include/avsr_hip.h).
#include "prims.h"
#include "avsr_hip.h"
#include <string.h>

static thread_local char g_err[256] = "";

extern "C" void avsr_set_error(const char* msg) {
    strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}
extern "C" const char* avsr_last_error(void) { return g_err; }
extern "C" int avsr_abi_version(void) { return 1; }
// 1 when this library is the host-side emulator build used by the CPU tests, 0 for the gfx950 build.
extern "C" int avsr_is_emulator(void) {
#ifdef AVSR_EMU
    return 1;
#else
    return 0;
#endif
}
