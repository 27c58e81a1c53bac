// common.hip -- status plumbing of the C ABI (see include/avsr_hip.h).
#include "prims.h"
#include "avsr_hip.h"
#include <string.h>
#include <stdio.h>

static thread_local char g_err[256] = "";

extern "C" void avsr_set_error(const char* msg) {
    strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}
extern "C" void avsr_set_error2(const char* where, const char* what) {
    snprintf(g_err, sizeof(g_err), "%s: %s", where ? where : "", what ? what : "");
}
static thread_local int g_launch_err = 0;
extern "C" void avsr_note_launch(int err) {
    if (err != 0 && g_launch_err == 0) g_launch_err = err;
}
extern "C" int avsr_take_launch_error(void) {
    int e = g_launch_err;
    g_launch_err = 0;
    return e;
}
extern "C" const char* avsr_last_error(void) { return g_err; }
extern "C" int avsr_abi_version(void) { return 1; }
// process-wide tuning knobs (benchmarks; see avsr_tune in avsr_hip.h)
int avsr_tune_knobs[32] = {0};
extern "C" int avsr_tune(int knob, int value) {
    if (knob < 0 || knob >= 32) {
        avsr_set_error("tune: unknown knob");
        return 1;
    }
    avsr_tune_knobs[knob] = value;
    return 0;
}
// 1 when this library is the host-side emulator build used by the CPU tests, 0 for the gfx950 build.
extern "C" int avsr_is_emulator(void) {
#ifdef AVSR_EMU
    return 1;
#else
    return 0;
#endif
}
