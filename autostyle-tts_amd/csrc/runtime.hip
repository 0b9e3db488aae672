// runtime.hip -- error reporting and ABI version for libastts.so
#include "common.h"

namespace astts {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

}  // namespace astts

extern "C" {

int astts_abi_version(void) { return ASTTS_ABI_VERSION; }

const char* astts_last_error_string(void) { return astts::g_err; }

}  // extern "C"
