// api.hip -- version / error reporting of the C ABI (include/ofasr.h).
#include <stdarg.h>
#include <string.h>
#include "ofasr_common.h"

namespace ofasr {
static thread_local char g_err[512] = {0};
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace ofasr

OFASR_EXPORT int ofasr_version(void) { return OFASR_VERSION; }
OFASR_EXPORT const char* ofasr_last_error_string(void) { return ofasr::g_err; }
OFASR_EXPORT const char* ofasr_status_string(int status) {
    switch (status) {
        case OFASR_OK: return "ok";
        case OFASR_ERR_INVALID_ARG: return "invalid argument";
        case OFASR_ERR_UNSUPPORTED: return "unsupported shape or dtype";
        case OFASR_ERR_WORKSPACE: return "workspace missing or too small";
        case OFASR_ERR_LAUNCH: return "kernel launch failed";
        default: return "unknown status";
    }
}
