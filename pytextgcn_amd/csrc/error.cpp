// Thread-local error text behind tgcn_last_error() and the ABI version query.
#include <cstdarg>
#include <cstdio>

#include "common.h"

namespace tgcn {
namespace {
thread_local char g_error[512] = "";
}

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}
}  // namespace tgcn

extern "C" {
int tgcn_abi_version(void) { return TGCN_ABI_VERSION; }
const char *tgcn_last_error(void) { return tgcn::g_error; }
}
