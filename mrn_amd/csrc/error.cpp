// Thread-local last-error string for the C ABI (mrn_last_error) and the library version.
#include <stdarg.h>
#include "common.hpp"

static thread_local char g_err[512] = "";

void mrn_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

MRN_EXPORT const char* mrn_last_error(void) { return g_err; }
MRN_EXPORT int mrn_version(void) { return 100; }
