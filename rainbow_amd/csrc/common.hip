// common.hip — error string + ABI version for librainbow_hip.so.
#include "rb_common.h"

#include <string.h>

static thread_local char g_rb_error[1024] = "";

void rb_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_rb_error, sizeof(g_rb_error), fmt, ap);
  va_end(ap);
}

extern "C" {
const char* rb_last_error(void) { return g_rb_error; }
int rb_abi_version(void) { return 1; }
}
