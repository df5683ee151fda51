// ia_error.cpp -- error reporting for the C ABI (thread-local message buffer).
#include <stdarg.h>
#include <stdio.h>

#include "../../include/instantavatar_hip.h"

thread_local char ia_err_buf[512] = "";

int ia_set_error(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(ia_err_buf, sizeof(ia_err_buf), fmt, ap);
  va_end(ap);
  return code;
}

extern "C" const char *ia_last_error(void) { return ia_err_buf; }
extern "C" int ia_version(void) { return 100; }
