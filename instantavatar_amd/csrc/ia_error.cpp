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

// What this library was built from: "<translation unit>=<hash>;..." over compiler flags, shared headers and source
// (instantavatar_amd/build.py passes it; the marker lets build.py read it from the file without loading it).
#ifndef IA_SOURCE_MANIFEST
#define IA_SOURCE_MANIFEST ""
#endif
static const char ia_manifest[] = "IA_SOURCE_MANIFEST=" IA_SOURCE_MANIFEST;
extern "C" const char *ia_source_manifest(void) { return ia_manifest + sizeof("IA_SOURCE_MANIFEST=") - 1; }
