// libprl.so: ABI version and thread-local error reporting.
#include "prl_common.h"

namespace prl {

char* error_buffer() {
  static thread_local char buf[512] = {0};
  return buf;
}

int set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(error_buffer(), 512, fmt, ap);
  va_end(ap);
  return code;
}

}  // namespace prl

extern "C" int prl_abi_version(void) { return PRL_ABI_VERSION; }

extern "C" const char* prl_last_error(void) { return prl::error_buffer(); }
