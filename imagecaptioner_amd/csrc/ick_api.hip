// ick_api.hip — library-level entry points (ABI version, error text).
#include "ick_common.h"

namespace ick {
char* err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}
}  // namespace ick

extern "C" const char* ick_last_error(void) { return ick::err_buf(); }
extern "C" int ick_abi_version(void) { return ICK_ABI_VERSION; }
