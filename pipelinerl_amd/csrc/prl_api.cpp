// libprl.so: ABI version and thread-local error reporting.
#include <atomic>
#include <cstdint>

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

namespace {
std::atomic<int64_t> g_tuning[PRL_TUNE_COUNT];
struct TuningInit {
  TuningInit() {
    for (auto& t : g_tuning) t.store(PRL_TUNE_UNSET, std::memory_order_relaxed);
  }
} g_tuning_init;
}  // namespace

int64_t tuning(int key, int64_t dflt) {
  const int64_t v = g_tuning[key].load(std::memory_order_relaxed);
  return v == PRL_TUNE_UNSET ? dflt : v;
}

}  // namespace prl

extern "C" int prl_set_tuning(int32_t key, int64_t value) {
  PRL_CHECK_ARG(key >= 0 && key < PRL_TUNE_COUNT, "unknown tuning key %d", key);
  prl::g_tuning[key].store(value, std::memory_order_relaxed);
  return PRL_OK;
}

extern "C" int prl_get_tuning(int32_t key, int64_t* value) {
  PRL_CHECK_ARG(key >= 0 && key < PRL_TUNE_COUNT && value, "unknown tuning key %d", key);
  *value = prl::g_tuning[key].load(std::memory_order_relaxed);
  return PRL_OK;
}

extern "C" int prl_abi_version(void) { return PRL_ABI_VERSION; }

extern "C" const char* prl_last_error(void) { return prl::error_buffer(); }
