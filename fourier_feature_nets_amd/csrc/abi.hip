// ABI bookkeeping: version and the per-thread last-error string.
#include <string.h>

#include "common.h"

namespace ffn {
static thread_local char g_error[256] = "";

void set_error(const char* what, hipError_t code) {
    snprintf(g_error, sizeof(g_error), "%s: %s", what, hipGetErrorString(code));
}
}  // namespace ffn

extern "C" int ffn_abi_version(void) { return FFN_ABI_VERSION; }
extern "C" const char* ffn_last_error_string(void) { return ffn::g_error; }
