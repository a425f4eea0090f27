#include "errors.hpp"

#include "../../../include/rt/abi.h"

namespace rt {
namespace {
thread_local std::string g_last_error;
}
int set_error(int code, const std::string& msg) { g_last_error = msg; return code; }
const char* last_error() { return g_last_error.c_str(); }
}  // namespace rt

extern "C" const char* rt_last_error(void) { return rt::last_error(); }
extern "C" uint32_t rt_abi_version(void) { return RT_ABI_VERSION; }
