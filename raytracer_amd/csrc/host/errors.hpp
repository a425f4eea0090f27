// Thread-local error message behind rt_last_error() (include/rt/abi.h).
#pragma once
#include <string>

namespace rt {
int set_error(int code, const std::string& msg);
const char* last_error();
}  // namespace rt
