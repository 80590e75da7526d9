// host_util.h -- error reporting shared by the host-side translation units (internal).
#pragma once
#include <string>

namespace mimc3 {
// records the message for mimc3_last_error() (thread-local) and returns `code`
int fail(int code, const char *msg);
int fail(int code, const std::string &msg);
}  // namespace mimc3
