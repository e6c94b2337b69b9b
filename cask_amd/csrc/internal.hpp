// Shared between the translation units of libcask_hip.so (not installed).
#pragma once
#include <string>

namespace caskhip {
// records the thread-local message behind cask_hip_last_error() and returns `code`
int report_failure(int code, const std::string &msg);
}  // namespace caskhip
