// theia_hip_internal.h -- shared by the translation units of libtheia_hip.so
#pragma once
#include <string>

namespace thip {
extern thread_local std::string g_last_error;
int set_error(int code, const char* fmt, ...);
int ensure_device();  // lazily selects device 0 unless theia_hip_init chose one
}  // namespace thip
