// theia_hip_core.hip -- device selection, error reporting, version.
#include <hip/hip_runtime.h>

#include <execinfo.h>
#include <signal.h>
#include <unistd.h>

#include <atomic>
#include <cstring>

#include "theia_hip.h"
#include "theia_hip_internal.h"

namespace thip {

static std::atomic<int> g_device{-1};

int ensure_device() {
  int dev = g_device.load();
  if (dev < 0) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
      return set_error(THEIA_HIP_ERR_NO_DEVICE, "no HIP device available (%s); the HIP backend has no CPU fallback",
                       e == hipSuccess ? "count = 0" : hipGetErrorString(e));
    dev = 0;
    g_device.store(0);
  }
  hipError_t e = hipSetDevice(dev);
  if (e != hipSuccess) return set_error(THEIA_HIP_ERR_NO_DEVICE, "hipSetDevice(%d) failed: %s", dev, hipGetErrorString(e));
  return 0;
}

}  // namespace thip

extern "C" {

int theia_hip_init(int device_ordinal) {
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0)
    return thip::set_error(THEIA_HIP_ERR_NO_DEVICE, "no HIP device available (%s)", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
  if (device_ordinal < 0 || device_ordinal >= count)
    return thip::set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "device ordinal %d out of range [0,%d)", device_ordinal, count);
  thip::g_device.store(device_ordinal);
  return thip::ensure_device();
}

int theia_hip_shutdown(void) {
  theia_hip_release_scratch();   // the block caches (up to 6 GiB of device and 2 GiB of pinned host memory) go back to the runtime
  thip::g_device.store(-1);
  return 0;
}

int theia_hip_device_count(int* count) {
  if (!count) return thip::set_error(THEIA_HIP_ERR_INVALID_ARGUMENT, "null count");
  *count = 0;
  hipError_t e = hipGetDeviceCount(count);
  if (e != hipSuccess) { *count = 0; return thip::set_error(THEIA_HIP_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
  return 0;
}

const char* theia_hip_last_error(void) { return thip::g_last_error.c_str(); }

const char* theia_hip_version(void) { return "pytheiasfm_amd 0.1.0 (gfx950)"; }

// Development aid (THEIA_HIP_ABORT_TRACE=1 in pytheiasfm_amd/_capi.py): native backtrace on SIGABRT / SIGSEGV.
static void thip_abort_trace(int sig) {
  void* frames[64];
  const int n = backtrace(frames, 64);
  const char msg[] = "theia_hip: fatal signal, native backtrace:\n";
  (void)!write(2, msg, sizeof(msg) - 1);
  backtrace_symbols_fd(frames, n, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}
void theia_hip_debug_install_abort_trace(void) {
  signal(SIGABRT, thip_abort_trace);
  signal(SIGSEGV, thip_abort_trace);
}

}  // extern "C"
