// Library-level entry points: ABI version, error text, device query.
#include <string.h>

#include "common.hpp"

namespace rc {
static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
int fail(int code, const std::string& msg) { g_err = msg; return code; }
}  // namespace rc

extern "C" {

int rc_abi_version(void) { return RC_ABI_VERSION; }

const char* rc_last_error(void) { return rc::g_err.c_str(); }

const char* rc_build_info(void) { return "librealcam_hip gfx950 (" __VERSION__ ")"; }

int rc_device_arch(char* buf, size_t buflen) {
    RC_REQUIRE(buf != nullptr && buflen > 0, "rc_device_arch: null buffer");
    int dev = 0;
    RC_HIP_CHECK(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    RC_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    strncpy(buf, prop.gcnArchName, buflen - 1);
    buf[buflen - 1] = '\0';
    return RC_OK;
}

}  // extern "C"
