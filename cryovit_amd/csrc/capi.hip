// C-ABI housekeeping: version, device probe, thread-local error string.
#include "../../include/cryovit_hip.h"
#include "host_util.h"
#include <string.h>

static thread_local char g_err[512] = "";

int cvx_fail(const char* msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return CVX_ERR_ARG;
}
int cvx_fail_hip(hipError_t e, const char* what) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return CVX_ERR_HIP;
}
int cvx_check_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : cvx_fail_hip(e, "kernel launch");
}

extern "C" const char* cvx_last_error(void) { return g_err; }
extern "C" int cvx_version(void) { return 1; }
extern "C" int cvx_device_arch(char* buf, int buflen) {
    if (!buf || buflen <= 0) return cvx_fail("cvx_device_arch: bad buffer");
    buf[0] = 0;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
        (void)hipGetLastError();
        return 0;
    }
    int dev = 0;
    hipDeviceProp_t p;
    CVX_HIP(hipGetDevice(&dev));
    CVX_HIP(hipGetDeviceProperties(&p, dev));
    snprintf(buf, buflen, "%s", p.gcnArchName);
    return 0;
}
