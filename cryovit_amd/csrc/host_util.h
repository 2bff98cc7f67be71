// Host-side error plumbing shared by the launchers (thread-local last-error string, HIP status checks).
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>

int cvx_fail(const char* msg);                    // records msg, returns CVX_ERR_ARG
int cvx_fail_hip(hipError_t e, const char* what); // records "what: <hip error>", returns CVX_ERR_HIP
int cvx_check_launch();                           // hipGetLastError() -> 0 / CVX_ERR_HIP

#define CVX_HIP(expr)                                       \
    do {                                                    \
        hipError_t _e = (expr);                             \
        if (_e != hipSuccess) return cvx_fail_hip(_e, #expr); \
    } while (0)
