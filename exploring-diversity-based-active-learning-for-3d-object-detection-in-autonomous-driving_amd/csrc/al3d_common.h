// Shared host-side helpers for the C-ABI entry points (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/al3d.h"

#define AL3D_WAVE 64

extern thread_local char g_al3d_err[512];

static inline int al3d_fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_al3d_err, sizeof(g_al3d_err), fmt, ap);
    va_end(ap);
    return code;
}

#define AL3D_REQUIRE(cond, ...) \
    do { if (!(cond)) return al3d_fail(AL3D_EINVAL, __VA_ARGS__); } while (0)

#define AL3D_CHECK_LAUNCH(name)                                                        \
    do {                                                                               \
        hipError_t e_ = hipGetLastError();                                             \
        if (e_ != hipSuccess)                                                          \
            return al3d_fail(AL3D_ELAUNCH, "%s: %s", name, hipGetErrorString(e_));     \
    } while (0)

static inline int64_t al3d_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t al3d_align(int64_t a, int64_t b) { return al3d_cdiv(a, b) * b; }
