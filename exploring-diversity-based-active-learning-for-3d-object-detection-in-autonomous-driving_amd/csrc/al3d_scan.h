// Internal (not part of the C ABI): device-wide exclusive prefix sum of int32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

int64_t al3d_scan_workspace_bytes(int64_t n);
// out[i] = sum(in[0..i)); in/out may alias; ws >= al3d_scan_workspace_bytes(n).
int al3d_exclusive_scan_i32(const int* in, int* out, int64_t n, void* ws, hipStream_t stream);
