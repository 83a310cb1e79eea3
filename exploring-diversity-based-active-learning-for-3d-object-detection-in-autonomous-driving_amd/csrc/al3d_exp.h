// Deterministic table-driven exp for the distance-map normalisation
// (1 - exp(-d), reference spatial_temporal_selector.py:142-144).
// x = (128 m + j) ln2/128 + r;  exp(x) = 2^m * T[j] * (1 + p(r)), every rounding
// explicit (fma only where written; the translation unit is built with
// -ffp-contract=off) so host and device agree bit for bit.  |err| < 0.51 ulp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "al3d_exp_table.h"

__device__ static const uint64_t k_al3d_exp_tab[128][2] = AL3D_EXP_TABLE_INIT;

__device__ __forceinline__ double al3d_exp_f64(double x)
{
    if (x != x) return x;
    if (x > 709.782712893384) return __builtin_inf();
    if (x < -745.1332191019412) return 0.0;
    double kd = rint(x * AL3D_EXP_INV_LN2N);
    long long k = (long long)kd;
    double r = fma(kd, -AL3D_EXP_LN2N_HI, x);
    r = fma(kd, -AL3D_EXP_LN2N_LO, r);
    long long j = k & 127, m = k >> 7;
    double r2 = r * r;
    double p = fma(r, 1.0 / 120.0, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r2, r);
    double th = __longlong_as_double((long long)k_al3d_exp_tab[j][0]);
    double tl = __longlong_as_double((long long)k_al3d_exp_tab[j][1]);
    double res = th + fma(th, p, tl);
    return ldexp(res, (int)m);
}

__device__ __forceinline__ float al3d_exp_f32(float x) { return (float)al3d_exp_f64((double)x); }
