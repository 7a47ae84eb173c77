// Element types of the generic (VALU) kernels: load to / round in / store from the accumulate type.
// Shared by the forward (fa2_generic.hip) and backward (fa2_bwd_generic.hip) catch-all kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

namespace {

__device__ __forceinline__ float round_fmt(float x, int mbits, int emin, float maxv, bool has_inf) {
    if (x == 0.0f || isnan(x)) return x;
    if (isinf(x)) return has_inf ? x : NAN;
    int e;
    (void)frexpf(fabsf(x), &e);
    e -= 1;
    if (e < emin) e = emin;
    const float q = ldexpf(1.0f, e - mbits);
    float r = rintf(x / q) * q;  // RTNE; x/q exact
    if (fabsf(r) > maxv) r = has_inf ? copysignf(INFINITY, x) : NAN;
    return r;
}

struct ElemF64 {
    using acc_t = double;
    using st_t = double;
    static __device__ acc_t load(const void *p, int64_t i) { return ((const double *)p)[i]; }
    static __device__ acc_t round(acc_t x) { return x; }
    static __device__ void store(void *p, int64_t i, acc_t x) { ((double *)p)[i] = x; }
};
struct ElemF32 {
    using acc_t = float;
    static __device__ acc_t load(const void *p, int64_t i) { return ((const float *)p)[i]; }
    static __device__ acc_t round(acc_t x) { return x; }
    static __device__ void store(void *p, int64_t i, acc_t x) { ((float *)p)[i] = x; }
};
struct ElemF16 {
    using acc_t = float;
    static __device__ acc_t load(const void *p, int64_t i) { return (float)((const _Float16 *)p)[i]; }
    static __device__ acc_t round(acc_t x) { return (float)(_Float16)x; }
    static __device__ void store(void *p, int64_t i, acc_t x) { ((_Float16 *)p)[i] = (_Float16)x; }
};
struct ElemBF16 {
    using acc_t = float;
    static __device__ acc_t load(const void *p, int64_t i) {
        return __builtin_bit_cast(float, (uint32_t)((const uint16_t *)p)[i] << 16);
    }
    static __device__ uint16_t bits(float x) {  // RTNE, NaN stays NaN
        uint32_t u = __builtin_bit_cast(uint32_t, x);
        if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
        return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
    }
    static __device__ acc_t round(acc_t x) { return __builtin_bit_cast(float, (uint32_t)bits(x) << 16); }
    static __device__ void store(void *p, int64_t i, acc_t x) { ((uint16_t *)p)[i] = bits(x); }
};
struct ElemF8E5M2 {  // = the top byte of an fp16
    using acc_t = float;
    static __device__ acc_t load(const void *p, int64_t i) {
        const uint16_t h = (uint16_t)((const uint8_t *)p)[i] << 8;
        return (float)__builtin_bit_cast(_Float16, h);
    }
    static __device__ acc_t round(acc_t x) { return round_fmt(x, 2, -14, 57344.0f, true); }
    static __device__ void store(void *p, int64_t i, acc_t x) {
        const _Float16 h = (_Float16)round(x);  // exact: the e5m2 grid is a subset of fp16
        ((uint8_t *)p)[i] = (uint8_t)(__builtin_bit_cast(uint16_t, h) >> 8);
    }
};
struct ElemF8E4M3 {  // OCP e4m3fn: bias 7, no inf, S.1111.111 = NaN, max 448
    using acc_t = float;
    static __device__ acc_t load(const void *p, int64_t i) {
        const uint32_t b = ((const uint8_t *)p)[i];
        const uint32_t e = (b >> 3) & 15u, m = b & 7u;
        float v;
        if (e == 15u && m == 7u) v = NAN;
        else if (e == 0u) v = ldexpf((float)m, -9);
        else v = ldexpf((float)(8u + m), (int)e - 10);
        return (b & 0x80u) ? -v : v;
    }
    static __device__ acc_t round(acc_t x) { return round_fmt(x, 3, -6, 448.0f, false); }
    static __device__ void store(void *p, int64_t i, acc_t x) {
        const float r = round(x);
        uint8_t s = signbit(r) ? 0x80 : 0x00, out;
        const float a = fabsf(r);
        if (isnan(r)) out = 0x7f;
        else if (a == 0.0f) out = 0;
        else {
            int e;
            (void)frexpf(a, &e);
            e -= 1;
            if (e < -6) out = (uint8_t)ldexpf(a, 9);
            else out = (uint8_t)(((e + 7) << 3) | ((int)ldexpf(a, 3 - e) - 8));
        }
        ((uint8_t *)p)[i] = s | out;
    }
};

}  // namespace
