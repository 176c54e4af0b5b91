// rt_math.hpp -- the reference's Vector / Ray / Sphere arithmetic, restated for gfx950 device code.
//
// Every function keeps the reference's operation ORDER and is compiled with -ffp-contract=off, so each
// + - * / sqrt rounds once, exactly like the Rust CPU path (an FMA changes 0.24 % of the pixels).
// T is float (RFloat = f32, vec.rs:6) or double (the type-alias swap).
#pragma once
#include <hip/hip_runtime.h>
#include <limits>

namespace rt {

template <typename T> struct V3 { T x, y, z; };
template <typename T> struct alignas(sizeof(T) * 4) Item { T cx, cy, cz, r; };   // Sphere{center, radius} primitive.rs:38-42

template <typename T> __host__ __device__ __forceinline__ constexpr T inf() { return std::numeric_limits<T>::infinity(); }
template <typename T> __host__ __device__ __forceinline__ constexpr T eps() { return std::numeric_limits<T>::epsilon(); }

__device__ __forceinline__ float rsqrt_exact(float x) { return __builtin_sqrtf(x); }    // IEEE correctly rounded
__device__ __forceinline__ double rsqrt_exact(double x) { return __builtin_sqrt(x); }

// Correctly rounded f32 sqrt for the traversal loops.  Same result as rsqrt_exact() for every input (checked exhaustively on
// the device by rt_selftest_sqrt), a third of the instructions of the compiler's expansion: y = v_rsq_f32(x) is within 1 ulp
// of 1/sqrt(x); g = x*y is then within a few ulp of the root, r = x - g*g is exact in one FMA, and g + r*(y/2) in one more FMA
// rounds to the IEEE root (Markstein's final step: the last FMA sees the exact residual).  Zeros, denormals and tiny values
// of either sign (denormal residuals would lose bits), +inf (0 * inf) and NaN take the general path; that branch is almost
// never taken.
__device__ __forceinline__ float sqrt_rn_lean(float x)
{
    if (__builtin_expect(!(__builtin_fabsf(x) >= 0x1p-96f && x < __builtin_huge_valf()), 0)) return __builtin_sqrtf(x);
    const float y = __builtin_amdgcn_rsqf(x);
    const float g = x * y, h = 0.5f * y;
    const float r = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(r, h, g);
}
__device__ __forceinline__ double sqrt_rn_lean(double x) { return __builtin_sqrt(x); }

// Correctly rounded f32 reciprocal, 3 instructions instead of the 11 of the compiler's IEEE division: y = v_rcp_f32(x) is within
// 1 ulp of 1/x, e = 1 - x*y is exact in one FMA and y + e*y in one more rounds to RN(1/x).  Same bits as `1.0f / x` for every x
// with 2^-100 <= |x| <= 2^100 (all 3.4e9 of them compared on the device: rt_selftest_rcp); anything else takes the division.
__device__ __forceinline__ float rcp_rn_lean(float x)
{
    if (__builtin_expect(!(__builtin_fabsf(x) >= 0x1p-100f && __builtin_fabsf(x) <= 0x1p100f), 0)) return 1.0f / x;
    const float y = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, y, 1.0f);
    return __builtin_fmaf(e, y, y);
}
__device__ __forceinline__ double rcp_rn_lean(double x) { return 1.0 / x; }

// vec.rs:15-72
template <typename T> __device__ __forceinline__ V3<T> add(V3<T> a, V3<T> b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
template <typename T> __device__ __forceinline__ V3<T> sub(V3<T> a, V3<T> b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
template <typename T> __device__ __forceinline__ V3<T> mulf(V3<T> a, T m) { return { a.x * m, a.y * m, a.z * m }; }
// vec.rs:77-79: x*x' + y*y' + z*z' evaluated left to right
template <typename T> __device__ __forceinline__ T dot(V3<T> a, V3<T> b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
// vec.rs:87-95: multiply by len.recip(), a true division
template <typename T> __device__ __forceinline__ V3<T> normalized(V3<T> a)
{
    T len = sqrt_rn_lean(dot(a, a));       // == rsqrt_exact for every input (rt_selftest_sqrt), fewer instructions in f32
    return mulf(a, T(1.0) / len);
}
// f32: one range test for both lean sequences (|a|^2 in [2^-96, inf) puts the length in [2^-48, 2^64), inside rcp_rn_lean's range);
// the general path is a real branch -- left to itself the compiler evaluates both sides and selects (16 more instructions per ray).
#ifndef RT_NORMALIZED_GENERAL              // -DRT_NORMALIZED_GENERAL builds the generic version everywhere (A/B timing only)
template <> __device__ __forceinline__ V3<float> normalized<float>(V3<float> a)
{
    const float x = dot(a, a);
    float inv;
    if (__builtin_expect(x >= 0x1p-96f && x < __builtin_huge_valf(), 1)) {
        const float y = __builtin_amdgcn_rsqf(x);
        const float g = x * y, h = 0.5f * y;
        const float len = __builtin_fmaf(__builtin_fmaf(-g, g, x), h, g);          // sqrt_rn_lean
        const float z = __builtin_amdgcn_rcpf(len);
        inv = __builtin_fmaf(__builtin_fmaf(-len, z, 1.0f), z, z);                 // rcp_rn_lean
    } else {
        asm volatile("; normalized: general path" ::: "memory");                   // not to be if-converted
        inv = 1.0f / __builtin_sqrtf(x);
    }
    return mulf(a, inv);
}
#endif

// primitive.rs:55-72 Sphere::distance_from_ray, comparisons kept in the reference's sense (NaN falls through
// exactly as `if disc < 0.0 { return INF }` lets it).
template <typename T>
__device__ __forceinline__ T distance_from_ray(T cx, T cy, T cz, T radius, V3<T> o, V3<T> d)
{
    V3<T> v = { cx - o.x, cy - o.y, cz - o.z };
    T b = dot(v, d);
    T disc = (b * b - dot(v, v)) + radius * radius;
    if (disc < T(0.0)) return inf<T>();
    T s = rsqrt_exact(disc);
    T t2 = b + s;
    if (t2 < T(0.0)) return inf<T>();
    T t1 = b - s;
    return t1 > T(0.0) ? t1 : t2;
}

// render.rs:96-103, the `scale` closure of set_pixel_from_vector: 0.5 + 255 v, clamp above, truncate.
template <typename T> __device__ __forceinline__ unsigned scale_u8(T v)
{
    T r = T(0.5) + T(255.0) * v;
    if (r > T(255.0)) return 255u;
    if (!(r > T(0.0))) return 0u;          // Rust `as u8` saturates, NaN -> 0
    return (unsigned)r;                    // toward zero
}
// Same value without branches: v_cvt_u32_f32 truncates toward zero and saturates (r <= 0 and NaN -> 0, huge -> 2^32-1), and
// every r > 255 truncates to >= 255.
template <> __device__ __forceinline__ unsigned scale_u8<float>(float v)
{
    const float r = 0.5f + 255.0f * v;
    unsigned u;
    asm("v_cvt_u32_f32_e32 %0, %1" : "=v"(u) : "v"(r));
    return min(u, 255u);
}

}  // namespace rt
