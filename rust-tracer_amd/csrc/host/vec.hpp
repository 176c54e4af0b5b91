// vec.hpp -- host-side mirror of vec.rs: RFloat alias + Vector.  Same operation order, one rounding per op
// (the host is built with -ffp-contract=off).  Swap the alias with -DRTRACE_RFLOAT=double (vec.rs:6).
#pragma once
#include <cmath>

#ifndef RTRACE_RFLOAT
#define RTRACE_RFLOAT float
#endif

namespace rtrace {

using RFloat = RTRACE_RFLOAT;                                     // pub type RFloat = f32;  vec.rs:6

struct Vector {                                                   // vec.rs:8-13
    RFloat x = 0, y = 0, z = 0;

    Vector operator+(const Vector &r) const { return { x + r.x, y + r.y, z + r.z }; }     // vec.rs:15-27
    Vector operator-(const Vector &r) const { return { x - r.x, y - r.y, z - r.z }; }     // vec.rs:29-40
    Vector operator*(const Vector &r) const { return { x * r.x, y * r.y, z * r.z }; }     // vec.rs:42-53
    bool operator==(const Vector &r) const { return x == r.x && y == r.y && z == r.z; }
    bool operator!=(const Vector &r) const { return !(*this == r); }

    Vector mulfed(RFloat m) const { return { x * m, y * m, z * m }; }                     // vec.rs:57-63
    Vector &mulf(RFloat m) { x = x * m; y = y * m; z = z * m; return *this; }             // vec.rs:67-72
    RFloat dot(const Vector &r) const { return x * r.x + y * r.y + z * r.z; }             // vec.rs:77-79
    RFloat len() const { return std::sqrt(dot(*this)); }                                  // vec.rs:82-84
    Vector &normalize() { RFloat l = len(); return mulf(RFloat(1) / l); }                 // vec.rs:87-90 (recip)
    Vector normalized() const { return mulfed(RFloat(1) / len()); }                       // vec.rs:93-95
};

}  // namespace rtrace
