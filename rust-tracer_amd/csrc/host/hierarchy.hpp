// hierarchy.hpp -- the automatic bounding-sphere hierarchy for an ARBITRARY sphere list (not in the reference, whose only
// scene builder is `pyramid`, group.rs:28-65).  ONE implementation for both hosts: csrc/host/scene.hpp (rtrace --scene) includes it, and the
// library exports it as rt_build_hierarchy (include/rtrace_hip.h), which scene.py's build_hierarchy calls; scene.py keeps a plain
// restatement of the same arithmetic (build_hierarchy_reference) that the tests hold against it.
//
// The tree is an input of the renderer, not reference arithmetic: whatever it is, the walk over it is the reference's.  What it decides is
// how many tests a ray makes.  Round 6 (NOTES.md R5 had measured it on the 100,000-sphere scene: 125.6 -> 110.3 tests per
// ray): (a) a group's bound is a near-minimal enclosing sphere -- the box-centre sphere improved by Badoiu-Clarkson steps towards the
// farthest sphere surface -- instead of the sphere around the box centre, (b) a group's two halves are visited nearer-to-the-eye first, so
// that a ray's hit.distance is small when it reaches the farther half and the reference's cull rule (group.rs:73) cuts it.
//
// All arithmetic in double, every operation written out in the order scene.py's restatement performs it (no contraction: the hosts are
// built with -ffp-contract=off): both produce the same items, bounds and ranges bit for bit.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <numeric>
#include <stdexcept>
#include <vector>

namespace rt_host {

struct FlatHierarchy {
    std::vector<double> items;      // 4 per item (cx, cy, cz, r), DFS order
    std::vector<double> bounds;     // 4 per group, DFS pre-order
    std::vector<int32_t> ranges;    // 2 per group: first item, item count
    std::vector<uint64_t> order;    // items[k] = spheres[order[k]]
};

constexpr int kTightSteps = 20;     // Badoiu-Clarkson steps per bound

// Enclosing sphere of the spheres sp[idx]: out = {cx, cy, cz, radius}; radius inflated by 1e-4 so that it still encloses after rounding to f32.
inline void enclosing_sphere(const double *sp, const std::vector<size_t> &idx, int steps, double out[4])
{
    double lo[3] = { 1e300, 1e300, 1e300 }, hi[3] = { -1e300, -1e300, -1e300 };
    for (size_t i : idx)
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::min(lo[k], sp[4 * i + k] - sp[4 * i + 3]);
            hi[k] = std::max(hi[k], sp[4 * i + k] + sp[4 * i + 3]);
        }
    double c[3] = { (lo[0] + hi[0]) * 0.5, (lo[1] + hi[1]) * 0.5, (lo[2] + hi[2]) * 0.5 };
    auto reach = [&](size_t i) {                                    // |c_i - c| + r_i
        const double dx = sp[4 * i] - c[0], dy = sp[4 * i + 1] - c[1], dz = sp[4 * i + 2] - c[2];
        return std::sqrt((dx * dx + dy * dy) + dz * dz) + sp[4 * i + 3];
    };
    for (int k = 1; k <= steps && idx.size() > 1; ++k) {
        size_t j = idx[0];
        double far = reach(j);
        for (size_t i : idx) { const double d = reach(i); if (d > far) { far = d; j = i; } }      // the first of the farthest
        const double vx = sp[4 * j] - c[0], vy = sp[4 * j + 1] - c[1], vz = sp[4 * j + 2] - c[2];
        const double n = std::sqrt((vx * vx + vy * vy) + vz * vz);
        if (n == 0.0) break;
        const double fx = sp[4 * j] + (vx / n) * sp[4 * j + 3], fy = sp[4 * j + 1] + (vy / n) * sp[4 * j + 3], fz = sp[4 * j + 2] + (vz / n) * sp[4 * j + 3];
        const double w = (double)(k + 1);
        c[0] = c[0] + (fx - c[0]) / w; c[1] = c[1] + (fy - c[1]) / w; c[2] = c[2] + (fz - c[2]) / w;
    }
    double rad = 0.0;
    for (size_t i : idx) rad = std::max(rad, reach(i));
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2];
    out[3] = rad * (1.0 + 1e-4) + 1e-30;
}

namespace detail {
inline void build(const double *sp, std::vector<size_t> &idx, const double bound[4], size_t leaf_size, const double *eye, int steps, FlatHierarchy &h)
{
    const size_t gi = h.ranges.size() / 2;
    h.bounds.insert(h.bounds.end(), bound, bound + 4);
    h.ranges.push_back((int32_t)h.order.size());
    h.ranges.push_back(0);
    if (idx.size() <= leaf_size) {
        for (size_t i : idx) { h.order.push_back(i); h.items.insert(h.items.end(), sp + 4 * i, sp + 4 * i + 4); }
    } else {
        double cmin[3] = { 1e300, 1e300, 1e300 }, cmax[3] = { -1e300, -1e300, -1e300 };
        for (size_t i : idx)
            for (int k = 0; k < 3; ++k) { cmin[k] = std::min(cmin[k], sp[4 * i + k]); cmax[k] = std::max(cmax[k], sp[4 * i + k]); }
        int axis = 0;                                             // the first axis of the largest extent
        for (int k = 1; k < 3; ++k) if (cmax[k] - cmin[k] > cmax[axis] - cmin[axis]) axis = k;
        std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return sp[4 * a + axis] < sp[4 * b + axis]; });
        const size_t half = idx.size() / 2;
        std::vector<size_t> part[2] = { std::vector<size_t>(idx.begin(), idx.begin() + half), std::vector<size_t>(idx.begin() + half, idx.end()) };
        std::vector<size_t>().swap(idx);
        double b[2][4];
        enclosing_sphere(sp, part[0], steps, b[0]);
        enclosing_sphere(sp, part[1], steps, b[1]);
        int first = 0;
        if (eye) {                                                // the half whose bound's centre is nearer to the eye first (a tie keeps the split's order)
            double key[2];
            for (int k = 0; k < 2; ++k) {
                const double dx = b[k][0] - eye[0], dy = b[k][1] - eye[1], dz = b[k][2] - eye[2];
                key[k] = (dx * dx + dy * dy) + dz * dz;
            }
            if (key[1] < key[0]) first = 1;
        }
        build(sp, part[first], b[first], leaf_size, eye, steps, h);
        build(sp, part[1 - first], b[1 - first], leaf_size, eye, steps, h);
    }
    h.ranges[2 * gi + 1] = (int32_t)(h.order.size() - (size_t)h.ranges[2 * gi]);
}
}  // namespace detail

// spheres: 4 doubles each.  eye: NULL = the split's own order.  steps: kTightSteps, or 0 for round 5's box-centre bounds.
inline FlatHierarchy build_hierarchy(const double *spheres, size_t n, size_t leaf_size, const double *eye, int steps = kTightSteps)
{
    if (n == 0) throw std::invalid_argument("build_hierarchy needs at least one sphere");
    FlatHierarchy h;
    h.items.reserve(4 * n); h.order.reserve(n);
    std::vector<size_t> idx(n);
    std::iota(idx.begin(), idx.end(), (size_t)0);
    double b[4];
    enclosing_sphere(spheres, idx, steps, b);
    detail::build(spheres, idx, b, std::max<size_t>(1, leaf_size), eye, steps, h);
    return h;
}

}  // namespace rt_host
