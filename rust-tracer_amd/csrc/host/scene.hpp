// scene.hpp -- host-side mirror of primitive.rs / group.rs / Scene (render.rs:138-167): the Scene is built and
// owned on the host exactly as the reference does; flatten() turns the group tree into the DFS arrays the C ABI
// consumes (items in traversal order + each group's bound and item range, pre-order).
#pragma once
#include <algorithm>
#include <array>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <numeric>
#include <stdexcept>
#include <string>
#include <vector>

#include "vec.hpp"
#include "hierarchy.hpp"

namespace rtrace {

struct Sphere {                                                   // primitive.rs:38-51
    Vector center;
    RFloat radius = 1.0;
};

struct SphericalGroup;
struct Pair {                                                     // enum Pair<I, G>  group.rs:7-10
    bool is_group = false;
    Sphere item;
    std::unique_ptr<SphericalGroup> group;
};

struct SphericalGroup {                                           // TypedGroup<Sphere, Sphere>  group.rs:17-20, 86
    Sphere bound;
    std::vector<Pair> children;

    // group.rs:28-56
    static Pair pyramid_recursive(uint32_t level, const Vector &p, RFloat r)
    {
        Pair out;
        Sphere s{ p, r };
        if (level == 1) { out.item = s; return out; }
        auto g = std::make_unique<SphericalGroup>();
        g->children.reserve(5);
        Pair own; own.item = s;
        g->children.push_back(std::move(own));
        g->bound.center = p;
        g->bound.radius = RFloat(3.0) * r;
        const RFloat rn = RFloat(3.0) * r / std::sqrt(RFloat(12.0));
        for (int dz : { -1, 1 })
            for (int dx : { -1, 1 }) {
                const Vector np = p + Vector{ RFloat(dx) * rn, rn, RFloat(dz) * rn };
                g->children.push_back(pyramid_recursive(level - 1, np, r * RFloat(0.5)));
            }
        out.is_group = true;
        out.group = std::move(g);
        return out;
    }

    // group.rs:58-65
    static std::unique_ptr<SphericalGroup> pyramid(uint32_t level, const Vector &origin, RFloat radius)
    {
        if (!(level > 1)) throw std::invalid_argument("Levels equal or smaller than one cause empty groups");
        return std::move(pyramid_recursive(level, origin, radius).group);
    }

    // Bounding-sphere hierarchy for an ARBITRARY sphere list (not in the reference, whose only builder is `pyramid`):
    // csrc/host/hierarchy.hpp -- median splits along the longest axis of the centres until at most leaf_size spheres remain, near-minimal
    // enclosing spheres as bounds, the half nearer to `eye` first.  The result is an ordinary TypedGroup tree.  The same code is what
    // scene.py's build_hierarchy runs (the library exports it: rt_build_hierarchy), so both hosts produce the same items, bounds and
    // ranges bit for bit (tests/test_host_and_abi.py).
    using Sphere4 = std::array<double, 4>;
    static std::unique_ptr<SphericalGroup> from_spheres_auto(const std::vector<Sphere4> &sp, size_t leaf_size = 4, const Vector *eye = nullptr)
    {
        if (sp.empty()) throw std::invalid_argument("build_hierarchy needs at least one sphere");
        const double e[3] = { eye ? (double)eye->x : 0.0, eye ? (double)eye->y : 0.0, eye ? (double)eye->z : 0.0 };
        const rt_host::FlatHierarchy h = rt_host::build_hierarchy(&sp[0][0], sp.size(), leaf_size, eye ? e : nullptr);
        size_t gi = 0;
        return from_flat(h, gi);
    }

    // TypedGroup::count  group.rs:93-109 -> (num_groups, num_items)
    void count(size_t &ng, size_t &ni) const
    {
        ng += 1;
        for (const Pair &c : children) {
            if (c.is_group) c.group->count(ng, ni); else ni += 1;
        }
    }

private:
    static std::unique_ptr<SphericalGroup> from_flat(const rt_host::FlatHierarchy &h, size_t &gi)
    {
        auto g = std::make_unique<SphericalGroup>();
        const size_t me = gi++, n_groups = h.ranges.size() / 2;
        g->bound.center = Vector{ (RFloat)h.bounds[4 * me], (RFloat)h.bounds[4 * me + 1], (RFloat)h.bounds[4 * me + 2] };
        g->bound.radius = (RFloat)h.bounds[4 * me + 3];
        const size_t first = (size_t)h.ranges[2 * me], end = first + (size_t)h.ranges[2 * me + 1];
        auto item_child = [&](size_t k) {
            Pair p;
            p.item = Sphere{ Vector{ (RFloat)h.items[4 * k], (RFloat)h.items[4 * k + 1], (RFloat)h.items[4 * k + 2] }, (RFloat)h.items[4 * k + 3] };
            g->children.push_back(std::move(p));
        };
        size_t item = first;
        while (gi < n_groups && (size_t)h.ranges[2 * gi] < end) {            // pre-order: the next group is mine while it starts inside my range
            for (; item < (size_t)h.ranges[2 * gi]; ++item) item_child(item);
            const size_t child_end = (size_t)h.ranges[2 * gi] + (size_t)h.ranges[2 * gi + 1];
            Pair p;
            p.is_group = true;
            p.group = from_flat(h, gi);
            g->children.push_back(std::move(p));
            item = child_end;
        }
        for (; item < end; ++item) item_child(item);
        return g;
    }
};

struct FlatScene {                                                // what rt_scene_create takes
    std::vector<RFloat> items;                                    // 4 per item: cx, cy, cz, r  (DFS order)
    std::vector<RFloat> bounds;                                   // 4 per group, pre-order
    std::vector<int32_t> ranges;                                  // 2 per group: first item, item count
};

struct Scene {                                                    // render.rs:138-142
    std::unique_ptr<SphericalGroup> group;
    Vector directional_light;
    Vector eye;

    static Scene with_level(uint32_t level)                       // Scene::default() is with_level(8)  render.rs:144-166
    {
        Scene s;
        s.group = SphericalGroup::pyramid(level, Vector{ 0.0, -1.0, 0.0 }, 1.0);
        s.directional_light = Vector{ -1.0, -3.0, 2.0 }.normalized();
        s.eye = Vector{ 0.0, 0.0, -4.0 };
        return s;
    }
    static Scene default_scene() { return with_level(8); }

    // A sphere list from a file, with an automatically built hierarchy (SphericalGroup::from_spheres_auto).  Text: one sphere per
    // line `cx cy cz r`; optional lines `light x y z` (direction the light travels, normalised here like Scene::default does,
    // render.rs:154-159) and `eye x y z`; `#` starts a comment.  A file whose name ends in .f32 holds raw little-endian f32
    // quadruples instead.
    static Scene from_file(const std::string &path, size_t leaf_size = 4)
    {
        std::vector<SphericalGroup::Sphere4> sp;
        Vector light{ -1.0, -3.0, 2.0 }, eye{ 0.0, 0.0, -4.0 };
        FILE *f = fopen(path.c_str(), "rb");
        if (!f) throw std::runtime_error("cannot open scene file " + path);
        const bool raw = path.size() > 4 && path.compare(path.size() - 4, 4, ".f32") == 0;
        if (raw) {
            float q[4];
            while (fread(q, sizeof(float), 4, f) == 4) sp.push_back({ (double)q[0], (double)q[1], (double)q[2], (double)q[3] });
        } else {
            char line[512];
            size_t ln = 0;
            while (fgets(line, sizeof line, f)) {
                ++ln;
                if (char *h = strchr(line, '#')) *h = '\0';
                char *p = line;
                while (*p == ' ' || *p == '\t') ++p;
                if (*p == '\0' || *p == '\n' || *p == '\r') continue;
                double v[4];
                int n = 0;
                const bool is_light = strncmp(p, "light", 5) == 0, is_eye = strncmp(p, "eye", 3) == 0;
                if (is_light) p += 5; else if (is_eye) p += 3;
                for (; n < 4; ++n) {
                    char *end = nullptr;
                    v[n] = strtod(p, &end);
                    if (end == p) break;
                    p = end;
                }
                if ((is_light || is_eye) ? n != 3 : n != 4) {
                    fclose(f);
                    throw std::runtime_error(path + ":" + std::to_string(ln) + ": expected `cx cy cz r`, `light x y z` or `eye x y z`");
                }
                if (is_light) light = Vector{ (RFloat)v[0], (RFloat)v[1], (RFloat)v[2] };
                else if (is_eye) eye = Vector{ (RFloat)v[0], (RFloat)v[1], (RFloat)v[2] };
                else sp.push_back({ v[0], v[1], v[2], v[3] });
            }
        }
        fclose(f);
        if (sp.empty()) throw std::runtime_error("scene file " + path + " holds no sphere");
        Scene s;
        s.group = SphericalGroup::from_spheres_auto(sp, leaf_size, &eye);
        s.directional_light = light.normalized();
        s.eye = eye;
        return s;
    }

    FlatScene flatten() const
    {
        FlatScene f;
        flatten_rec(*group, f);
        return f;
    }

private:
    static void flatten_rec(const SphericalGroup &g, FlatScene &f)
    {
        const size_t bi = f.ranges.size() / 2;
        const int32_t first = (int32_t)(f.items.size() / 4);
        f.bounds.insert(f.bounds.end(), { g.bound.center.x, g.bound.center.y, g.bound.center.z, g.bound.radius });
        f.ranges.push_back(first);
        f.ranges.push_back(0);
        for (const Pair &c : g.children) {
            if (c.is_group) flatten_rec(*c.group, f);
            else f.items.insert(f.items.end(), { c.item.center.x, c.item.center.y, c.item.center.z, c.item.radius });
        }
        f.ranges[2 * bi + 1] = (int32_t)(f.items.size() / 4) - first;
    }
};

}  // namespace rtrace
