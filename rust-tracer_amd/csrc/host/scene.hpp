// scene.hpp -- host-side mirror of primitive.rs / group.rs / Scene (render.rs:138-167): the Scene is built and
// owned on the host exactly as the reference does; flatten() turns the group tree into the DFS arrays the C ABI
// consumes (items in traversal order + each group's bound and item range, pre-order).
#pragma once
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <vector>

#include "vec.hpp"

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

    // TypedGroup::count  group.rs:93-109 -> (num_groups, num_items)
    void count(size_t &ng, size_t &ni) const
    {
        ng += 1;
        for (const Pair &c : children) {
            if (c.is_group) c.group->count(ng, ni); else ni += 1;
        }
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
