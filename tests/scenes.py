"""Random scene generators shared by the tests and tools/soak.py (numpy only: importing this module does not load the oracle)."""
import numpy as np


def random_nested_scene(seed, depth=3, fan=3, leaf_items=3, concentric=False):
    """A random laminar tree (items and sub-groups interleaved in random order) with tight-ish random bounds:
    some bounds do NOT enclose their subtree, so culling really changes results and order matters.
    concentric: every group's FIRST child is an item at the centre of the group's bound (like the reference's pyramid,
    group.rs:37-41) -- the scene shape for which the f32 traversal loops fuse the bound's step with that item's."""
    rng = np.random.default_rng(seed)
    items, bounds, ranges = [], [], []

    def rec(d, centre, scale):
        bi = len(bounds)
        bounds.append(None); ranges.append(None)
        first = len(items)
        if concentric:
            items.append((centre[0], centre[1], centre[2], float(rng.uniform(0.1, 0.5) * scale)))
        kids = ["item"] * leaf_items + (["group"] * fan if d > 0 else [])
        rng.shuffle(kids)
        for k in kids:
            c = centre + rng.uniform(-scale, scale, 3)
            if k == "item":
                items.append((c[0], c[1], c[2], float(rng.uniform(0.08, 0.35) * scale)))
            else:
                rec(d - 1, c, scale * 0.55)
        bounds[bi] = (centre[0], centre[1], centre[2], float(scale * rng.uniform(1.2, 2.6)))
        ranges[bi] = (first, len(items) - first)

    rec(depth, np.array([0.0, 0.0, 0.0]), 1.2)
    f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
    return f32(items), f32(bounds), np.asarray(ranges, dtype=np.int32)


def hundred_thousand_spheres(seed=5, n=100000):
    """BASELINE config 5's "100k spheres" as an arbitrary list: n spheres in a box in front of the default eye,
    f32-representable values.  The hierarchy is built by the host (scene.py build_hierarchy / csrc/host/scene.hpp)."""
    rng = np.random.default_rng(seed)
    sp = np.concatenate([rng.uniform([-3, -2, 0], [3, 2, 6], (n, 3)), rng.uniform(0.01, 0.03, (n, 1))], axis=1)
    return sp.astype(np.float32).astype(np.float64)
