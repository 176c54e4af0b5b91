// host_tests.cpp -- the reference's in-module unit tests that concern host-side types, restated against the C++
// host mirror (vec.rs:105-170, primitive.rs:175-179, group.rs:172-184, render.rs:466-499).  Exit status 0 = pass.
// basic_rendering needs a GPU (there is no CPU fallback); without one it is reported as SKIP.
#include <cstdio>
#include <cstdlib>

#include "render.hpp"

using namespace rtrace;

static int failures = 0;
#define CHECK(cond)                                                              \
    do {                                                                         \
        if (!(cond)) { fprintf(stderr, "FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); ++failures; } \
    } while (0)

static void vec_basics()                                          // vec.rs:105-148
{
    Vector v32{ 5.0, 4.0, 0.0 };
    CHECK(v32.x == 5.0);
    CHECK(v32 == v32);
    CHECK(!(v32 != v32));
    { Vector copy = v32; copy.x = 10.0; CHECK(v32.x == 5.0 && copy.x == 10.0); }
    Vector v = v32 + v32;
    CHECK(v.x == v32.x + v32.x && v.y == v32.y + v32.y && v.z == v32.z + v32.z);
    v = v32 - v32;
    CHECK(v.x == 0.0);
    v = v32 * v32;
    CHECK(v.x == v32.x * v32.x);
    v = v32.mulfed(3.0);
    CHECK(v.x == v32.x * RFloat(3.0));
    v.mulf(2.0);
    CHECK(v.x == v32.x * RFloat(3.0) * RFloat(2.0));
}

static void vec_default_and_normalize()                           // vec.rs:150-170
{
    Vector a, b{};
    CHECK(a == b);
    Vector v{ 2.0, 0.0, 0.0 };
    CHECK(v.len() == 2.0);
    CHECK(v.normalized().len() == 1.0);
    CHECK(v.normalize().len() == 1.0);
}

static void sphere_default()                                      // primitive.rs:175-179
{
    Sphere s;
    CHECK(s.radius != 0.0);
}

static void pyramid()                                             // group.rs:172-184
{
    auto g = SphericalGroup::pyramid(8, Vector{ 1.0, -1.0, 0.0 }, 1.0);
    CHECK(g->children.size() == 5);
    size_t ng = 0, ni = 0;
    g->count(ng, ni);
    CHECK(ng == 5461 && ni == 21845);
    bool threw = false;
    try { SphericalGroup::pyramid(1, Vector{}, 1.0); } catch (const std::invalid_argument &) { threw = true; }
    CHECK(threw);                                                 // assert!(level > 1)  group.rs:59
    const FlatScene f = Scene::default_scene().flatten();
    CHECK(f.items.size() == 4 * 21845 && f.bounds.size() == 4 * 5461 && f.ranges.size() == 2 * 5461);
    CHECK(f.ranges[0] == 0 && f.ranges[1] == 21845 && f.ranges[2] == 1 && f.ranges[3] == 5461);
}

static void image_region()                                        // render.rs:483-499
{
    ImageRegion r{ 2, 18, 34, 2 };
    CHECK(r.width() == 32);
    CHECK(r.height() == 16);
    CHECK(r.area() == 16 * 32);
    CHECK(r.contains(r));
    ImageRegion l = r;
    l.l = 1;
    CHECK(l.contains(r));
    CHECK(!r.contains(l));
}

static void bucket_list()
{
    auto b = Renderer::buckets(RenderOptions{ 64, 128, 2 });
    CHECK(b.size() == 2 && b[1] == (ImageRegion{ 0, 128, 64, 64 }));
    CHECK(Renderer::buckets(RenderOptions{ 800, 600, 1 }).size() == 130);
    CHECK(Renderer::buckets(RenderOptions{ 1920, 1080, 1 }).size() == 510);
}

struct DummyWriter : RGBABufferWriter {                           // render.rs:448-461
    bool begin_called = false;
    size_t write_count = 0;
    void begin(uint16_t, uint16_t) override { begin_called = true; }
    void write_rgba_buffer(const RGBABuffer &) override { write_count += 1; }
};

static void basic_rendering()                                     // render.rs:466-481
{
    int n = 0;
    if (rt_device_count(&n) != RT_OK || n < 1) { printf("SKIP basic_rendering (no GPU; the backend has no CPU fallback)\n"); return; }
    const Scene s = Scene::default_scene();
    Backend be;
    be.devices.push_back(std::make_shared<DeviceScene>(s, 0));
    ThreadPool pool(1);
    DummyWriter dw;
    Renderer::render(RenderOptions{ 64, 128, 2 }, be, dw, pool);
    CHECK(dw.begin_called);
    CHECK(dw.write_count == 2);
}

// `host_tests --hierarchy <scene file>`: prints the flattened arrays of Scene::from_file's automatically built hierarchy as
// "<n items> <n groups> <crc32 of the item bytes> <crc32 of the bound bytes> <crc32 of the range bytes>" -- the Python mirror's
// build_hierarchy must give the same five numbers (tests/test_host_and_abi.py).
static uint32_t crc32_of(const void *data, size_t n)
{
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; ++i) { uint32_t c = i; for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1; table[i] = c; }
        init = true;
    }
    uint32_t c = 0xFFFFFFFFu;
    const uint8_t *p = static_cast<const uint8_t *>(data);
    for (size_t i = 0; i < n; ++i) c = table[(c ^ p[i]) & 0xFFu] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

int main(int argc, char **argv)
{
    if (argc == 3 && std::string(argv[1]) == "--hierarchy") {
        try {
            const FlatScene f = Scene::from_file(argv[2]).flatten();
            printf("%zu %zu %u %u %u\n", f.items.size() / 4, f.ranges.size() / 2, crc32_of(f.items.data(), f.items.size() * sizeof(RFloat)),
                   crc32_of(f.bounds.data(), f.bounds.size() * sizeof(RFloat)), crc32_of(f.ranges.data(), f.ranges.size() * sizeof(int32_t)));
            return 0;
        } catch (const std::exception &e) { fprintf(stderr, "%s\n", e.what()); return 2; }
    }
    vec_basics();
    vec_default_and_normalize();
    sphere_default();
    pyramid();
    image_region();
    bucket_list();
    basic_rendering();
    if (failures) { fprintf(stderr, "%d check(s) failed\n", failures); return 1; }
    printf("host_tests ok\n");
    return 0;
}
