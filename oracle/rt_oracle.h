/*
 * rt_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement ("port") of rust-tracer's hot path
 *   Renderer::render_region -> Renderer::raytrace -> SphericalGroup::intersect -> Sphere::distance_from_ray
 *   (/root/reference/src/rust/render.rs:171-255, group.rs:68-84, primitive.rs:53-85)
 * plus the pieces either side that define its inputs and outputs (pyramid builder group.rs:27-66,
 * Scene::default render.rs:144-166, 64x64 bucket scheduler render.rs:260-310, PPM writer render.rs:359-407).
 *
 * PINNING: f32 is pinned against the reference's own output image src/img/rtrace-output.png
 * (1024x768, spp 4, `make image`; tests/golden/make_image_1024x768_spp4.*) with 0 differing pixels, and
 * against every known answer of the reference's in-module unit tests (tests/test_oracle_known_answers.py).
 * f64 ("type-alias swap") is PARITY UNPINNED: the reference holds no f64 vector.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.  The product
 * (librtrace_hip.so) never links, loads or calls it.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_F32 = 0, ORC_F64 = 1 };
enum { ORC_MODE_HIERARCHY = 0,   /* the reference's bounding-sphere tree traversal (group.rs:72-83) */
       ORC_MODE_FLAT = 1,        /* every item in DFS order, no culling (the GPU flat-scan semantics) */
       ORC_MODE_ANYHIT_EXIT = 2  /* OR-able flag: shadow rays stop at their first hit (same pixels, fewer tests) */
     };

typedef struct {
    uint64_t primary;        /* primary samples traced */
    uint64_t hits;           /* primary rays that hit something */
    uint64_t shadow;         /* shadow rays cast */
    uint64_t occluded;       /* shadow rays that hit something */
    uint64_t sphere_tests;   /* Sphere::intersect calls on items */
    uint64_t bound_tests;    /* bound.distance_from_ray calls */
} orc_stats;

typedef struct orc_scene orc_scene;

/* Scene::default() (render.rs:144-166) with a variable pyramid level (8 in the reference). */
orc_scene *orc_scene_default(int prec, unsigned level);
/* SphericalGroup::pyramid(level, origin, radius) + light (normalised here, in REAL) + eye. */
orc_scene *orc_scene_pyramid(int prec, unsigned level, const double origin[3], double radius,
                             const double light_unnormalised[3], const double eye[3]);
/* One group {bound, children = the n spheres as Items, in order}; what group::tests::setup_group builds
 * (group.rs:118-151) and what BASELINE config 1 ("3 spheres, 1 light") needs. */
orc_scene *orc_scene_from_spheres(int prec, const double *spheres4, int n, const double bound4[4],
                                  const double light_unnormalised[3], const double eye[3]);
/* Arbitrary nesting: DFS items + pre-order group ranges {first, count} (the description rt_scene_create takes);
 * ranges[0] must be the root {0, n}. */
orc_scene *orc_scene_from_ranges(int prec, const double *items4, int n, const double *bounds4, const int32_t *ranges2, int nb,
                                 const double light_unnormalised[3], const double eye[3]);
void orc_scene_free(orc_scene *s);

void orc_scene_counts(const orc_scene *s, int *n_groups, int *n_items);     /* TypedGroup::count group.rs:93-109 */
int orc_scene_flatten(const orc_scene *s, void *out_real4);                 /* REAL[4*n]: cx,cy,cz,r in DFS order */
int orc_scene_bounds(const orc_scene *s, void *out_real4, int32_t *out_first_count);
void orc_scene_light_eye(const orc_scene *s, void *light_real3, void *eye_real3);
int orc_scene_precision(const orc_scene *s);

/* Renderer::render_region (render.rs:218-255); region as ImageRegion{l,t,r,b} with t > b; rgba is the
 * tile-local RGBABuffer (4 B/px, row 0 = y == b).  stats are ADDED to *st (may be NULL). */
void orc_render_region(const orc_scene *s, int mode, unsigned w, unsigned h, unsigned spp,
                       unsigned l, unsigned t, unsigned r, unsigned b, uint8_t *rgba, orc_stats *st);

/* Renderer::render (render.rs:260-310): 64x64 buckets row-major, nthreads pool workers, tiles blitted into
 * the full frame (set_pixels_from_buffer render.rs:112-126).  Unlike the reference, edge buckets are clipped
 * instead of asserting w%64==0 && h%64==0.  Returns the number of buckets. */
int orc_render(const orc_scene *s, int mode, unsigned w, unsigned h, unsigned spp, unsigned nthreads,
               uint8_t *frame_rgba, orc_stats *st);

/* PPM writer (render.rs:359-407): "P6\n{w} {h}\n255\n" + RGB, or P5 with ((r+g+b) as f32 / 3.0) as u8. */
int orc_write_ppm(const char *path, const uint8_t *frame_rgba, unsigned w, unsigned h, int rgb);

/* Known-answer probes for the reference's unit tests (values travel as double; computed in REAL). */
double orc_sphere_distance_from_ray(int prec, const double sphere4[4], const double ray6[6]);
void orc_sphere_intersect(int prec, const double sphere4[4], const double ray6[6], double hit_distance_in,
                          double out_distance_pos4[4]);
void orc_scene_intersect(const orc_scene *s, int mode, const double ray6[6], double hit_distance_in,
                         double out_distance_pos4[4]);
void orc_vec_normalized(int prec, const double v3[3], double out3[3], double *len_in, double *len_out);

#ifdef __cplusplus
}
#endif
#endif
