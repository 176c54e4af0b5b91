/*
 * rt_oracle.c -- TEST INFRASTRUCTURE ONLY: C-ABI front-end of the CPU restatement (see rt_oracle.h).
 * Build: gcc -std=c11 -O2 -ffp-contract=off -fno-fast-math -fPIC -shared -pthread (oracle/Makefile).
 */
#define _GNU_SOURCE
#include "rt_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define ORC_CHUNK 64u            /* CHUNK_SIZE render.rs:264 */

/* ---- f32: the reference as shipped (vec.rs:6) ---- */
#define REAL float
#define SFX f32
#define RSQRT(x) sqrtf(x)
#define RINF ((float)INFINITY)
#define REPS FLT_EPSILON          /* f32::EPSILON render.rs:199 */
#include "rt_oracle_impl.h"
#undef REAL
#undef SFX
#undef RSQRT
#undef RINF
#undef REPS

/* ---- f64: RFloat = f64, every hard-coded f32 constant promoted; parity unpinned ---- */
#define REAL double
#define SFX f64
#define RSQRT(x) sqrt(x)
#define RINF ((double)INFINITY)
#define REPS DBL_EPSILON
#include "rt_oracle_impl.h"
#undef REAL
#undef SFX
#undef RSQRT
#undef RINF
#undef REPS

struct orc_scene {
    int prec;
    scene_f32 *s32;
    scene_f64 *s64;
};

static orc_scene *wrap(int prec, scene_f32 *a, scene_f64 *b)
{
    if (!a && !b) return NULL;
    orc_scene *s = (orc_scene *)calloc(1, sizeof *s);
    s->prec = prec; s->s32 = a; s->s64 = b;
    return s;
}

orc_scene *orc_scene_pyramid(int prec, unsigned level, const double origin[3], double radius,
                             const double light_unnormalised[3], const double eye[3])
{
    if (prec == ORC_F32) return wrap(prec, scene_pyramid_f32(level, origin, radius, light_unnormalised, eye), NULL);
    return wrap(prec, NULL, scene_pyramid_f64(level, origin, radius, light_unnormalised, eye));
}

orc_scene *orc_scene_default(int prec, unsigned level)
{
    /* render.rs:147-164: pyramid(8, (0,-1,0), 1.0); light (-1,-3,2).normalized(); eye (0,0,-4) */
    const double o[3] = { 0.0, -1.0, 0.0 }, l[3] = { -1.0, -3.0, 2.0 }, e[3] = { 0.0, 0.0, -4.0 };
    return orc_scene_pyramid(prec, level, o, 1.0, l, e);
}

orc_scene *orc_scene_from_spheres(int prec, const double *spheres4, int n, const double bound4[4],
                                  const double light_unnormalised[3], const double eye[3])
{
    if (prec == ORC_F32) return wrap(prec, scene_from_spheres_f32(spheres4, n, bound4, light_unnormalised, eye), NULL);
    return wrap(prec, NULL, scene_from_spheres_f64(spheres4, n, bound4, light_unnormalised, eye));
}

orc_scene *orc_scene_from_ranges(int prec, const double *items4, int n, const double *bounds4, const int32_t *ranges2, int nb,
                                 const double light_unnormalised[3], const double eye[3])
{
    if (prec == ORC_F32) return wrap(prec, scene_from_ranges_f32(items4, n, bounds4, ranges2, nb, light_unnormalised, eye), NULL);
    return wrap(prec, NULL, scene_from_ranges_f64(items4, n, bounds4, ranges2, nb, light_unnormalised, eye));
}

void orc_scene_free(orc_scene *s)
{
    if (!s) return;
    scene_free_f32(s->s32);
    scene_free_f64(s->s64);
    free(s);
}

int orc_scene_precision(const orc_scene *s) { return s->prec; }

void orc_scene_counts(const orc_scene *s, int *n_groups, int *n_items)
{
    if (s->prec == ORC_F32) { *n_groups = s->s32->n_groups; *n_items = s->s32->n_flat; }
    else { *n_groups = s->s64->n_groups; *n_items = s->s64->n_flat; }
}

int orc_scene_flatten(const orc_scene *s, void *out)
{
    if (s->prec == ORC_F32) { memcpy(out, s->s32->flat, sizeof(sphere_f32) * (size_t)s->s32->n_flat); return s->s32->n_flat; }
    memcpy(out, s->s64->flat, sizeof(sphere_f64) * (size_t)s->s64->n_flat);
    return s->s64->n_flat;
}

int orc_scene_bounds(const orc_scene *s, void *out_real4, int32_t *out_first_count)
{
    int nb = 0, ni = 0;
    if (s->prec == ORC_F32) bounds_rec_f32(s->s32->group, (float *)out_real4, out_first_count, &nb, &ni);
    else bounds_rec_f64(s->s64->group, (double *)out_real4, out_first_count, &nb, &ni);
    return nb;
}

void orc_scene_light_eye(const orc_scene *s, void *light3, void *eye3)
{
    if (s->prec == ORC_F32) { memcpy(light3, &s->s32->directional_light, 12); memcpy(eye3, &s->s32->eye, 12); }
    else { memcpy(light3, &s->s64->directional_light, 24); memcpy(eye3, &s->s64->eye, 24); }
}

void orc_render_region(const orc_scene *s, int mode, unsigned w, unsigned h, unsigned spp,
                       unsigned l, unsigned t, unsigned r, unsigned b, uint8_t *rgba, orc_stats *st)
{
    orc_stats local; memset(&local, 0, sizeof local);
    if (s->prec == ORC_F32) render_region_f32(s->s32, mode, w, h, spp, l, t, r, b, rgba, &local);
    else render_region_f64(s->s64, mode, w, h, spp, l, t, r, b, rgba, &local);
    if (st) {
        st->primary += local.primary; st->hits += local.hits; st->shadow += local.shadow;
        st->occluded += local.occluded; st->sphere_tests += local.sphere_tests; st->bound_tests += local.bound_tests;
    }
}

int orc_render(const orc_scene *s, int mode, unsigned w, unsigned h, unsigned spp, unsigned nthreads,
               uint8_t *frame_rgba, orc_stats *st)
{
    if (s->prec == ORC_F32) return render_f32(s->s32, mode, w, h, spp, nthreads, frame_rgba, st);
    return render_f64(s->s64, mode, w, h, spp, nthreads, frame_rgba, st);
}

/* render.rs:359-407 write_buffer_with_header */
int orc_write_ppm(const char *path, const uint8_t *frame, unsigned w, unsigned h, int rgb)
{
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    fprintf(f, "%s\n%u %u\n255\n", rgb ? "P6" : "P5", w, h);
    size_t n = (size_t)w * h;
    uint8_t *row = (uint8_t *)malloc(n * 3);
    size_t o = 0;
    for (size_t i = 0; i < n; ++i) {
        const uint8_t *b = frame + i * 4;
        if (rgb) { row[o++] = b[0]; row[o++] = b[1]; row[o++] = b[2]; }
        else row[o++] = (uint8_t)(((float)b[0] + (float)b[1] + (float)b[2]) / 3.0f);   /* render.rs:399 */
    }
    size_t wr = fwrite(row, 1, o, f);
    free(row);
    fclose(f);
    return wr == o ? 0 : -1;
}

/* ---- known-answer probes ---- */

double orc_sphere_distance_from_ray(int prec, const double s4[4], const double r6[6])
{
    if (prec == ORC_F32) {
        sphere_f32 s = { { (float)s4[0], (float)s4[1], (float)s4[2] }, (float)s4[3] };
        ray_f32 r = { { (float)r6[0], (float)r6[1], (float)r6[2] }, { (float)r6[3], (float)r6[4], (float)r6[5] } };
        return (double)sphere_distance_from_ray_f32(&s, &r);
    }
    sphere_f64 s = { { s4[0], s4[1], s4[2] }, s4[3] };
    ray_f64 r = { { r6[0], r6[1], r6[2] }, { r6[3], r6[4], r6[5] } };
    return sphere_distance_from_ray_f64(&s, &r);
}

void orc_sphere_intersect(int prec, const double s4[4], const double r6[6], double hit_in, double out[4])
{
    if (prec == ORC_F32) {
        sphere_f32 s = { { (float)s4[0], (float)s4[1], (float)s4[2] }, (float)s4[3] };
        ray_f32 r = { { (float)r6[0], (float)r6[1], (float)r6[2] }, { (float)r6[3], (float)r6[4], (float)r6[5] } };
        hit_f32 h = { (float)hit_in, { 0.f, 0.f, 0.f } };
        sphere_intersect_f32(&s, &h, &r);
        out[0] = h.distance; out[1] = h.pos.x; out[2] = h.pos.y; out[3] = h.pos.z;
        return;
    }
    sphere_f64 s = { { s4[0], s4[1], s4[2] }, s4[3] };
    ray_f64 r = { { r6[0], r6[1], r6[2] }, { r6[3], r6[4], r6[5] } };
    hit_f64 h = { hit_in, { 0., 0., 0. } };
    sphere_intersect_f64(&s, &h, &r);
    out[0] = h.distance; out[1] = h.pos.x; out[2] = h.pos.y; out[3] = h.pos.z;
}

void orc_scene_intersect(const orc_scene *s, int mode, const double r6[6], double hit_in, double out[4])
{
    orc_stats st; memset(&st, 0, sizeof st);
    if (s->prec == ORC_F32) {
        ray_f32 r = { { (float)r6[0], (float)r6[1], (float)r6[2] }, { (float)r6[3], (float)r6[4], (float)r6[5] } };
        hit_f32 h = { (float)hit_in, { 0.f, 0.f, 0.f } };
        scene_intersect_f32(s->s32, mode, 0, &h, &r, &st);
        out[0] = h.distance; out[1] = h.pos.x; out[2] = h.pos.y; out[3] = h.pos.z;
        return;
    }
    ray_f64 r = { { r6[0], r6[1], r6[2] }, { r6[3], r6[4], r6[5] } };
    hit_f64 h = { hit_in, { 0., 0., 0. } };
    scene_intersect_f64(s->s64, mode, 0, &h, &r, &st);
    out[0] = h.distance; out[1] = h.pos.x; out[2] = h.pos.y; out[3] = h.pos.z;
}

void orc_vec_normalized(int prec, const double v3[3], double out3[3], double *len_in, double *len_out)
{
    if (prec == ORC_F32) {
        vec_f32 v = { (float)v3[0], (float)v3[1], (float)v3[2] };
        vec_f32 n = vnormalized_f32(v);
        out3[0] = n.x; out3[1] = n.y; out3[2] = n.z;
        *len_in = vlen_f32(v); *len_out = vlen_f32(n);
        return;
    }
    vec_f64 v = { v3[0], v3[1], v3[2] };
    vec_f64 n = vnormalized_f64(v);
    out3[0] = n.x; out3[1] = n.y; out3[2] = n.z;
    *len_in = vlen_f64(v); *len_out = vlen_f64(n);
}
