/*
 * rt_oracle_impl.h -- TEST INFRASTRUCTURE ONLY (parity oracle + reported CPU baseline).
 *
 * Body of the CPU restatement of rust-tracer's per-pixel ray-sphere path, written once
 * and instantiated for two real types by rt_oracle.c:
 *      REAL = float   SFX = f32   (the reference as shipped: `pub type RFloat = f32`, vec.rs:6)
 *      REAL = double  SFX = f64   (the "type-alias swap" of BASELINE.json config 3; parity UNPINNED,
 *                                  the reference holds no f64 vectors -- SURVEY.md H6)
 *
 * Every function cites the reference file:line it restates.  Arithmetic rules kept from the
 * reference: every + - * / individually rounded (build with -ffp-contract=off), operations in
 * source order, IEEE sqrt and true division (`recip()` = 1.0/x).
 *
 * Nothing outside tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this.
 */

#ifndef REAL
#error "include from rt_oracle.c with REAL / SFX / RSQRT / RINF defined"
#endif

#define CAT_(a, b) a##_##b
#define CAT(a, b) CAT_(a, b)
#define N(name) CAT(name, SFX)

typedef struct { REAL x, y, z; } N(vec);

/* vec.rs:15-53  Add / Sub / Mul, component-wise */
static inline N(vec) N(vadd)(N(vec) a, N(vec) b) { N(vec) r = { a.x + b.x, a.y + b.y, a.z + b.z }; return r; }
static inline N(vec) N(vsub)(N(vec) a, N(vec) b) { N(vec) r = { a.x - b.x, a.y - b.y, a.z - b.z }; return r; }
static inline N(vec) N(vmul)(N(vec) a, N(vec) b) { N(vec) r = { a.x * b.x, a.y * b.y, a.z * b.z }; return r; }
/* vec.rs:57-72  mulfed (copy) and mulf (in place) are the same arithmetic */
static inline N(vec) N(vmulf)(N(vec) a, REAL m) { N(vec) r = { a.x * m, a.y * m, a.z * m }; return r; }
/* vec.rs:77-79  dot = x*x' + y*y' + z*z', left to right */
static inline REAL N(vdot)(N(vec) a, N(vec) b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
/* vec.rs:82-84 */
static inline REAL N(vlen)(N(vec) a) { return RSQRT(N(vdot)(a, a)); }
/* vec.rs:87-95  normalize / normalized: multiply by len.recip() (a true division 1/len) */
static inline N(vec) N(vnormalized)(N(vec) a) { REAL l = N(vlen)(a); return N(vmulf)(a, (REAL)1.0 / l); }

/* primitive.rs:9-13, 15-36, 38-51 */
typedef struct { N(vec) pos, dir; } N(ray);
typedef struct { REAL distance; N(vec) pos; } N(hit);
typedef struct { N(vec) center; REAL radius; } N(sphere);

/* primitive.rs:55-72  Sphere::distance_from_ray */
static inline REAL N(sphere_distance_from_ray)(const N(sphere) *s, const N(ray) *r)
{
    N(vec) v = N(vsub)(s->center, r->pos);
    REAL b = N(vdot)(v, r->dir);
    REAL disc = b * b - N(vdot)(v, v) + s->radius * s->radius;
    if (disc < (REAL)0.0) return RINF;
    REAL d = RSQRT(disc);
    REAL t2 = b + d;
    if (t2 < (REAL)0.0) return RINF;
    REAL t1 = b - d;
    return t1 > (REAL)0.0 ? t1 : t2;
}

/* primitive.rs:77-84  Sphere::intersect: strict `<` nearest update, hit.pos becomes the unit normal */
static inline void N(sphere_intersect)(const N(sphere) *s, N(hit) *h, const N(ray) *r)
{
    REAL distance = N(sphere_distance_from_ray)(s, r);
    if (distance >= h->distance) return;
    h->distance = distance;
    h->pos = N(vnormalized)(N(vadd)(r->pos, N(vsub)(N(vmulf)(r->dir, distance), s->center)));
}

/* group.rs:7-20  Pair<Item, Group> / TypedGroup{bound, children} */
typedef struct N(group) N(group);
typedef struct {
    int is_group;
    N(sphere) item;   /* valid when !is_group */
    N(group) *group;  /* valid when is_group */
} N(pair);
struct N(group) {
    N(sphere) bound;
    N(pair) *children;
    int n_children, cap_children;
};

static void N(group_push)(N(group) *g, N(pair) p)
{
    if (g->n_children == g->cap_children) {
        g->cap_children = g->cap_children ? g->cap_children * 2 : 5;
        g->children = (N(pair) *)realloc(g->children, sizeof(N(pair)) * (size_t)g->cap_children);
    }
    g->children[g->n_children++] = p;
}

static N(group) *N(group_new)(void)
{
    N(group) *g = (N(group) *)calloc(1, sizeof(N(group)));
    /* `#[derive(Default)]` group.rs:16 with Sphere::default() primitive.rs:44-50: centre 0, radius 1 */
    g->bound.radius = (REAL)1.0;
    return g;
}

static void N(group_free)(N(group) *g)
{
    if (!g) return;
    for (int i = 0; i < g->n_children; ++i)
        if (g->children[i].is_group) N(group_free)(g->children[i].group);
    free(g->children);
    free(g);
}

/* group.rs:28-56  SphericalGroup::pyramid_recursive */
static N(pair) N(pyramid_recursive)(unsigned level, N(vec) p, REAL r)
{
    N(pair) out;
    memset(&out, 0, sizeof out);
    N(sphere) s = { p, r };
    if (level == 1) { out.is_group = 0; out.item = s; return out; }

    N(group) *g = N(group_new)();
    N(pair) own; memset(&own, 0, sizeof own); own.item = s;
    N(group_push)(g, own);
    g->bound.center = p;
    g->bound.radius = (REAL)3.0 * r;

    REAL rn = (REAL)3.0 * r / RSQRT((REAL)12.0);          /* group.rs:43 */
    static const int sgn[2] = { -1, 1 };
    for (int iz = 0; iz < 2; ++iz) {                       /* dz outer  group.rs:44 */
        for (int ix = 0; ix < 2; ++ix) {                   /* dx inner  group.rs:45 */
            N(vec) off = { (REAL)sgn[ix] * rn, rn, (REAL)sgn[iz] * rn };
            N(vec) np = N(vadd)(p, off);
            N(group_push)(g, N(pyramid_recursive)(level - 1, np, r * (REAL)0.5));
        }
    }
    out.is_group = 1; out.group = g;
    return out;
}

/* group.rs:72-83  TypedGroup::intersect -- bound cull with `>=`, then children in insertion order.
 * anyhit != 0 (shadow rays only, ORC_MODE_ANYHIT_EXIT): stop at the first hit.  The reference keeps walking,
 * but render.rs:208 only asks has_missed(), so pixels are identical; only the test counters differ -- this is
 * the traversal the GPU SKIP kernel executes, and its counters must equal these. */
static int N(group_intersect)(const N(group) *g, N(hit) *h, const N(ray) *r, orc_stats *st, int anyhit)
{
    st->bound_tests++;
    if (N(sphere_distance_from_ray)(&g->bound, r) >= h->distance) return 0;
    for (int i = 0; i < g->n_children; ++i) {
        const N(pair) *c = &g->children[i];
        if (c->is_group) { if (N(group_intersect)(c->group, h, r, st, anyhit)) return 1; }
        else {
            st->sphere_tests++;
            N(sphere_intersect)(&c->item, h, r);
            if (anyhit && !(h->distance == RINF)) return 1;
        }
    }
    return 0;
}

/* render.rs:138-142  Scene, plus the DFS-flattened item list the GPU boundary consumes */
typedef struct {
    N(group) *group;
    N(vec) directional_light, eye;
    N(sphere) *flat;      /* items in traversal (DFS, insertion) order */
    int n_flat;
    int n_groups;
} N(scene);

static void N(flatten_rec)(const N(group) *g, N(sphere) *out, int *n, int *ng)
{
    (*ng)++;
    for (int i = 0; i < g->n_children; ++i) {
        const N(pair) *c = &g->children[i];
        if (c->is_group) N(flatten_rec)(c->group, out, n, ng);
        else { if (out) out[*n] = c->item; (*n)++; }
    }
}

static void N(scene_finish)(N(scene) *s)
{
    int n = 0, ng = 0;
    N(flatten_rec)(s->group, NULL, &n, &ng);
    s->flat = (N(sphere) *)malloc(sizeof(N(sphere)) * (size_t)(n ? n : 1));
    n = 0; ng = 0;
    N(flatten_rec)(s->group, s->flat, &n, &ng);
    s->n_flat = n; s->n_groups = ng;
}

/* Bounds in DFS pre-order: for every group its bound sphere and the [first, first+count) range of
 * flattened item indices its subtree covers (a subtree is contiguous in DFS order).  Outer groups
 * come before the groups nested inside them. */
static void N(bounds_rec)(const N(group) *g, REAL *ob4, int32_t *or2, int *nb, int *ni)
{
    int my = (*nb)++;
    int first = *ni;
    for (int i = 0; i < g->n_children; ++i) {
        const N(pair) *c = &g->children[i];
        if (c->is_group) N(bounds_rec)(c->group, ob4, or2, nb, ni);
        else (*ni)++;
    }
    if (ob4) {
        REAL *o = ob4 + 4 * (size_t)my;
        o[0] = g->bound.center.x; o[1] = g->bound.center.y; o[2] = g->bound.center.z; o[3] = g->bound.radius;
    }
    if (or2) { or2[2 * my] = first; or2[2 * my + 1] = *ni - first; }
}

/* flat-scan traversal (the GPU kernel's semantics): every item in DFS order, same per-item rule */
static void N(flat_intersect)(const N(scene) *s, N(hit) *h, const N(ray) *r, orc_stats *st)
{
    for (int i = 0; i < s->n_flat; ++i) N(sphere_intersect)(&s->flat[i], h, r);
    st->sphere_tests += (uint64_t)s->n_flat;
}

static inline void N(scene_intersect)(const N(scene) *s, int mode, int is_shadow, N(hit) *h, const N(ray) *r, orc_stats *st)
{
    if ((mode & 1) == ORC_MODE_FLAT) N(flat_intersect)(s, h, r, st);
    else N(group_intersect)(s->group, h, r, st, is_shadow && (mode & ORC_MODE_ANYHIT_EXIT));
}

/* render.rs:171-215  Renderer::raytrace */
static REAL N(raytrace)(const N(scene) *s, int mode, const N(ray) *r, N(vec) *c, orc_stats *st)
{
    const N(vec) OBJECT = { (REAL)0xae / (REAL)255.0, (REAL)0x31 / (REAL)255.0, (REAL)0x31 / (REAL)255.0 };
    const N(vec) BACKGROUND = { (REAL)0x22 / (REAL)255.0, (REAL)0x0a / (REAL)255.0, (REAL)0x0a / (REAL)255.0 };
    const N(vec) AMBIENT_OFFSET = { BACKGROUND.x * (REAL)0.8, BACKGROUND.y * (REAL)0.8, BACKGROUND.z * (REAL)0.8 };

    st->primary++;
    N(hit) h; h.distance = RINF; h.pos.x = h.pos.y = h.pos.z = (REAL)0.0;    /* Hit::missed() primitive.rs:22-27 */
    N(scene_intersect)(s, mode, 0, &h, r, st);
    if (h.distance == RINF) {                                                  /* has_missed primitive.rs:29-31 */
        *c = N(vadd)(*c, BACKGROUND);
        return (REAL)0.0;
    }
    st->hits++;
    REAL g = N(vdot)(h.pos, s->directional_light);
    if (g >= (REAL)0.0) {
        *c = N(vadd)(*c, AMBIENT_OFFSET);
        return (REAL)0.0;
    }
    /* render.rs:199   p = (r.pos + r.dir*d) + n*(d*sqrt(EPSILON)) */
    N(vec) nscaled = N(vmulf)(h.pos, h.distance * RSQRT(REPS));
    N(vec) p = N(vadd)(N(vadd)(r->pos, N(vmulf)(r->dir, h.distance)), nscaled);

    h.distance = RINF;                                                         /* set_missed render.rs:202 */
    N(ray) sr; sr.pos = p; sr.dir = N(vmulf)(s->directional_light, (REAL)-1.0);
    st->shadow++;
    N(scene_intersect)(s, mode, 1, &h, &sr, st);
    if (h.distance == RINF) {
        *c = N(vadd)(N(vadd)(*c, N(vmulf)(OBJECT, -g)), AMBIENT_OFFSET);       /* render.rs:209 */
        return (REAL)1.0;
    } else {
        st->occluded++;
        *c = N(vadd)(N(vadd)(*c, BACKGROUND), N(vmulf)(AMBIENT_OFFSET, -g));   /* render.rs:212 */
        return (REAL)0.0;
    }
}

/* render.rs:96-103  the `scale` closure of set_pixel_from_vector */
static inline uint8_t N(scale_u8)(REAL v)
{
    REAL r = (REAL)0.5 + (REAL)255.0 * v;
    if (r > (REAL)255.0) return 255;
    if (!(r > (REAL)0.0)) return 0;       /* Rust `as u8` saturates; NaN -> 0 */
    return (uint8_t)r;                    /* truncation toward zero */
}

/* render.rs:218-255  Renderer::render_region.  buf: tile-local RGBA, row 0 = y == b (render.rs:69-71) */
static void N(render_region)(const N(scene) *scene, int mode, unsigned w, unsigned h, unsigned spp,
                             unsigned l, unsigned t, unsigned rr, unsigned b, uint8_t *buf, orc_stats *st)
{
    REAL ssf = (REAL)spp;
    REAL total_recip = (REAL)1.0 / (ssf * ssf);
    REAL width = (REAL)w, height = (REAL)h;
    unsigned rw = rr - l;
    N(ray) ray; ray.pos = scene->eye; ray.dir.x = ray.dir.y = ray.dir.z = (REAL)0.0;

    for (unsigned y = b; y < t; ++y) {
        for (unsigned x = l; x < rr; ++x) {
            N(vec) g = { (REAL)0.0, (REAL)0.0, (REAL)0.0 };
            REAL alpha = (REAL)0.0;
            for (unsigned ssx = 0; ssx < spp; ++ssx) {          /* ssx OUTER render.rs:236 */
                for (unsigned ssy = 0; ssy < spp; ++ssy) {
                    REAL xres = (REAL)x + (REAL)ssx / ssf;
                    REAL yres = (REAL)y + (REAL)ssy / ssf;
                    ray.dir.x = xres - width / (REAL)2.0;
                    ray.dir.y = (height - yres) - height / (REAL)2.0;
                    ray.dir.z = width;
                    ray.dir = N(vnormalized)(ray.dir);
                    alpha += N(raytrace)(scene, mode, &ray, &g, st);
                }
            }
            g = N(vmulf)(g, total_recip);
            alpha *= total_recip;
            uint8_t *px = buf + ((size_t)(y - b) * rw + (x - l)) * 4;
            px[0] = N(scale_u8)(g.x); px[1] = N(scale_u8)(g.y); px[2] = N(scale_u8)(g.z); px[3] = N(scale_u8)(alpha);
        }
    }
}

/* ---- scene constructors ------------------------------------------------------------------ */

/* render.rs:144-166  Scene::default (level 8 there) / group.rs:58-65 SphericalGroup::pyramid */
static N(scene) *N(scene_pyramid)(unsigned level, const double o[3], double radius, const double lu[3], const double e[3])
{
    if (level <= 1) return NULL;                      /* assert!(level > 1) group.rs:59 */
    N(scene) *s = (N(scene) *)calloc(1, sizeof(N(scene)));
    N(vec) origin = { (REAL)o[0], (REAL)o[1], (REAL)o[2] };
    N(pair) root = N(pyramid_recursive)(level, origin, (REAL)radius);
    s->group = root.group;
    N(vec) l = { (REAL)lu[0], (REAL)lu[1], (REAL)lu[2] };
    s->directional_light = N(vnormalized)(l);         /* render.rs:154-159 */
    s->eye.x = (REAL)e[0]; s->eye.y = (REAL)e[1]; s->eye.z = (REAL)e[2];
    N(scene_finish)(s);
    return s;
}

static N(scene) *N(scene_from_spheres)(const double *sp4, int n, const double b4[4], const double lu[3], const double e[3])
{
    N(scene) *s = (N(scene) *)calloc(1, sizeof(N(scene)));
    s->group = N(group_new)();
    s->group->bound.center.x = (REAL)b4[0]; s->group->bound.center.y = (REAL)b4[1];
    s->group->bound.center.z = (REAL)b4[2]; s->group->bound.radius = (REAL)b4[3];
    for (int i = 0; i < n; ++i) {
        N(pair) p; memset(&p, 0, sizeof p);
        p.item.center.x = (REAL)sp4[4 * i]; p.item.center.y = (REAL)sp4[4 * i + 1];
        p.item.center.z = (REAL)sp4[4 * i + 2]; p.item.radius = (REAL)sp4[4 * i + 3];
        N(group_push)(s->group, p);
    }
    N(vec) l = { (REAL)lu[0], (REAL)lu[1], (REAL)lu[2] };
    s->directional_light = N(vnormalized)(l);
    s->eye.x = (REAL)e[0]; s->eye.y = (REAL)e[1]; s->eye.z = (REAL)e[2];
    N(scene_finish)(s);
    return s;
}

/* General tree from DFS items + pre-order group ranges (the same description the C ABI takes): group k bounds
 * items [first, first+count).  Returns NULL if the ranges do not nest. */
static N(group) *N(tree_rec)(const double *it4, const double *b4, const int32_t *rg, int nb, int *bi, int *pos)
{
    int k = (*bi)++;
    int first = rg[2 * k], end = first + rg[2 * k + 1];
    if (first != *pos) return NULL;
    N(group) *g = N(group_new)();
    g->bound.center.x = (REAL)b4[4 * k]; g->bound.center.y = (REAL)b4[4 * k + 1];
    g->bound.center.z = (REAL)b4[4 * k + 2]; g->bound.radius = (REAL)b4[4 * k + 3];
    while (*pos < end) {
        N(pair) p; memset(&p, 0, sizeof p);
        if (*bi < nb && rg[2 * (*bi)] == *pos) {
            if (rg[2 * (*bi)] + rg[2 * (*bi) + 1] > end) { N(group_free)(g); return NULL; }
            p.is_group = 1;
            p.group = N(tree_rec)(it4, b4, rg, nb, bi, pos);
            if (!p.group) { N(group_free)(g); return NULL; }
        } else {
            p.item.center.x = (REAL)it4[4 * (*pos)]; p.item.center.y = (REAL)it4[4 * (*pos) + 1];
            p.item.center.z = (REAL)it4[4 * (*pos) + 2]; p.item.radius = (REAL)it4[4 * (*pos) + 3];
            (*pos)++;
        }
        N(group_push)(g, p);
    }
    return g;
}

static N(scene) *N(scene_from_ranges)(const double *it4, int n, const double *b4, const int32_t *rg, int nb,
                                      const double lu[3], const double e[3])
{
    if (nb < 1 || rg[0] != 0 || rg[1] != n) return NULL;     /* the first range is the root group over all items */
    int bi = 0, pos = 0;
    N(group) *root = N(tree_rec)(it4, b4, rg, nb, &bi, &pos);
    if (!root || bi != nb || pos != n) { N(group_free)(root); return NULL; }
    N(scene) *s = (N(scene) *)calloc(1, sizeof(N(scene)));
    s->group = root;
    N(vec) l = { (REAL)lu[0], (REAL)lu[1], (REAL)lu[2] };
    s->directional_light = N(vnormalized)(l);
    s->eye.x = (REAL)e[0]; s->eye.y = (REAL)e[1]; s->eye.z = (REAL)e[2];
    N(scene_finish)(s);
    return s;
}

static void N(scene_free)(N(scene) *s)
{
    if (!s) return;
    N(group_free)(s->group);
    free(s->flat);
    free(s);
}

/* ---- render.rs:260-310  Renderer::render: bucket scheduler on a worker pool --------------------- */

typedef struct {
    const N(scene) *scene;
    int mode;
    unsigned w, h, spp, tiles_x, n_tiles;
    uint8_t *frame;
    atomic_uint next;
    pthread_mutex_t lock;
    orc_stats total;
} N(job);

static void *N(worker)(void *arg)
{
    N(job) *j = (N(job) *)arg;
    orc_stats st; memset(&st, 0, sizeof st);
    uint8_t *tile = (uint8_t *)malloc(ORC_CHUNK * ORC_CHUNK * 4);        /* RGBABuffer::new render.rs:80-85 */
    for (;;) {
        unsigned id = atomic_fetch_add(&j->next, 1u);
        if (id >= j->n_tiles) break;
        unsigned x = (id % j->tiles_x) * ORC_CHUNK, y = (id / j->tiles_x) * ORC_CHUNK;
        unsigned r = x + ORC_CHUNK < j->w ? x + ORC_CHUNK : j->w;         /* clipped edge bucket (H5) */
        unsigned t = y + ORC_CHUNK < j->h ? y + ORC_CHUNK : j->h;
        N(render_region)(j->scene, j->mode, j->w, j->h, j->spp, x, t, r, y, tile, &st);
        /* set_pixels_from_buffer render.rs:112-126: row-wise blit; regions are disjoint, no lock needed */
        size_t rowb = (size_t)(r - x) * 4;
        for (unsigned yy = y; yy < t; ++yy)
            memcpy(j->frame + ((size_t)yy * j->w + x) * 4, tile + (size_t)(yy - y) * rowb, rowb);
    }
    free(tile);
    pthread_mutex_lock(&j->lock);
    j->total.primary += st.primary; j->total.hits += st.hits; j->total.shadow += st.shadow;
    j->total.occluded += st.occluded; j->total.sphere_tests += st.sphere_tests; j->total.bound_tests += st.bound_tests;
    pthread_mutex_unlock(&j->lock);
    return NULL;
}

static int N(render)(const N(scene) *s, int mode, unsigned w, unsigned h, unsigned spp, unsigned nthreads,
                     uint8_t *frame, orc_stats *st)
{
    N(job) j; memset(&j, 0, sizeof j);
    j.scene = s; j.mode = mode; j.w = w; j.h = h; j.spp = spp; j.frame = frame;
    j.tiles_x = (w + ORC_CHUNK - 1) / ORC_CHUNK;
    j.n_tiles = j.tiles_x * ((h + ORC_CHUNK - 1) / ORC_CHUNK);
    atomic_init(&j.next, 0u);
    pthread_mutex_init(&j.lock, NULL);
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    pthread_t th[256];
    for (unsigned i = 0; i < nthreads; ++i) pthread_create(&th[i], NULL, N(worker), &j);
    for (unsigned i = 0; i < nthreads; ++i) pthread_join(th[i], NULL);
    pthread_mutex_destroy(&j.lock);
    if (st) {
        st->primary += j.total.primary; st->hits += j.total.hits; st->shadow += j.total.shadow;
        st->occluded += j.total.occluded; st->sphere_tests += j.total.sphere_tests; st->bound_tests += j.total.bound_tests;
    }
    return (int)j.n_tiles;
}

#undef CAT_
#undef CAT
#undef N
