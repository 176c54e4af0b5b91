// tools/wave_sim.cpp -- design-time estimator, not part of the product and not a parity oracle.
//
// Replays the skip-pointer walk of k_render_skip on the CPU for every 8x8-pixel wave of a frame of the default scene and
// counts wave steps (quiet: no awake lane's line meets the node; hit: some does), with and without a per-wave pre-cull of
// the node stream (primary: a cone around the wave's ray directions; shadow: a beam around the wave's parallel shadow rays).
// It also checks on every wave that the pre-cull is conservative: no node an awake lane's computed discriminant is >= 0 for
// may be missing from the culled stream.
//
//   g++ -O2 -ffp-contract=off -o /tmp/wave_sim tools/wave_sim.cpp && /tmp/wave_sim [level] [w] [h]
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct V3 { float x, y, z; };
static inline V3 add(V3 a, V3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
static inline V3 sub(V3 a, V3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
static inline V3 mulf(V3 a, float m) { return { a.x * m, a.y * m, a.z * m }; }
static inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline V3 normalized(V3 a) { float l = sqrtf(dot(a, a)); return mulf(a, 1.0f / l); }

struct Node { V3 c; float r; uint32_t skip; int item; int parent; };   // skip: node index behind the subtree (ITEM: i + 1)
static std::vector<Node> nodes;

static void pyramid(unsigned level, V3 p, float r, int parent)
{
    if (level == 1) { nodes.push_back({ p, r, (uint32_t)nodes.size() + 1, 1, parent }); return; }
    const int me = (int)nodes.size();
    nodes.push_back({ p, 3.0f * r, 0, 0, parent });
    nodes.push_back({ p, r, (uint32_t)nodes.size() + 1, 1, me });
    const float rn = 3.0f * r / sqrtf(12.0f);
    const int sgn[2] = { -1, 1 };
    for (int iz = 0; iz < 2; ++iz)
        for (int ix = 0; ix < 2; ++ix)
            pyramid(level - 1, add(p, { sgn[ix] * rn, rn, sgn[iz] * rn }), r * 0.5f, me);
    nodes[me].skip = (uint32_t)nodes.size();
}

static inline float dist_from_ray(V3 c, float r, V3 o, V3 d, float *disc_out)
{
    const V3 v = sub(c, o);
    const float b = dot(v, d);
    const float disc = b * b - dot(v, v) + r * r;
    *disc_out = disc;
    if (disc < 0.0f) return INFINITY;
    const float s = sqrtf(disc);
    const float t2 = b + s;
    if (t2 < 0.0f) return INFINITY;
    const float t1 = b - s;
    return t1 > 0.0f ? t1 : t2;
}

struct Tally { uint64_t quiet = 0, hit = 0, waves = 0, cand = 0, max_steps = 0, rounds = 0, tested = 0; };

int main(int argc, char **argv)
{
    const unsigned level = argc > 1 ? atoi(argv[1]) : 8, W = argc > 2 ? atoi(argv[2]) : 1920, H = argc > 3 ? atoi(argv[3]) : 1080;
    const unsigned P = argc > 4 ? atoi(argv[4]) : 8;       // wave patch edge (<= 8)
    pyramid(level, { 0.0f, -1.0f, 0.0f }, 1.0f, -1);
    const size_t n = nodes.size();
    const V3 eye = { 0, 0, -4 }, light = normalized({ -1.0f, -3.0f, 2.0f }), sdir = mulf(light, -1.0f);
    fprintf(stderr, "%zu nodes\n", n);

    // children lists (BFS expansion)
    std::vector<std::vector<int>> kids(n);
    for (size_t i = 1; i < n; ++i) kids[nodes[i].parent].push_back((int)i);

    Tally pb, pc, sb, sc;
    uint64_t violations = 0, lanes_prim = 0, lanes_shad = 0;
    std::vector<uint8_t> keep(n);
    std::vector<int> frontier, nextf;
    const float EPS = 5.9604645e-8f;
    for (unsigned y0 = 0; y0 < H; y0 += P)
        for (unsigned x0 = 0; x0 < W; x0 += P) {
            const unsigned L = P * P;
            V3 dir[64]; bool inside[64];
            float best[64]; int bitem[64]; uint32_t resume[64];
            for (unsigned l = 0; l < L; ++l) {
                const unsigned x = x0 + l % P, y = y0 + l / P;
                inside[l] = x < W && y < H;
                dir[l] = normalized({ (float)x - W / 2.0f, ((float)H - (float)y) - H / 2.0f, (float)W });
            }
            // ---- cone around the live directions
            V3 ax = { 0, 0, 0 };
            for (unsigned l = 0; l < L; ++l) if (inside[l]) ax = add(ax, dir[l]);
            ax = normalized(ax);
            float cosmin = 1.0f;
            for (unsigned l = 0; l < L; ++l) if (inside[l]) cosmin = std::min(cosmin, dot(ax, dir[l]) / sqrtf(dot(dir[l], dir[l])));
            cosmin = cosmin * (1.0f - 8 * EPS) - 8 * EPS;
            const float sinmax = sqrtf(std::max(0.0f, 1.0f - cosmin * cosmin)) * (1.0f + 8 * EPS) + 1e-7f;
            auto cone_keep = [&](const Node &nd) {
                const V3 v = sub(nd.c, eye);
                const float vv = dot(v, v), rr = nd.r * nd.r;
                const float s = fabsf(dot(v, ax));
                const float p2 = std::max(0.0f, vv - s * s);
                const float p = sqrtf(p2);
                const float reff = sqrtf(rr + 32 * EPS * (vv + rr));
                return p * cosmin - s * sinmax <= reff + 64 * EPS * sqrtf(vv);
            };
            std::fill(keep.begin(), keep.end(), 0);
            frontier.clear();
            unsigned rounds = 0; uint64_t tested = 1;
            if (cone_keep(nodes[0])) { keep[0] = 1; frontier.push_back(0); }
            while (!frontier.empty()) {
                ++rounds;
                nextf.clear();
                for (int g : frontier)
                    for (int k : kids[g]) {
                        ++tested;
                        if (cone_keep(nodes[k])) { keep[k] = 1; if (!nodes[k].item) nextf.push_back(k); }
                    }
                frontier.swap(nextf);
            }
            pc.rounds += rounds; pc.tested += tested;

            // ---- primary walk, full stream and culled stream (same per-lane state machine)
            for (int pass = 0; pass < 2; ++pass) {
                Tally &T = pass ? pc : pb;
                for (unsigned l = 0; l < L; ++l) { best[l] = INFINITY; bitem[l] = -1; resume[l] = inside[l] ? 0u : 0xFFFFFFFFu; }
                uint64_t steps = 0;
                size_t i = 0;
                while (i < n) {
                    const Node &nd = nodes[i];
                    if (pass && !keep[i]) { i = nd.skip; continue; }       // not in the culled stream (subtree gone with it)
                    bool any_cand = false, any_enter = false;
                    for (unsigned l = 0; l < L; ++l) {
                        const bool active = i >= resume[l];
                        float disc;
                        const float d = dist_from_ray(nd.c, nd.r, eye, dir[l], &disc);
                        if (active && disc >= 0.0f) {
                            any_cand = true;
                            if (!pass && !keep[i]) ++violations;
                        }
                        if (!active) continue;
                        if (!nd.item) { if (d >= best[l]) resume[l] = nd.skip; else any_enter = true; }
                        else if (!(d >= best[l])) { best[l] = d; bitem[l] = (int)i; }
                    }
                    ++steps;
                    if (any_cand) ++T.hit; else ++T.quiet;
                    i = (nd.item || any_enter) ? i + 1 : nd.skip;
                }
                T.max_steps = std::max(T.max_steps, steps);
                ++T.waves;
                if (pass) for (size_t k = 0; k < n; ++k) T.cand += keep[k];
            }

            // ---- shade -> shadow rays
            V3 sp[64]; bool need[64]; unsigned n_need = 0;
            for (unsigned l = 0; l < L; ++l) {
                need[l] = false;
                if (!inside[l]) continue;
                ++lanes_prim;
                if (best[l] == INFINITY) continue;
                const Node &it = nodes[bitem[l]];
                const V3 nrm = normalized(add(eye, sub(mulf(dir[l], best[l]), it.c)));
                const float g = dot(nrm, light);
                if (g >= 0.0f) continue;
                const V3 ns = mulf(nrm, best[l] * sqrtf(1.1920929e-7f));
                sp[l] = add(add(eye, mulf(dir[l], best[l])), ns);
                need[l] = true; ++n_need; ++lanes_shad;
            }
            if (!n_need) continue;
            // ---- beam around the parallel shadow rays: plane basis perpendicular to sdir
            V3 e1 = normalized({ -sdir.y, sdir.x, 0.0f });
            V3 e2 = { sdir.y * e1.z - sdir.z * e1.y, sdir.z * e1.x - sdir.x * e1.z, sdir.x * e1.y - sdir.y * e1.x };
            float q1[64], q2[64], ol[64], m1 = 0, m2 = 0, omin = INFINITY, omag = 0;
            float lo1 = INFINITY, hi1 = -INFINITY, lo2 = INFINITY, hi2 = -INFINITY;
            for (unsigned l = 0; l < L; ++l) if (need[l]) {
                q1[l] = dot(sp[l], e1); q2[l] = dot(sp[l], e2); ol[l] = dot(sp[l], sdir);
                lo1 = std::min(lo1, q1[l]); hi1 = std::max(hi1, q1[l]); lo2 = std::min(lo2, q2[l]); hi2 = std::max(hi2, q2[l]);
                omin = std::min(omin, ol[l]);
                omag = std::max(omag, sqrtf(dot(sp[l], sp[l])));
            }
            m1 = 0.5f * (lo1 + hi1); m2 = 0.5f * (lo2 + hi2);
            float rho = 0;
            for (unsigned l = 0; l < L; ++l) if (need[l]) rho = std::max(rho, sqrtf((q1[l] - m1) * (q1[l] - m1) + (q2[l] - m2) * (q2[l] - m2)));
            auto beam_keep = [&](const Node &nd) {
                const float w1 = dot(nd.c, e1) - m1, w2 = dot(nd.c, e2) - m2, cl = dot(nd.c, sdir);
                const float cmag = sqrtf(dot(nd.c, nd.c));
                const float scale = (cmag + omag) * (cmag + omag) + nd.r * nd.r;
                const float reff = sqrtf(nd.r * nd.r + 48 * EPS * scale);
                const float slack = 32 * EPS * (cmag + omag);
                if (sqrtf(w1 * w1 + w2 * w2) > reff + rho + slack) return false;
                if (cl + reff + slack < omin) return false;     // t2 = b + root < 0 for every ray
                return true;
            };
            std::fill(keep.begin(), keep.end(), 0);
            frontier.clear();
            rounds = 0; tested = 1;
            if (beam_keep(nodes[0])) { keep[0] = 1; frontier.push_back(0); }
            while (!frontier.empty()) {
                ++rounds;
                nextf.clear();
                for (int g : frontier)
                    for (int k : kids[g]) {
                        ++tested;
                        if (beam_keep(nodes[k])) { keep[k] = 1; if (!nodes[k].item) nextf.push_back(k); }
                    }
                frontier.swap(nextf);
            }
            sc.rounds += rounds; sc.tested += tested;
            bool occ_ref[64];
            for (int pass = 0; pass < 2; ++pass) {
                Tally &T = pass ? sc : sb;
                bool occ[64];
                for (unsigned l = 0; l < L; ++l) { occ[l] = false; resume[l] = need[l] ? 0u : 0xFFFFFFFFu; }
                uint64_t steps = 0;
                size_t i = 0;
                while (i < n) {
                    const Node &nd = nodes[i];
                    if (pass && !keep[i]) { i = nd.skip; continue; }
                    bool any_cand = false, any_enter = false, any_fin = false;
                    for (unsigned l = 0; l < L; ++l) {
                        const bool active = i >= resume[l];
                        if (!active) continue;
                        float disc;
                        const float d = dist_from_ray(nd.c, nd.r, sp[l], sdir, &disc);
                        if (disc >= 0.0f) any_cand = true;
                        const bool hit = d < INFINITY;
                        if (hit && !pass && !keep[i]) ++violations;
                        if (!nd.item) { if (!hit) resume[l] = nd.skip; else any_enter = true; }
                        else if (hit) { occ[l] = true; resume[l] = 0xFFFFFFFFu; any_fin = true; }
                    }
                    ++steps;
                    if (any_cand) ++T.hit; else ++T.quiet;
                    size_t ni = (nd.item || any_enter) ? i + 1 : nd.skip;
                    if (any_fin) {
                        uint32_t m = 0xFFFFFFFFu;
                        for (unsigned l = 0; l < L; ++l) m = std::min(m, resume[l] == 0xFFFFFFFFu ? 0xFFFFFFFFu : std::max<uint32_t>(resume[l], (uint32_t)i + 1));
                        ni = m == 0xFFFFFFFFu ? n : m;
                    }
                    i = ni;
                }
                T.max_steps = std::max(T.max_steps, steps);
                ++T.waves;
                if (pass) { for (size_t k = 0; k < n; ++k) T.cand += keep[k]; for (unsigned l = 0; l < L; ++l) if (need[l] && occ[l] != occ_ref[l]) ++violations; }
                else for (unsigned l = 0; l < L; ++l) occ_ref[l] = occ[l];
            }
        }
    auto show = [](const char *name, const Tally &t) {
        printf("%-22s waves %7llu  steps/wave %7.1f (quiet %6.1f, hit %6.1f)  max %5llu  cand/wave %6.1f  rounds %4.1f  tested/wave %6.1f\n", name,
               (unsigned long long)t.waves, double(t.quiet + t.hit) / t.waves, double(t.quiet) / t.waves, double(t.hit) / t.waves,
               (unsigned long long)t.max_steps, double(t.cand) / t.waves, double(t.rounds) / t.waves, double(t.tested) / t.waves);
    };
    show("primary full stream", pb);
    show("primary culled", pc);
    show("shadow full stream", sb);
    show("shadow culled", sc);
    printf("lanes: %llu primary, %llu shadow; conservativeness violations: %llu\n", (unsigned long long)lanes_prim,
           (unsigned long long)lanes_shad, (unsigned long long)violations);
    return violations != 0;
}
