// tools/coop_sim.cpp -- design-time estimator for the lane-cooperative walk (csrc/rt_coop.hpp); not part of the product, not an oracle.
//
// For every pixel of a frame of the default scene it replays (a) the reference's DFS with hit.distance culling (group.rs:72-83) and
// (b) the breadth-first GATHER the cooperative kernel makes: every node all of whose ancestors return a finite bound distance, level by
// level, no culling by hit.distance.  It reports how much larger (b) is than (a) per cost class, checks that the nearest item of (b)
// by (distance, DFS index) is the reference's hit whenever every ancestor bound of that item is no farther than the item itself (the
// exactness condition of DESIGN.md 4.4), and counts how often the condition fails.
//
//   g++ -O2 -ffp-contract=off -o /tmp/coop_sim tools/coop_sim.cpp && /tmp/coop_sim [level] [w] [h]
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

struct Node { V3 c; float r; uint32_t skip; int item; int parent; int depth; };
static std::vector<Node> nodes;

static void pyramid(unsigned level, V3 p, float r, int parent, int depth)
{
    if (level == 1) { nodes.push_back({ p, r, (uint32_t)nodes.size() + 1, 1, parent, depth }); return; }
    const int me = (int)nodes.size();
    nodes.push_back({ p, 3.0f * r, 0, 0, parent, depth });
    nodes.push_back({ p, r, (uint32_t)nodes.size() + 1, 1, me, depth + 1 });
    const float rn = 3.0f * r / sqrtf(12.0f);
    const int sgn[2] = { -1, 1 };
    for (int iz = 0; iz < 2; ++iz)
        for (int ix = 0; ix < 2; ++ix)
            pyramid(level - 1, add(p, { sgn[ix] * rn, rn, sgn[iz] * rn }), r * 0.5f, me, depth + 1);
    nodes[me].skip = (uint32_t)nodes.size();
}

static inline float dist_from_ray(V3 c, float r, V3 o, V3 d)
{
    const V3 v = sub(c, o);
    const float b = dot(v, d);
    const float disc = b * b - dot(v, v) + r * r;
    if (disc < 0.0f) return INFINITY;
    const float s = sqrtf(disc);
    const float t2 = b + s;
    if (t2 < 0.0f) return INFINITY;
    const float t1 = b - s;
    return t1 > 0.0f ? t1 : t2;
}

int main(int argc, char **argv)
{
    const unsigned level = argc > 1 ? atoi(argv[1]) : 8, W = argc > 2 ? atoi(argv[2]) : 1920, H = argc > 3 ? atoi(argv[3]) : 1080;
    pyramid(level, { 0.0f, -1.0f, 0.0f }, 1.0f, -1, 0);
    const size_t n = nodes.size();
    const V3 eye = { 0, 0, -4 }, light = normalized({ -1.0f, -3.0f, 2.0f }), sdir = mulf(light, -1.0f);
    std::vector<std::vector<int>> kids(n);
    for (size_t i = 1; i < n; ++i) kids[nodes[i].parent].push_back((int)i);

    // classes by the reference's tests per pixel (primary + shadow)
    const unsigned edges[] = { 0, 25, 50, 100, 150, 200, 300, 400, 100000 };
    constexpr int NC = 8;
    struct Cls { uint64_t px = 0, ref_p = 0, ref_s = 0, clo_p = 0, clo_s = 0, prn_p = 0, lev_p = 0, lev_s = 0; } cls[NC];
    uint64_t check_fail = 0, mismatch = 0, prune_mismatch = 0, shadow_mismatch = 0, total_ref = 0;
    std::vector<uint32_t> cost((size_t)W * H), coop_cost((size_t)W * H);
    std::vector<int> frontier, nextf;
    std::vector<float> fanc, nanc;
    for (unsigned y = 0; y < H; ++y)
        for (unsigned x = 0; x < W; ++x) {
            const V3 dir = normalized({ (float)x - W / 2.0f, ((float)H - (float)y) - H / 2.0f, (float)W });
            // (a) reference DFS
            float best = INFINITY; int bitem = -1; unsigned ref_p = 0;
            for (size_t i = 0; i < n;) {
                const Node &nd = nodes[i];
                const float d = dist_from_ray(nd.c, nd.r, eye, dir);
                ++ref_p;
                if (!nd.item) { i = d >= best ? nd.skip : i + 1; }
                else { if (!(d >= best)) { best = d; bitem = (int)i; } ++i; }
            }
            // (b) gather: BFS over the closure; (c) the same with bounds pruned against the nearest item of the levels above
            unsigned clo_p = 0, lev_p = 0, prn_p = 0;
            float cbest = INFINITY; int citem = -1; float canc = 0.0f;
            for (int pass = 0; pass < 2; ++pass) {
                frontier.assign(1, 0); fanc.assign(1, 0.0f);
                float ub = INFINITY, pbest = INFINITY; int pitem = -1;
                unsigned cnt = 0, lev = 0;
                while (!frontier.empty()) {
                    ++lev;
                    nextf.clear(); nanc.clear();
                    float lev_ub = ub;
                    for (size_t k = 0; k < frontier.size(); ++k) {
                        const Node &nd = nodes[frontier[k]];
                        const float d = dist_from_ray(nd.c, nd.r, eye, dir);
                        ++cnt;
                        if (nd.item) {
                            if (d < pbest || (d == pbest && d < INFINITY && frontier[k] < pitem)) { pbest = d; pitem = frontier[k]; if (!pass) canc = fanc[k]; }
                            if (d < lev_ub) lev_ub = d;
                        } else if (d < INFINITY && !(pass && d > ub)) {
                            for (int c : kids[frontier[k]]) { nextf.push_back(c); nanc.push_back(std::max(fanc[k], d)); }
                        }
                    }
                    ub = lev_ub;
                    frontier.swap(nextf); fanc.swap(nanc);
                }
                if (!pass) { clo_p = cnt; lev_p = lev; cbest = pbest; citem = pitem; }
                else { prn_p = cnt; if (pbest != best || (best < INFINITY && pitem != bitem)) ++prune_mismatch; }
            }
            const bool ok = !(cbest < INFINITY) || canc <= cbest;
            if (!ok) ++check_fail;
            else if (cbest != best || (best < INFINITY && citem != bitem)) ++mismatch;
            // shade -> shadow ray
            unsigned ref_s = 0, clo_s = 0, lev_s = 0;
            if (best < INFINITY) {
                const Node &it = nodes[bitem];
                const V3 nrm = normalized(add(eye, sub(mulf(dir, best), it.c)));
                const float g = dot(nrm, light);
                if (g < 0.0f) {
                    const V3 ns = mulf(nrm, best * sqrtf(1.1920929e-7f));
                    const V3 sp = add(add(eye, mulf(dir, best)), ns);
                    bool occ = false;
                    for (size_t i = 0; i < n;) {
                        const Node &nd = nodes[i];
                        const float d = dist_from_ray(nd.c, nd.r, sp, sdir);
                        ++ref_s;
                        if (!nd.item) i = d < INFINITY ? i + 1 : nd.skip;
                        else { if (d < INFINITY) { occ = true; break; } ++i; }
                    }
                    bool cocc = false;
                    frontier.assign(1, 0);
                    while (!frontier.empty() && !cocc) {
                        ++lev_s;
                        nextf.clear();
                        for (int f : frontier) {
                            const Node &nd = nodes[f];
                            const float d = dist_from_ray(nd.c, nd.r, sp, sdir);
                            ++clo_s;
                            if (nd.item) { if (d < INFINITY) cocc = true; }
                            else if (d < INFINITY) for (int c : kids[f]) nextf.push_back(c);
                        }
                        frontier.swap(nextf);
                    }
                    if (cocc != occ) ++shadow_mismatch;
                }
            }
            const unsigned ref = ref_p + ref_s;
            total_ref += ref;
            cost[(size_t)y * W + x] = ref;
            coop_cost[(size_t)y * W + x] = clo_p + clo_s;
            int c = 0;
            while (ref >= edges[c + 1]) ++c;
            Cls &C = cls[c];
            ++C.px; C.ref_p += ref_p; C.ref_s += ref_s; C.clo_p += clo_p; C.clo_s += clo_s; C.prn_p += prn_p; C.lev_p += lev_p; C.lev_s += lev_s;
        }
    printf("%ux%u L%u: %zu nodes, %llu reference tests\n", W, H, level, n, (unsigned long long)total_ref);
    printf("class (ref tests/px)   pixels   ref prim  ref shad | gather prim (pruned)  gather shad | levels p / s\n");
    for (int c = 0; c < NC; ++c) {
        const Cls &C = cls[c];
        if (!C.px) continue;
        printf("  %4u .. %-6u     %8llu   %7.1f  %7.1f  |  %7.1f  (%7.1f)    %7.1f     |  %4.1f / %4.1f\n", edges[c], edges[c + 1], (unsigned long long)C.px,
               double(C.ref_p) / C.px, double(C.ref_s) / C.px, double(C.clo_p) / C.px, double(C.prn_p) / C.px, double(C.clo_s) / C.px,
               double(C.lev_p) / C.px, double(C.lev_s) / C.px);
    }
    printf("winner check failed on %llu pixels; gather != reference on %llu checked pixels; pruned gather != reference on %llu; shadow != on %llu\n",
           (unsigned long long)check_fail, (unsigned long long)mismatch, (unsigned long long)prune_mismatch, (unsigned long long)shadow_mismatch);
    // 16x16 blocks by their heaviest pixel: how many blocks hold the chains a frame waits for
    const unsigned bw = (W + 15) / 16, bh = (H + 15) / 16;
    std::vector<std::pair<unsigned, unsigned>> blk;      // (max ref cost, sum gather cost)
    for (unsigned by = 0; by < bh; ++by)
        for (unsigned bx = 0; bx < bw; ++bx) {
            unsigned m = 0, s = 0;
            for (unsigned y = by * 16; y < std::min(H, by * 16 + 16); ++y)
                for (unsigned x = bx * 16; x < std::min(W, bx * 16 + 16); ++x) { m = std::max(m, cost[(size_t)y * W + x]); s += coop_cost[(size_t)y * W + x]; }
            blk.push_back({ m, s });
        }
    std::sort(blk.begin(), blk.end(), [](auto a, auto b) { return a.first > b.first; });
    printf("blocks by heaviest pixel: ");
    for (unsigned thr : { 400u, 300u, 250u, 200u, 150u, 120u, 100u }) {
        size_t k = 0; uint64_t g = 0;
        while (k < blk.size() && blk[k].first >= thr) { g += blk[k].second; ++k; }
        printf(" >=%u: %zu blocks (%llu gather tests)", thr, k, (unsigned long long)g);
    }
    printf(" of %zu\n", blk.size());

    // ---- pixel-granular split by a 256^2 cost map (what rt_capi.hip's cost map offers): pixels whose map cell is >= thr go to the cooperative
    // pass, the classic 8x8 waves walk the rest.  Reports the cooperative pass's tests and the classic waves' step counts (union walk).
    {
        const int R = 256;
        std::vector<uint32_t> map((size_t)R * R, 0);
        for (int Y = 0; Y < R; ++Y)
            for (int X = 0; X < R; ++X) {
                const V3 dir = normalized({ (float)X - R / 2.0f, ((float)R - (float)Y) - R / 2.0f, (float)R });
                float best = INFINITY; int bitem = -1; unsigned t = 0;
                for (size_t i = 0; i < n;) {
                    const Node &nd = nodes[i];
                    const float d = dist_from_ray(nd.c, nd.r, eye, dir);
                    ++t;
                    if (!nd.item) i = d >= best ? nd.skip : i + 1;
                    else { if (!(d >= best)) { best = d; bitem = (int)i; } ++i; }
                }
                if (best < INFINITY) {
                    const Node &it = nodes[bitem];
                    const V3 nrm = normalized(add(eye, sub(mulf(dir, best), it.c)));
                    if (dot(nrm, light) < 0.0f) {
                        const V3 sp = add(add(eye, mulf(dir, best)), mulf(nrm, best * sqrtf(1.1920929e-7f)));
                        for (size_t i = 0; i < n;) {
                            const Node &nd = nodes[i];
                            const float d = dist_from_ray(nd.c, nd.r, sp, sdir);
                            ++t;
                            if (!nd.item) i = d < INFINITY ? i + 1 : nd.skip;
                            else { if (d < INFINITY) break; ++i; }
                        }
                    }
                }
                map[(size_t)Y * R + X] = t;
            }
        auto col = [&](unsigned x) { return std::min(R - 1, std::max(0, (int)((uint64_t)x * R / W))); };
        auto row = [&](unsigned y) { return std::min(R - 1, std::max(0, (int)std::floor(((double)y - H / 2.0) * R / W + R / 2.0))); };
        for (unsigned thr : { 1000000u, 300u, 200u, 150u, 120u, 100u, 80u }) {
            // dilate: a pixel is heavy if any map cell within 1 of its own is >= thr
            auto heavy = [&](unsigned x, unsigned y) {
                const int X = col(x), Y = row(y);
                for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx) {
                    const int xx = std::min(R - 1, std::max(0, X + dx)), yy = std::min(R - 1, std::max(0, Y + dy));
                    if (map[(size_t)yy * R + xx] >= thr) return true;
                }
                return false;
            };
            uint64_t hpx = 0, htests = 0, missed = 0;
            for (unsigned y = 0; y < H; ++y) for (unsigned x = 0; x < W; ++x) {
                if (heavy(x, y)) { ++hpx; htests += coop_cost[(size_t)y * W + x]; }
                else if (cost[(size_t)y * W + x] >= thr + thr / 4) ++missed;
            }
            // classic remainder: union walk per 8x8 wave (primary + shadow), only light pixels live
            std::vector<unsigned> wsteps;
            uint64_t sum_steps = 0;
            for (unsigned y0 = 0; y0 < H; y0 += 8) for (unsigned x0 = 0; x0 < W; x0 += 8) {
                V3 dir[64]; bool live[64]; float best[64]; int bitem[64]; uint32_t resume[64]; unsigned nl = 0;
                for (unsigned l = 0; l < 64; ++l) {
                    const unsigned x = x0 + l % 8, y = y0 + l / 8;
                    live[l] = x < W && y < H && !heavy(x, y);
                    nl += live[l];
                    dir[l] = normalized({ (float)x - W / 2.0f, ((float)H - (float)y) - H / 2.0f, (float)W });
                    best[l] = INFINITY; bitem[l] = -1; resume[l] = live[l] ? 0u : 0xFFFFFFFFu;
                }
                if (!nl) continue;
                unsigned steps = 0;
                for (size_t i = 0; i < n;) {
                    const Node &nd = nodes[i];
                    bool enter = false;
                    for (unsigned l = 0; l < 64; ++l) {
                        if (i < resume[l]) continue;
                        const float d = dist_from_ray(nd.c, nd.r, eye, dir[l]);
                        if (!nd.item) { if (d >= best[l]) resume[l] = nd.skip; else enter = true; }
                        else if (!(d >= best[l])) { best[l] = d; bitem[l] = (int)i; }
                    }
                    ++steps;
                    i = (nd.item || enter) ? i + 1 : nd.skip;
                }
                V3 sp[64]; bool need[64]; unsigned nn = 0;
                for (unsigned l = 0; l < 64; ++l) {
                    need[l] = false;
                    if (!live[l] || best[l] == INFINITY) continue;
                    const Node &it = nodes[bitem[l]];
                    const V3 nrm = normalized(add(eye, sub(mulf(dir[l], best[l]), it.c)));
                    if (dot(nrm, light) >= 0.0f) continue;
                    sp[l] = add(add(eye, mulf(dir[l], best[l])), mulf(nrm, best[l] * sqrtf(1.1920929e-7f)));
                    need[l] = true; ++nn;
                }
                if (nn) {
                    for (unsigned l = 0; l < 64; ++l) resume[l] = need[l] ? 0u : 0xFFFFFFFFu;
                    for (size_t i = 0; i < n;) {
                        const Node &nd = nodes[i];
                        bool enter = false, fin = false;
                        for (unsigned l = 0; l < 64; ++l) {
                            if (i < resume[l]) continue;
                            const bool hit = dist_from_ray(nd.c, nd.r, sp[l], sdir) < INFINITY;
                            if (!nd.item) { if (!hit) resume[l] = nd.skip; else enter = true; }
                            else if (hit) { resume[l] = 0xFFFFFFFFu; fin = true; }
                        }
                        ++steps;
                        size_t ni = (nd.item || enter) ? i + 1 : nd.skip;
                        if (fin) {
                            uint32_t m = 0xFFFFFFFFu;
                            for (unsigned l = 0; l < 64; ++l) m = std::min(m, resume[l] == 0xFFFFFFFFu ? 0xFFFFFFFFu : std::max<uint32_t>(resume[l], (uint32_t)i + 1));
                            ni = m == 0xFFFFFFFFu ? n : m;
                        }
                        i = ni;
                    }
                }
                wsteps.push_back(steps); sum_steps += steps;
            }
            std::sort(wsteps.begin(), wsteps.end());
            auto pc = [&](double p) { return wsteps[std::min(wsteps.size() - 1, (size_t)(p * wsteps.size()))]; };
            printf("thr %7u: coop %7llu px (%.2f %%), %9llu gather tests; light pixels with ref cost >= 1.25 thr: %llu | classic 8x8 waves %zu: steps sum %llu  median %u p90 %u p99 %u max %u\n",
                   thr, (unsigned long long)hpx, 100.0 * hpx / ((double)W * H), (unsigned long long)htests, (unsigned long long)missed, wsteps.size(),
                   (unsigned long long)sum_steps, pc(0.5), pc(0.9), pc(0.99), wsteps.back());
        }
    }
    return (mismatch || shadow_mismatch) ? 1 : 0;
}
