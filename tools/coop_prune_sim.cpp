// tools/coop_prune_sim.cpp -- design-time estimator (round 5); not part of the product, not an oracle.
//
// VERDICT r4 item 6 asked for the cooperative gather (csrc/rt_coop.hpp) to be PRUNED by the nearest hit found so far.  This replays, for every
// QxQ quad of a frame of the default scene, the kernel's own schedule -- a LIFO work list of (ray, group) pairs, 64 / fan-out pairs per round,
// one (ray, child) test per lane -- twice: as the kernel runs it (the whole closure), and with a group dropped, at push and at pop time, when
// a certified lower bound of everything below it is farther than the ray's nearest item at the start of the round.  It reports tests and
// ROUNDS per quad (a round is one dependent trip to L2 + ~75 instructions: the rounds are what a cooperative wave's duration is made of)
// by the quad's heaviest pixel, and checks the pruned result against the reference's DFS.
//
// The certificate (NOTES.md R5): with every descendant item inside the bound (checked per scene in double) the item's true entry distance is
// >= the bound's; an f32 distance of primitive.rs:55-72 is within sqrt(K eps)(|v| + R) of the true one (K ~ 12: the discriminant's absolute
// error is <= K eps |v|^2, and at grazing incidence the root turns that into its square root), so lb = d_bound (1 - rho) with rho = 2^-8 is
// below every descendant's COMPUTED distance whenever R / |v| <= 1/3, and the winner theorem of DESIGN.md 4.4 goes through unchanged with
// "pruned items are strictly farther than F".  So pruning can be made sound.  What this tool shows is that it does not pay: the quads a
// small frame waits for (heaviest pixel >= 150 reference tests) lose 1 - 8 % of their rounds (3 - 19 % with every round's pushes sorted by
// distance, which the kernel cannot do for free), because a heavy pixel's ray grazes bounds and hits nothing early: its closure IS what the
// reference's DFS visits.
//
//   g++ -O2 -ffp-contract=off -o /tmp/coop_prune_sim tools/coop_prune_sim.cpp && /tmp/coop_prune_sim [level] [w] [h] [Q] [rho] [sorted 0|1|2]
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
    const unsigned level = argc > 1 ? atoi(argv[1]) : 8, W = argc > 2 ? atoi(argv[2]) : 800, H = argc > 3 ? atoi(argv[3]) : 600;
    const int Q = argc > 4 ? atoi(argv[4]) : 2;       // quad edge
    const float rho = argc > 5 ? atof(argv[5]) : 1.0f / 256.0f;
    const int sorted = argc > 6 ? atoi(argv[6]) : 0;  // push children far-to-near so the nearest is popped first
    pyramid(level, { 0.0f, -1.0f, 0.0f }, 1.0f, -1, 0);
    const size_t n = nodes.size();
    const V3 eye = { 0, 0, -4 };
    std::vector<std::vector<int>> kids(n);
    for (size_t i = 1; i < n; ++i) kids[nodes[i].parent].push_back((int)i);
    const unsigned fan = 5, per = 64 / fan;
    const unsigned edges[] = { 0, 50, 100, 150, 200, 300, 400, 100000 };
    constexpr int NC = 7;
    struct Cls { uint64_t quads = 0, ref = 0, t0 = 0, r0 = 0, t1 = 0, r1 = 0, maxr0 = 0, maxr1 = 0, mism = 0, maxstack = 0; } cls[NC];
    struct Pair { int node; int ray; float anc, lb; };
    for (unsigned y0 = 0; y0 < H; y0 += Q)
        for (unsigned x0 = 0; x0 < W; x0 += Q) {
            V3 dir[16]; float rbest[16]; int ritem[16]; unsigned refmax = 0, refsum = 0; int nr = 0;
            for (int l = 0; l < Q * Q; ++l) {
                const unsigned x = x0 + l % Q, y = y0 + l / Q;
                if (x >= W || y >= H) continue;
                dir[nr] = normalized({ (float)x - W / 2.0f, ((float)H - (float)y) - H / 2.0f, (float)W });
                float best = INFINITY; int bitem = -1; unsigned t = 0;
                for (size_t i = 0; i < n;) {
                    const Node &nd = nodes[i];
                    const float d = dist_from_ray(nd.c, nd.r, eye, dir[nr]);
                    ++t;
                    if (!nd.item) i = d >= best ? nd.skip : i + 1;
                    else { if (!(d >= best)) { best = d; bitem = (int)i; } ++i; }
                }
                rbest[nr] = best; ritem[nr] = bitem; refmax = std::max(refmax, t); refsum += t; ++nr;
            }
            int c = 0;
            while (refmax >= edges[c + 1]) ++c;
            Cls &C = cls[c];
            ++C.quads; C.ref += refsum;
            for (int pass = 0; pass < 2; ++pass) {
                std::vector<Pair> st;
                float best[16]; int bitem[16]; float banc[16];
                for (int r = 0; r < nr; ++r) { best[r] = INFINITY; bitem[r] = -1; banc[r] = 0; st.push_back({ -1, r, 0.0f, 0.0f }); }
                uint64_t tests = 0, rounds = 0; size_t maxst = 0;
                while (!st.empty()) {
                    ++rounds;
                    const size_t ne = std::min<size_t>(per, st.size());
                    std::vector<Pair> take(st.end() - ne, st.end());
                    st.resize(st.size() - ne);
                    float U[16];
                    for (int r = 0; r < nr; ++r) U[r] = best[r];
                    std::vector<Pair> pushes;
                    for (size_t e = 0; e < ne; ++e) {
                        const Pair &p = take[ne - 1 - e];
                        if (pass && p.lb > U[p.ray]) continue;          // pop-time prune (lanes idle)
                        std::vector<int> ch;
                        if (p.node < 0) ch.push_back(0); else ch = kids[p.node];
                        std::vector<Pair> mine;
                        for (int k : ch) {
                            const Node &nd = nodes[k];
                            const float d = dist_from_ray(nd.c, nd.r, eye, dir[p.ray]);
                            ++tests;
                            if (!(d < INFINITY)) continue;
                            if (nd.item) {
                                if (d < best[p.ray] || (d == best[p.ray] && k < bitem[p.ray])) { best[p.ray] = d; bitem[p.ray] = k; banc[p.ray] = p.anc; }
                            } else {
                                const float lb = std::max(p.lb, d * (1.0f - rho));
                                if (pass && lb > U[p.ray]) continue;
                                mine.push_back({ k, p.ray, std::max(p.anc, d), lb });
                            }
                        }
                        if (sorted) std::sort(mine.begin(), mine.end(), [](const Pair &a, const Pair &b) { return a.lb > b.lb; });
                        for (auto &m : mine) pushes.push_back(m);
                    }
                    if (sorted == 2) std::sort(pushes.begin(), pushes.end(), [](const Pair &a, const Pair &b) { return a.lb > b.lb; });
                    for (auto &m : pushes) st.push_back(m);
                    maxst = std::max(maxst, st.size());
                }
                bool bad = false;
                for (int r = 0; r < nr; ++r) {
                    const bool failed = best[r] < INFINITY && banc[r] > best[r];
                    if (!failed && (best[r] != rbest[r] || (rbest[r] < INFINITY && bitem[r] != ritem[r]))) bad = true;
                }
                if (!pass) { C.t0 += tests; C.r0 += rounds; C.maxr0 = std::max<uint64_t>(C.maxr0, rounds); }
                else { C.t1 += tests; C.r1 += rounds; C.maxr1 = std::max<uint64_t>(C.maxr1, rounds); C.mism += bad; }
                C.maxstack = std::max<uint64_t>(C.maxstack, maxst);
            }
        }
    printf("%ux%u L%u quads %dx%d rho %g sorted %d\n", W, H, level, Q, Q, rho, sorted);
    printf("class (max ref tests/px)  quads   ref tests/quad | gather tests  rounds (max) | pruned tests  rounds (max) | mismatches  max stack\n");
    for (int c = 0; c < NC; ++c) {
        const Cls &C = cls[c];
        if (!C.quads) continue;
        printf("  %4u .. %-6u  %8llu   %8.1f | %8.1f  %6.1f (%3llu) | %8.1f  %6.1f (%3llu) | %llu  %llu\n", edges[c], edges[c + 1], (unsigned long long)C.quads,
               double(C.ref) / C.quads, double(C.t0) / C.quads, double(C.r0) / C.quads, (unsigned long long)C.maxr0, double(C.t1) / C.quads, double(C.r1) / C.quads,
               (unsigned long long)C.maxr1, (unsigned long long)C.mism, (unsigned long long)C.maxstack);
    }
    return 0;
}
