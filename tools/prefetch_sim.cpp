// tools/prefetch_sim.cpp -- design-time estimator, not part of the product and not a parity oracle.
//
// The skip-pointer walk's step is a DEPENDENT scalar load: the next node's address (`skip`) comes out of the record just fetched, and a
// dependent s_load_dwordx8 costs a wave ~115 cycles when it hits the scalar cache, ~195 when it comes from L2 (tools/scalar_latency_probe.hip).
// This replays every 8x8-pixel wave of a frame of the default scene through the fused walk (primary, then shadow) and prices its chain under
// three fetch schemes:  d1 = today's (the quiet successor `skip` requested at the top of a step);  d2 = records that also carry
// skip2 = skip(skip) -- two quiet successors in flight;  d3 = skip3 as well.  An entered node's child is requested when the step knows
// somebody enters.  Output: the sum and the maximum of the waves' chain times per scheme, and the lengths of the quiet runs.
//
//   g++ -O2 -ffp-contract=off -o /tmp/prefetch_sim tools/prefetch_sim.cpp && /tmp/prefetch_sim [w h [L [Cq]]]
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
struct V3 { float x, y, z; };
static inline V3 add(V3 a, V3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
static inline V3 sub(V3 a, V3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
static inline V3 mulf(V3 a, float m) { return { a.x * m, a.y * m, a.z * m }; }
static inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline V3 normalized(V3 a) { float l = sqrtf(dot(a, a)); return mulf(a, 1.0f / l); }
// compacted stream: a BOUND carries its own sphere (fused); items are the leaves
struct Node { V3 c; float r, own_r; uint32_t skip; int item; };
static std::vector<Node> nodes;
static void pyramid(unsigned level, V3 p, float r)
{
    if (level == 1) { nodes.push_back({ p, r, 0, (uint32_t)nodes.size() + 1, 1 }); return; }
    const int me = (int)nodes.size();
    nodes.push_back({ p, 3.0f * r, r, 0, 0 });
    const float rn = 3.0f * r / sqrtf(12.0f);
    const int sgn[2] = { -1, 1 };
    for (int iz = 0; iz < 2; ++iz) for (int ix = 0; ix < 2; ++ix) pyramid(level - 1, add(p, { sgn[ix] * rn, rn, sgn[iz] * rn }), r * 0.5f);
    nodes[me].skip = (uint32_t)nodes.size();
}
static inline float dist(V3 c, float r, V3 o, V3 d, float *disc_out)
{
    const V3 v = sub(c, o); const float b = dot(v, d); const float disc = b * b - dot(v, v) + r * r; *disc_out = disc;
    if (disc < 0.0f) return INFINITY; const float s = sqrtf(disc); const float t2 = b + s; if (t2 < 0.0f) return INFINITY; const float t1 = b - s; return t1 > 0.0f ? t1 : t2;
}
struct Step { uint32_t node; int kind; uint32_t next; };      // kind 0 quiet, 1 candidate nobody enters / item candidate, 2 entered
static double L = 115, Cq = 45, Cc = 130, Ce = 190, Cdec = 110;
// chain time of a step sequence with prefetch depth D (1, 2, 3)
static double chain(const std::vector<Step> &seq, int D)
{
    const size_t n = nodes.size();
    auto skipk = [&](uint32_t i, int k) { for (int j = 0; j < k && i < n; ++j) i = nodes[i].skip; return i; };
    std::map<uint32_t, double> inflight;
    double t = 0;
    double ready = L;            // the first record
    for (size_t s = 0; s < seq.size(); ++s) {
        const Step &st = seq[s];
        const double start = std::max(t, ready);
        // requests at the top of the step: the quiet successors not yet in flight
        for (int k = 1; k <= D; ++k) { const uint32_t a = skipk(st.node, k); if (a <= n && !inflight.count(a)) inflight[a] = start + L; }
        const double c = st.kind == 0 ? Cq : st.kind == 1 ? Cc : Ce;
        double nready;
        if (st.kind == 2) nready = start + Cdec + L;                     // the child, requested when the step knows somebody enters
        else { auto it = inflight.find(st.next); nready = it != inflight.end() ? it->second : start + c + L; }      // (a retired shadow lane may send the wave elsewhere)
        t = start + c;
        ready = nready;
        if (st.kind == 2) inflight.clear();                              // the banks are reused
        else { for (auto it = inflight.begin(); it != inflight.end();) it = it->first <= st.next && it->first != st.next ? inflight.erase(it) : ++it; }
    }
    return t;
}
int main(int argc, char **argv)
{
    const unsigned W = argc > 1 ? atoi(argv[1]) : 1920, H = argc > 2 ? atoi(argv[2]) : 1080;
    if (argc > 3) L = atof(argv[3]);
    if (argc > 4) Cq = atof(argv[4]);
    pyramid(8, { 0, -1, 0 }, 1.0f);
    const size_t n = nodes.size();
    const V3 eye = { 0, 0, -4 }, light = normalized({ -1, -3, 2 }), sdir = mulf(light, -1.0f);
    double sum[4] = { 0, 0, 0, 0 }, mx[4] = { 0, 0, 0, 0 };
    std::vector<double> all1, all2, all3;
    uint64_t steps = 0, quiet = 0, entered = 0, runs[9] = { 0 };
    for (unsigned y0 = 0; y0 < H; y0 += 8) for (unsigned x0 = 0; x0 < W; x0 += 8) {
        V3 dir[64]; bool in[64]; float best[64]; int bi[64]; bool own[64]; uint32_t res[64];
        for (unsigned l = 0; l < 64; ++l) { unsigned x = x0 + l % 8, y = y0 + l / 8; in[l] = x < W && y < H; dir[l] = normalized({ (float)x - W / 2.0f, ((float)H - (float)y) - H / 2.0f, (float)W }); best[l] = INFINITY; bi[l] = -1; own[l] = false; res[l] = in[l] ? 0u : 0xFFFFFFFFu; }
        std::vector<Step> seq;
        for (size_t i = 0; i < n;) {
            const Node &nd = nodes[i];
            bool cand = false, enter = false;
            for (unsigned l = 0; l < 64; ++l) {
                if (i < res[l]) continue;
                float disc; const float d = dist(nd.c, nd.r, eye, dir[l], &disc);
                if (disc >= 0) cand = true;
                if (!nd.item) { if (d >= best[l]) res[l] = nd.skip; else enter = true; }
                else if (!(d >= best[l])) { best[l] = d; bi[l] = (int)i; own[l] = false; }
            }
            size_t ni = nd.skip;
            if (!nd.item && enter) {
                for (unsigned l = 0; l < 64; ++l) {
                    if (i < res[l]) continue;
                    float disc; const float d = dist(nd.c, nd.own_r, eye, dir[l], &disc);
                    if (!(d >= best[l])) { best[l] = d; bi[l] = (int)i; own[l] = true; }
                }
                ni = i + 1;
            }
            seq.push_back({ (uint32_t)i, (!nd.item && enter) ? 2 : cand ? 1 : 0, (uint32_t)ni });
            i = ni;
        }
        // shadow
        V3 sp[64]; bool need[64]; unsigned nn = 0;
        for (unsigned l = 0; l < 64; ++l) { need[l] = false; if (!in[l] || best[l] == INFINITY) continue; const Node &it = nodes[bi[l]]; const V3 nrm = normalized(add(eye, sub(mulf(dir[l], best[l]), it.c))); if (dot(nrm, light) >= 0) continue; sp[l] = add(add(eye, mulf(dir[l], best[l])), mulf(nrm, best[l] * sqrtf(1.1920929e-7f))); need[l] = true; ++nn; }
        std::vector<Step> sseq;
        if (nn) {
            for (unsigned l = 0; l < 64; ++l) res[l] = need[l] ? 0u : 0xFFFFFFFFu;
            for (size_t i = 0; i < n;) {
                const Node &nd = nodes[i];
                bool cand = false, enter = false, fin = false;
                for (unsigned l = 0; l < 64; ++l) {
                    if (i < res[l]) continue;
                    float disc; const bool hit = dist(nd.c, nd.r, sp[l], sdir, &disc) < INFINITY;
                    if (disc >= 0) cand = true;
                    if (!nd.item) {
                        if (!hit) res[l] = nd.skip;
                        else { float d2; if (dist(nd.c, nd.own_r, sp[l], sdir, &d2) < INFINITY) { res[l] = 0xFFFFFFFFu; fin = true; } else enter = true; }
                    } else if (hit) { res[l] = 0xFFFFFFFFu; fin = true; }
                }
                size_t ni = (!nd.item && enter) ? i + 1 : nd.skip;
                if (fin) { uint32_t m = 0xFFFFFFFFu; for (unsigned l = 0; l < 64; ++l) m = std::min(m, res[l] == 0xFFFFFFFFu ? 0xFFFFFFFFu : std::max<uint32_t>(res[l], (uint32_t)i + 1)); ni = m == 0xFFFFFFFFu ? n : m; }
                sseq.push_back({ (uint32_t)i, (!nd.item && enter && !fin) ? 2 : cand ? 1 : 0, (uint32_t)ni });
                i = ni;
            }
        }
        double t[4] = { 0, 0, 0, 0 };
        for (int D = 1; D <= 3; ++D) t[D] = chain(seq, D) + (sseq.empty() ? 0.0 : chain(sseq, D));
        for (int D = 1; D <= 3; ++D) { sum[D] += t[D]; mx[D] = std::max(mx[D], t[D]); }
        all1.push_back(t[1]); all2.push_back(t[2]); all3.push_back(t[3]);
        unsigned run = 0;
        for (const std::vector<Step> *sq : { &seq, &sseq })
            for (const Step &s : *sq) {
                ++steps;
                if (s.kind == 2) { ++entered; runs[std::min(run, 8u)]++; run = 0; } else { if (s.kind == 0) ++quiet; ++run; }
            }
    }
    printf("%ux%u, L = %.0f cycles, quiet step %.0f, candidate %.0f, entered %.0f (child requested after %.0f)\n", W, H, L, Cq, Cc, Ce, Cdec);
    printf("steps %llu: quiet %llu (%.1f %%), entered %llu (%.1f %%)\n", (unsigned long long)steps, (unsigned long long)quiet, 100.0 * quiet / steps, (unsigned long long)entered, 100.0 * entered / steps);
    printf("steps between entered nodes (run length -> count):"); for (int k = 0; k < 9; ++k) printf(" %d%s:%llu", k, k == 8 ? "+" : "", (unsigned long long)runs[k]); printf("\n");
    for (int D = 1; D <= 3; ++D) printf("depth %d: sum of wave chains %.1f Mcycles (%.3f of depth 1), longest wave %.0f cycles = %.1f us at 2.4 GHz (%.3f of depth 1)\n", D, sum[D] / 1e6, sum[D] / sum[1], mx[D], mx[D] / 2400.0, mx[D] / mx[1]);
    std::sort(all1.begin(), all1.end()); std::sort(all2.begin(), all2.end()); std::sort(all3.begin(), all3.end());
    const size_t p99 = all1.size() * 99 / 100;
    printf("p99 wave: %.0f / %.0f / %.0f cycles\n", all1[p99], all2[p99], all3[p99]);
    return 0;
}
