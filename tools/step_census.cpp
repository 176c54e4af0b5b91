// tools/step_census.cpp -- design-time estimator, not part of the product and not a parity oracle.
//
// Census of the step types of the fused, filtered skip-pointer walk for every 8x8-pixel wave of a frame of the default scene (primary:
// quiet / BOUND with a candidate that nobody enters / entered, own sphere quiet / entered, own sphere a candidate / ITEM candidate;
// shadow: quiet / hit) with the vector instructions each costs in rt_skip_rot.hpp, and of the primary BOUND steps that the root-free
// decision of round 4 settles (rt_skip.hpp bound_shortcut_verdict: b <= 0, b < hit.distance, disc (1 + 2^-20) <= RN(b - hit.distance)^2) --
// checking on the way that the verdict never contradicts the reference's `d >= hit.distance`.
//
//   g++ -O2 -ffp-contract=off -o /tmp/step_census tools/step_census.cpp && /tmp/step_census [w h]
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
struct Node { V3 c; float r; uint32_t skip; int item; };
static std::vector<Node> nodes;
static void pyramid(unsigned level, V3 p, float r)
{
    if (level == 1) { nodes.push_back({ p, r, (uint32_t)nodes.size() + 1, 1 }); return; }
    const int me = (int)nodes.size();
    nodes.push_back({ p, 3.0f * r, 0, 0 });
    nodes.push_back({ p, r, (uint32_t)nodes.size() + 1, 1 });
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
int main(int argc, char **argv)
{
    const unsigned W = argc > 1 ? atoi(argv[1]) : 1920, H = argc > 2 ? atoi(argv[2]) : 1080;
    pyramid(8, { 0, -1, 0 }, 1.0f);
    const size_t n = nodes.size();
    const V3 eye = { 0, 0, -4 }, light = normalized({ -1, -3, 2 }), sdir = mulf(light, -1.0f);
    uint64_t pQ = 0, pB0 = 0, pB1 = 0, pB2 = 0, pI = 0, pLeafQ = 0, sQ = 0, sH = 0, waves = 0, swaves = 0;
    uint64_t dec1 = 0, dec2 = 0, dec1_b0 = 0;    // B0 steps in which every candidate lane has b >= best (root-free "no" possible) 
    for (unsigned y0 = 0; y0 < H; y0 += 8) for (unsigned x0 = 0; x0 < W; x0 += 8) {
        V3 dir[64]; bool in[64]; float best[64]; int bi[64]; uint32_t res[64];
        for (unsigned l = 0; l < 64; ++l) { unsigned x = x0 + l % 8, y = y0 + l / 8; in[l] = x < W && y < H; dir[l] = normalized({ (float)x - W / 2.0f, ((float)H - (float)y) - H / 2.0f, (float)W }); best[l] = INFINITY; bi[l] = -1; res[l] = in[l] ? 0u : 0xFFFFFFFFu; }
        ++waves;
        for (size_t i = 0; i < n;) {
            const Node &nd = nodes[i];
            bool cand = false, enter = false, undecided = false, undecided2 = false;
            for (unsigned l = 0; l < 64; ++l) {
                if (i < res[l]) continue;
                float disc; const float d = dist(nd.c, nd.r, eye, dir[l], &disc);
                if (disc >= 0) cand = true;
                if (disc >= 0 && !nd.item) {
                    const V3 v = sub(nd.c, eye); const float b = dot(v, dir[l]);
                    const float w = b - best[l];
                    const bool A = w < 0.0f;                      // best = INF: w = -INF
                    const float p = w * w, k = 1.0f + 0x1p-20f;
                    const bool N = !A && disc * k <= p;
                    const float q = fmaf(best[l], 0x1p-22f, w), r2 = q * q;
                    const bool G = !A && w <= best[l] && disc >= r2 * k;
                    if (!A && !N) undecided = true;
                    if (!A && !N && !G) undecided2 = true;
                    // sanity: the classification must agree with the reference's decision
                    const bool go = !(d >= best[l]);
                    if ((A && !go) || (N && go) || (G && !go)) { fprintf(stderr, "CLASSIFICATION WRONG %d %d %d go %d b %g best %g disc %g\n", A, N, G, go, b, best[l], disc); }
                }
                if (!nd.item) { if (d >= best[l]) res[l] = nd.skip; else enter = true; }
                else if (!(d >= best[l])) { best[l] = d; bi[l] = (int)i; }
            }
            if (nd.item) { if (cand) ++pI; else ++pLeafQ; ++i; continue; }
            if (!cand) { ++pQ; i = nd.skip; continue; }
            if (!undecided) ++dec1; if (!undecided2) ++dec2;
            if (!enter) { ++pB0; if (!undecided) ++dec1_b0; i = nd.skip; continue; }
            // fused: the own sphere (next node) for the lanes that entered
            const Node &own = nodes[i + 1];
            bool ocand = false;
            for (unsigned l = 0; l < 64; ++l) {
                if (i + 1 < res[l]) continue;
                float disc; const float d = dist(own.c, own.r, eye, dir[l], &disc);
                if (disc >= 0) ocand = true;
                if (!(d >= best[l])) { best[l] = d; bi[l] = (int)(i + 1); }
            }
            if (ocand) ++pB2; else ++pB1;
            i += 2;
        }
        // shadow
        V3 sp[64]; bool need[64]; unsigned nn = 0;
        for (unsigned l = 0; l < 64; ++l) { need[l] = false; if (!in[l] || best[l] == INFINITY) continue; const Node &it = nodes[bi[l]]; const V3 nrm = normalized(add(eye, sub(mulf(dir[l], best[l]), it.c))); if (dot(nrm, light) >= 0) continue; sp[l] = add(add(eye, mulf(dir[l], best[l])), mulf(nrm, best[l] * sqrtf(1.1920929e-7f))); need[l] = true; ++nn; }
        if (!nn) continue;
        ++swaves;
        for (unsigned l = 0; l < 64; ++l) res[l] = need[l] ? 0u : 0xFFFFFFFFu;
        for (size_t i = 0; i < n;) {
            const Node &nd = nodes[i];
            bool cand = false, enter = false, fin = false;
            for (unsigned l = 0; l < 64; ++l) {
                if (i < res[l]) continue;
                float disc; const bool hit = dist(nd.c, nd.r, sp[l], sdir, &disc) < INFINITY;
                if (disc >= 0) cand = true;
                if (!nd.item) { if (!hit) res[l] = nd.skip; else enter = true; }
                else if (hit) { res[l] = 0xFFFFFFFFu; fin = true; }
            }
            if (cand) ++sH; else ++sQ;
            size_t ni = (nd.item || enter) ? i + 1 : nd.skip;
            if (fin) { uint32_t m = 0xFFFFFFFFu; for (unsigned l = 0; l < 64; ++l) m = std::min(m, res[l] == 0xFFFFFFFFu ? 0xFFFFFFFFu : std::max<uint32_t>(res[l], (uint32_t)i + 1)); ni = m == 0xFFFFFFFFu ? n : m; }
            i = ni;
        }
    }
    printf("%ux%u: %llu waves (%llu with shadow rays)\n", W, H, (unsigned long long)waves, (unsigned long long)swaves);
    printf("primary: quiet BOUND %llu, quiet ITEM %llu, BOUND cand/no enter %llu, BOUND enter/own quiet %llu, BOUND enter/own cand %llu, ITEM cand %llu\n", (unsigned long long)pQ, (unsigned long long)pLeafQ,
           (unsigned long long)pB0, (unsigned long long)pB1, (unsigned long long)pB2, (unsigned long long)pI);
    const double valu = 5.0 * (pQ + pLeafQ) + 25.0 * pB0 + 28.0 * pB1 + 42.0 * pB2 + 27.0 * pI;
    printf("primary VALU estimate: %.2f M (quiet %.2f, B0 %.2f, B1 %.2f, B2 %.2f, I %.2f)\n", valu / 1e6, 5.0 * (pQ + pLeafQ) / 1e6, 25.0 * pB0 / 1e6, 28.0 * pB1 / 1e6, 42.0 * pB2 / 1e6, 27.0 * pI / 1e6);
    printf("BOUND-cand steps decided by A/N alone: %llu (of which no-enter %llu), by A/N/G: %llu, of %llu\n", (unsigned long long)dec1, (unsigned long long)dec1_b0, (unsigned long long)dec2, (unsigned long long)(pB0 + pB1 + pB2));
    printf("shadow: quiet %llu, hit %llu -> VALU estimate %.2f M (6 / ~14)\n", (unsigned long long)sQ, (unsigned long long)sH, (6.0 * sQ + 14.0 * sH) / 1e6);
    return 0;
}
