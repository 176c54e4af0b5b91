// tools/scalar_contention_probe.hip -- how the scalar data cache serves MANY waves: every wave of a launch chases its own way through one small
// buffer that stays in the scalar cache (8 KB, 64 B records, a random cycle; each wave starts somewhere else) with dependent s_load_dwordxN,
// and reports its cycles per step; the launch puts 1 .. 32 waves on every CU.  If a dependent load costs ~115 cycles for a lone wave and far
// more with 32 waves per CU, the traversal loops' waiting is the cache's REQUEST THROUGHPUT, not its latency.
//   hipcc -O2 --offload-arch=gfx950 -o scalar_contention_probe tools/scalar_contention_probe.hip && ./scalar_contention_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <numeric>
#include <random>
#include <vector>

template <int DW>
__global__ void k_chase(const unsigned *base, unsigned n_rec, unsigned steps, unsigned long long *out)
{
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    unsigned off = __builtin_amdgcn_readfirstlane((wave * 97u) % n_rec) * 64u;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0));
    for (unsigned i = 0; i < steps; ++i) {
        if (DW == 8) asm volatile("s_load_dwordx8 s[36:43], %1, %0\n s_waitcnt lgkmcnt(0)\n s_mov_b32 %0, s36\n" : "+s"(off) : "s"(base) : "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "memory");
        else if (DW == 4) asm volatile("s_load_dwordx4 s[36:39], %1, %0\n s_waitcnt lgkmcnt(0)\n s_mov_b32 %0, s36\n" : "+s"(off) : "s"(base) : "s36", "s37", "s38", "s39", "memory");
        else if (DW == 16) asm volatile("s_load_dwordx16 s[36:51], %1, %0\n s_waitcnt lgkmcnt(0)\n s_mov_b32 %0, s36\n" : "+s"(off) : "s"(base) : "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "memory");
        else asm volatile("s_load_dword s36, %1, %0\n s_waitcnt lgkmcnt(0)\n s_mov_b32 %0, s36\n" : "+s"(off) : "s"(base) : "s36", "memory");
    }
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1));
    if ((threadIdx.x & 63) == 0) out[wave] = t1 - t0;
    if (off == 0xFFFFFFFFu) out[0] = 0;
}

template <int DW>
static void run(const unsigned *d, unsigned n_rec, unsigned long long *d_out, int waves_per_cu, bool first)
{
    const unsigned steps = 4000;
    // 256 CUs; workgroups of 4 waves (one per SIMD); waves_per_cu / 4 workgroups per CU when >= 4, else smaller workgroups
    const int wg_waves = std::min(waves_per_cu, 4), wgs = 256 * std::max(1, waves_per_cu / 4);
    const int total_waves = wgs * wg_waves;
    std::vector<unsigned long long> h(total_waves);
    double best = 1e30, bestmax = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL((k_chase<DW>), dim3(wgs), dim3(64 * wg_waves), 0, nullptr, d, n_rec, steps, d_out);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d_out, sizeof(unsigned long long) * total_waves, hipMemcpyDeviceToHost);
        double sum = 0, mx = 0;
        for (auto v : h) { sum += (double)v; mx = std::max(mx, (double)v); }
        if (sum / total_waves / steps < best) { best = sum / total_waves / steps; bestmax = mx / steps; }
    }
    printf("%s  {\"load\": \"s_load_dwordx%d\", \"waves_per_cu\": %d, \"cycles_per_step_mean\": %.1f, \"cycles_per_step_slowest_wave\": %.1f, \"requests_per_cycle_per_cu\": %.4f}",
           first ? "" : ",\n", DW, waves_per_cu, best, bestmax, waves_per_cu / best);
}

int main()
{
    const unsigned bytes = 8192, n_rec = bytes / 64;
    std::vector<unsigned> perm(n_rec);
    std::iota(perm.begin(), perm.end(), 0u);
    std::mt19937 rng(7);
    std::shuffle(perm.begin(), perm.end(), rng);
    std::vector<unsigned> buf(bytes / 4, 0u);
    for (unsigned i = 0; i < n_rec; ++i) buf[(size_t)perm[i] * 16] = perm[(i + 1) % n_rec] * 64u;
    unsigned *d = nullptr;
    unsigned long long *d_out = nullptr;
    hipMalloc(&d, bytes);
    hipMemcpy(d, buf.data(), bytes, hipMemcpyHostToDevice);
    hipMalloc(&d_out, sizeof(unsigned long long) * 256 * 32);
    printf("{\"probe\": \"dependent scalar loads from an 8 KB buffer, every wave its own chain, N waves per CU on all 256 CUs\", \"results\": [\n");
    bool first = true;
    for (int w : { 1, 2, 4, 8, 16, 32 }) { run<8>(d, n_rec, d_out, w, first); first = false; }
    for (int w : { 1, 8, 32 }) run<4>(d, n_rec, d_out, w, false);
    for (int w : { 1, 8, 32 }) run<1>(d, n_rec, d_out, w, false);
    for (int w : { 1, 8, 32 }) run<16>(d, n_rec, d_out, w, false);
    printf("\n]}\n");
    return 0;
}
