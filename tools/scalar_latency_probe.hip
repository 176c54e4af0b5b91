// tools/scalar_latency_probe.hip -- what a DEPENDENT scalar load costs a wave (the traversal loops' step: the next node's address comes out of the
// record just fetched): one wave chases a random cycle of 64-byte-strided pointers through buffers of growing size with s_load_dwordx8 (the
// loops' own fetch), timed with s_memtime.  4 KB sits in the 16 KB scalar cache, 256 KB .. 2 MB in the XCD's L2, 64 MB beyond it.
//   hipcc -O2 --offload-arch=gfx950 -o scalar_latency_probe tools/scalar_latency_probe.hip && ./scalar_latency_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <numeric>
#include <random>
#include <vector>

__global__ void k_chase(const unsigned *base, unsigned start, unsigned steps, unsigned long long *out)
{
    unsigned off = start;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0));
    for (unsigned i = 0; i < steps; ++i) {
        // the record's first dword is the byte offset of the next record
        asm volatile("s_load_dwordx8 s[36:43], %1, %0\n s_waitcnt lgkmcnt(0)\n s_mov_b32 %0, s36\n"
                     : "+s"(off) : "s"(base) : "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "memory");
    }
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1));
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = off; }
}

int main()
{
    const size_t sizes[] = { 4u << 10, 12u << 10, 32u << 10, 256u << 10, 1u << 20, 2u << 20, 8u << 20, 64u << 20, 512u << 20 };
    unsigned long long *d_out = nullptr;
    hipMalloc(&d_out, 16);
    printf("{\"probe\": \"dependent s_load_dwordx8 chain, one wave, 64 B stride random cycle\", \"results\": [\n");
    bool first = true;
    for (size_t bytes : sizes) {
        const size_t n = bytes / 64;
        std::vector<unsigned> perm(n);
        std::iota(perm.begin(), perm.end(), 0u);
        std::mt19937 rng(12345);
        std::shuffle(perm.begin(), perm.end(), rng);
        std::vector<unsigned> buf(bytes / 4, 0u);
        for (size_t i = 0; i < n; ++i) buf[(size_t)perm[i] * 16] = perm[(i + 1) % n] * 64u;      // one cycle through every record
        unsigned *d = nullptr;
        hipMalloc(&d, bytes);
        hipMemcpy(d, buf.data(), bytes, hipMemcpyHostToDevice);
        const unsigned steps = (unsigned)std::min<size_t>(n * 4, 200000);
        unsigned long long h[2] = { 0, 0 };
        double best = 1e30;
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(k_chase, dim3(1), dim3(64), 0, nullptr, d, perm[0] * 64u, steps, d_out);
            hipDeviceSynchronize();
            hipMemcpy(h, d_out, 16, hipMemcpyDeviceToHost);
            best = std::min(best, (double)h[0] / steps);
        }
        // s_memtime counts at 100 MHz on gfx9 (REFCLK); report both raw ticks and ns
        printf("%s  {\"bytes\": %zu, \"steps\": %u, \"memtime_ticks_per_step\": %.3f, \"ns_per_step_at_100MHz\": %.1f}", first ? "" : ",\n", bytes, steps, best, best * 10.0);
        first = false;
        hipFree(d);
    }
    printf("\n]}\n");
    return 0;
}
