// tools/dispatch_rate_probe.hip -- what the workgroup dispatcher can do: a launch of W waves in workgroups of B threads whose every wave spins for
// a given time (0: returns at once) at the render kernels' register budget (8 waves per SIMD), timed with events.  Three questions:
//   * the launch RATE with empty waves (waves per us; per workgroup or per wave?),
//   * what a wave slot loses between two waves when the waves are short (the launch time against waves x spin / 8,192 slots),
//   * whether bigger workgroups change either.
//   hipcc -O2 --offload-arch=gfx950 -o dispatch_rate_probe tools/dispatch_rate_probe.hip && ./dispatch_rate_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

__global__ void __attribute__((amdgpu_waves_per_eu(8, 8))) k_spin(unsigned ticks, unsigned *sink)      // ticks of the 100 MHz clock
{
    if (ticks != 0u) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while ((unsigned)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) __builtin_amdgcn_s_sleep(1);
    }
    if (ticks == 0xFFFFFFFFu) sink[threadIdx.x] = 1;
}

static double time_launch(unsigned waves, unsigned wg_threads, unsigned ticks, unsigned *d_sink)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const unsigned wgs = waves * 64u / wg_threads;
    std::vector<float> t;
    for (int rep = 0; rep < 12; ++rep) {
        hipEventRecord(e0, nullptr);
        for (int k = 0; k < 10; ++k) hipLaunchKernelGGL(k_spin, dim3(wgs), dim3(wg_threads), 0, nullptr, ticks, d_sink);
        hipEventRecord(e1, nullptr);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2) t.push_back(ms * 100.0f);      // us per launch
    }
    std::sort(t.begin(), t.end());
    hipEventDestroy(e0); hipEventDestroy(e1);
    return t[t.size() / 2];
}

int main()
{
    unsigned *d_sink = nullptr;
    hipMalloc(&d_sink, 4096);
    printf("{\"device\": \"gfx950\", \"slots\": 8192, \"rows\": [\n");
    bool first = true;
    for (unsigned waves : { 8192u, 16384u, 32768u, 65536u })
        for (unsigned wg : { 64u, 256u, 512u, 1024u })
            for (unsigned ticks : { 0u, 100u, 300u, 500u }) {       // 0, 1, 3, 5 us
                const double us = time_launch(waves, wg, ticks, d_sink);
                const double ideal = ticks / 100.0 * std::max(1.0, waves / 8192.0);
                printf("%s  {\"waves\": %u, \"wg_threads\": %u, \"spin_us\": %.0f, \"launch_us\": %.2f, \"ideal_us\": %.1f, \"waves_per_us\": %.0f}", first ? "" : ",\n", waves, wg, ticks / 100.0, us, ideal,
                       waves / us);
                first = false;
                fflush(stdout);
            }
    printf("\n]}\n");
    return 0;
}
