// one-time costs of the first HIP calls of a process, in the order rt_scene_create makes them
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_small(float *p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 2.0f + 1.0f; }
#define LAP(what) do { double t = now(); printf("%-44s %8.3f ms\n", what, t - t0); t0 = t; } while (0)
int main(int argc, char **argv)
{
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    double t0 = now();
    int n = 0; hipGetDeviceCount(&n); LAP("hipGetDeviceCount (runtime init)");
    hipSetDevice(0); LAP("hipSetDevice");
    hipStream_t s = nullptr;
    if (mode & 1) { s = nullptr; } else { hipStreamCreateWithFlags(&s, hipStreamNonBlocking); LAP("hipStreamCreateWithFlags"); }
    float *d = nullptr; hipMalloc(&d, 350000); LAP("hipMalloc 350 KB (first)");
    float *d2 = nullptr; hipMalloc(&d2, 900000); LAP("hipMalloc 900 KB (second)");
    std::vector<float> h(87500, 1.0f);
    float *hp = nullptr;
    if (mode & 8) {
        hipHostMalloc(&hp, 350000, hipHostMallocDefault); memcpy(hp, h.data(), 350000);
        float *alias = nullptr; hipHostGetDevicePointer((void **)&alias, hp, 0); LAP("hipHostMalloc + alias");
        hipLaunchKernelGGL(k_small, dim3(342), dim3(256), 0, s, alias, 87500); hipStreamSynchronize(s); LAP("first kernel (pinned) + sync");
        hipLaunchKernelGGL(k_small, dim3(342), dim3(256), 0, s, d, 87500); hipStreamSynchronize(s); LAP("kernel (device) + sync");
        float *hc2 = nullptr; hipHostMalloc(&hc2, 4000000, hipHostMallocDefault); LAP("hipHostMalloc 4 MB");
        hipMemcpyAsync(hc2, d, 262144, hipMemcpyDeviceToHost, s); LAP("first D2H async to pinned (no H2D before)"); hipStreamSynchronize(s); LAP("sync");
        std::vector<float> pg(87500);
        hipMemcpyAsync(pg.data(), d, 262144, hipMemcpyDeviceToHost, s); LAP("D2H async to pageable"); hipStreamSynchronize(s); LAP("sync");
        hipMemcpy(pg.data(), d, 262144, hipMemcpyDeviceToHost); LAP("D2H blocking to pageable");
        hipMemcpyAsync(d, hp, 350000, hipMemcpyHostToDevice, s); LAP("first H2D async (after D2H)"); hipStreamSynchronize(s); LAP("sync");
        return 0;
    }
    if (mode & 4) {
        hipHostMalloc(&hp, 350000, hipHostMallocDefault); LAP("hipHostMalloc 350 KB"); memcpy(hp, h.data(), 350000);
        float *alias = nullptr; hipHostGetDevicePointer((void **)&alias, hp, 0); LAP("hipHostGetDevicePointer");
        hipLaunchKernelGGL(k_small, dim3(342), dim3(256), 0, s, alias, 87500); LAP("first kernel launch, reads/writes pinned host memory");
        hipStreamSynchronize(s); LAP("sync");
        hipLaunchKernelGGL(k_small, dim3(342), dim3(256), 0, s, d, 87500); hipStreamSynchronize(s); LAP("second launch (device memory) + sync");
        hipMemcpyAsync(d, hp, 350000, hipMemcpyHostToDevice, s); LAP("hipMemcpyAsync from pinned (first copy, after kernels)"); hipStreamSynchronize(s); LAP("sync");
        float *hc2 = nullptr; hipHostMalloc(&hc2, 600000, hipHostMallocDefault);
        hipMemcpyAsync(hc2, d, 262144, hipMemcpyDeviceToHost, s); LAP("first D2H async"); hipStreamSynchronize(s); LAP("sync");
        return 0;
    }
    if (mode & 2) { hipHostMalloc(&hp, 350000, hipHostMallocDefault); LAP("hipHostMalloc 350 KB"); memcpy(hp, h.data(), 350000); hipMemcpyAsync(d, hp, 350000, hipMemcpyHostToDevice, s); LAP("hipMemcpyAsync from pinned"); }
    else { hipMemcpyAsync(d, h.data(), 350000, hipMemcpyHostToDevice, s); LAP("hipMemcpyAsync from pageable (first)"); }
    hipStreamSynchronize(s); LAP("hipStreamSynchronize");
    hipMemcpyAsync(d2, h.data(), 350000, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); LAP("second pageable copy + sync");
    hipLaunchKernelGGL(k_small, dim3(342), dim3(256), 0, s, d, 87500); LAP("first kernel launch (this tiny module)");
    hipStreamSynchronize(s); LAP("sync");
    hipLaunchKernelGGL(k_small, dim3(342), dim3(256), 0, s, d, 87500); hipStreamSynchronize(s); LAP("second launch + sync");
    hipEvent_t e; hipEventCreate(&e); LAP("hipEventCreate");
    float *hc = nullptr; hipHostMalloc(&hc, 600000, hipHostMallocDefault); LAP("hipHostMalloc 600 KB");
    hipMemcpyAsync(hc, d, 262144, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); LAP("D2H to pinned + sync");
    return 0;
}
