/* Compiled as plain C99 against include/rtrace_hip.h and linked with librtrace_hip.so: the boundary is a C ABI, usable
 * without a C++ compiler or HIP headers.  Prints the struct layout the foreign-language bindings (Rust repr(C) block in
 * INTEGRATION.md, ctypes in capi.py) rely on, and exercises the entry points that need no GPU. */
#include <stddef.h>
#include <stdio.h>
#include <string.h>

#include "rtrace_hip.h"

int main(void)
{
    rt_region tiles[2] = { { 0, 64, 64, 0 }, { 64, 24, 96, 0 } };
    rt_options o = { 1920, 1080, 1 };
    int n = -1;
    rt_status st = rt_device_count(&n);
    printf("abi %d\n", rt_abi_version());
    printf("sizeof rt_options %zu rt_region %zu rt_range %zu rt_stats %zu\n", sizeof(rt_options), sizeof(rt_region),
           sizeof(rt_range), sizeof(rt_stats));
    printf("offsets stats: primary %zu hits %zu shadow %zu occluded %zu sphere_tests %zu bound_tests %zu tests_executed %zu primary_tests %zu device_ms %zu longest_wave_cycles %zu longest_wave_ref100mhz %zu\n",
           offsetof(rt_stats, primary), offsetof(rt_stats, hits), offsetof(rt_stats, shadow), offsetof(rt_stats, occluded),
           offsetof(rt_stats, sphere_tests), offsetof(rt_stats, bound_tests), offsetof(rt_stats, tests_executed),
           offsetof(rt_stats, primary_tests), offsetof(rt_stats, device_ms), offsetof(rt_stats, longest_wave_cycles), offsetof(rt_stats, longest_wave_ref100mhz));
    printf("built with: %s; last launch flags %u\n", rt_build_info(), (unsigned)rt_last_launch_flags());
    printf("bytes %llu\n", (unsigned long long)rt_tiles_rgba_bytes(tiles, 2));
    printf("devices status %d n %d (%s)\n", (int)st, n, rt_strerror(st));
    /* argument validation happens before any device is touched */
    {
        rt_scene *s = (rt_scene *)0;
        float v[3] = { 0, 0, 0 };
        rt_status bad = rt_scene_create(0, RT_F32, (const void *)0, 1, v, v, (const void *)0, (const rt_range *)0, 0, &s);
        printf("null items -> %d (%s)\n", (int)bad, rt_strerror(bad));
        if (bad != RT_ERR_INVALID_ARGUMENT || s != (rt_scene *)0) return 1;
    }
    (void)o;
    return (rt_abi_version() == RTRACE_HIP_ABI_VERSION && rt_tiles_rgba_bytes(tiles, 2) == (64u * 64u + 32u * 24u) * 4u) ? 0 : 1;
}
