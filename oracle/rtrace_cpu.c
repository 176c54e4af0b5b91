/*
 * rtrace_cpu.c -- TEST / BENCH INFRASTRUCTURE ONLY: the CPU restatement (oracle/rt_oracle.c) as a process, so that bench.py can put the
 * wall time of `make image` on this host's cores (process start -> file closed; /root/reference/Makefile:6-7 `time ./target/release/rtrace
 * --samples-per-pixel=4 --width=1024 --height=768 out.tga`) beside the MI355X backend's.  Same defaults as main.rs:22-90: width / height
 * 1024, one sample, RTRACEMAXPROCS threads (default 1), the default scene.  Never part of the product.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rt_oracle.h"

static int flag(const char *a, const char *name, const char **val, int *i, int argc, char **argv)
{
    const size_t n = strlen(name);
    if (strncmp(a, name, n) != 0) return 0;
    if (a[n] == '=') { *val = a + n + 1; return 1; }
    if (a[n] == '\0' && *i + 1 < argc) { *val = argv[++*i]; return 1; }
    return 0;
}

int main(int argc, char **argv)
{
    unsigned w = 1024, h = 1024, spp = 1, threads = 1, level = 8;
    const char *out = NULL, *v = NULL;
    if (getenv("RTRACEMAXPROCS")) { const long t = atol(getenv("RTRACEMAXPROCS")); if (t > 0) threads = (unsigned)t; }
    for (int i = 1; i < argc; ++i) {
        if (flag(argv[i], "--width", &v, &i, argc, argv)) w = (unsigned)atoi(v);
        else if (flag(argv[i], "--height", &v, &i, argc, argv)) h = (unsigned)atoi(v);
        else if (flag(argv[i], "--samples-per-pixel", &v, &i, argc, argv)) spp = (unsigned)atoi(v);
        else if (flag(argv[i], "--num-cores", &v, &i, argc, argv)) { if (atoi(v) > 1) threads = (unsigned)atoi(v); }
        else if (flag(argv[i], "--level", &v, &i, argc, argv)) level = (unsigned)atoi(v);
        else out = argv[i];
    }
    if (!out || !w || !h || !spp) { fprintf(stderr, "usage: rtrace_cpu [--width X] [--height Y] [--samples-per-pixel S] <output.tga>\n"); return 1; }
    orc_scene *s = orc_scene_default(ORC_F32, level);
    uint8_t *frame = (uint8_t *)calloc((size_t)w * h, 4);
    if (!s || !frame) return 2;
    orc_render(s, ORC_MODE_HIERARCHY, w, h, spp, threads, frame, NULL);
    const int rc = orc_write_ppm(out, frame, w, h, 1);
    free(frame);
    orc_scene_free(s);
    return rc == 0 ? 0 : 3;
}
