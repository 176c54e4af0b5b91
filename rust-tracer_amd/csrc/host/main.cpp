// main.cpp -- the `rtrace` binary: same flags, environment knob, defaults, quirks and exit behaviour as
// /root/reference/src/rust/main.rs:22-90 (clap 2.19 App "rtrace" 0.2.0), with the per-tile work done by the HIP
// backend.  Flags that do not exist in the reference (--device, --devices, --traversal, --level, --stats) only
// add GPU selection; none of the reference's flags changes meaning.
#include <cerrno>
#include <chrono>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <vector>

#include "render.hpp"
#ifdef RT_TEST_HOOKS
#include "../rt_debug.h"      // tests/c/rtrace_test only: the shipped rtrace binds nothing but include/rtrace_hip.h
#endif

using namespace rtrace;

namespace {

[[noreturn]] void panic(const std::string &msg)
{
    // `unwrap()` / `expect()` / `assert!` in the reference abort the process with exit status 101
    fprintf(stderr, "thread 'main' panicked at '%s'\n", msg.c_str());
    exit(101);
}

const char *USAGE =
    "rtrace 0.2.0\n"
    "Sebastian Thiel <byronimo@mail.com>\n"
    "A toy-raytracer for rendering a scene with spheres\n\n"
    "USAGE:\n"
    "    rtrace [OPTIONS] <output>\n\n"
    "OPTIONS:\n"
    "        --width <X>                     The width of the output image [default: 1024]\n"
    "        --height <Y>                    The height of the output image [default: 1024]\n"
    "        --samples-per-pixel <SAMPLES>   Amount of samples per pixel. 4 means 16 over-samples [default: 1]\n"
    "        --num-cores <numcores>          Amount of cores to do the rendering on [default: 1]\n"
    "                                        If this is not set, you may also use the RTRACEMAXPROCS\n"
    "                                        environment variable, e.g. RTRACEMAXPROCS=4.\n"
    "                                        The commandline always overrides environment variables.\n"
    "    -h, --help                          Prints help information\n"
    "    -V, --version                       Prints version information\n\n"
    "MI355X backend (not in the reference):\n"
    "        --device <N>                    first GPU to render on [default: 0]\n"
    "        --devices <N>                   number of GPUs; buckets are dealt round-robin and gathered over RCCL [default: 1]\n"
    "        --gather <rccl|host>            N > 1: one RCCL gather to the first GPU (default), or per-GPU host copies\n"
    "        --traversal <skip|flat>         hierarchy walk (default: the reference on every scene) or flat DFS scan (every item,\n"
    "                                        no culling: the same pixels when every bound encloses its group and the eye is outside)\n"
    "        --level <N>                     pyramid level of the default scene [default: 8]\n"
    "        --scene <file>                  render a sphere list instead (text: `cx cy cz r` per line, optional `light x y z` /\n"
    "                                        `eye x y z` lines; *.f32: raw f32 quadruples); a bounding-sphere hierarchy is built for it\n"
    "        --stats                         print ray counters and device time on stderr\n"
    "        --timings                       print where the process spent its wall time (one JSON object on stderr)\n"
    "        --strict-64                     panic like the reference unless width and height are multiples of 64\n\n"
    "ARGS:\n"
    "    <output>    Either a file with .tga extension, or - to write file to stdout\n";

template <typename T> T parse_or_panic(const std::string &s)
{
    // str::parse::<uN>().unwrap(): digits only (an optional leading '+'), value must fit
    const char *p = s.c_str();
    if (*p == '+') ++p;
    if (!*p) panic("called `Result::unwrap()` on an `Err` value: ParseIntError { kind: Empty }");
    unsigned long long v = 0;
    for (; *p; ++p) {
        if (*p < '0' || *p > '9') panic("called `Result::unwrap()` on an `Err` value: ParseIntError { kind: InvalidDigit }");
        v = v * 10 + (unsigned)(*p - '0');
        if (v > (unsigned long long)std::numeric_limits<T>::max())
            panic("called `Result::unwrap()` on an `Err` value: ParseIntError { kind: Overflow }");
    }
    return (T)v;
}

bool ends_with_tga_extension(const std::string &path)
{
    // Path::extension() == "tga": text after the last '.' of the file name, which must not start the name
    const size_t slash = path.find_last_of('/');
    const std::string name = slash == std::string::npos ? path : path.substr(slash + 1);
    const size_t dot = name.find_last_of('.');
    if (dot == std::string::npos || dot == 0) return false;
    return name.substr(dot + 1) == "tga";
}

std::string with_tga_extension(const std::string &path)
{
    const size_t slash = path.find_last_of('/');
    const size_t dot = path.find_last_of('.');
    if (dot == std::string::npos || (slash != std::string::npos && dot < slash) || dot == (slash == std::string::npos ? 0 : slash + 1))
        return path + ".tga";
    return path.substr(0, dot) + ".tga";
}

}  // namespace

int main(int argc, char **argv)
{
    using Clock = std::chrono::steady_clock;
    const Clock::time_point t_main = Clock::now();
    auto ms_since = [](Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); };
    // main.rs:24-29: RTRACEMAXPROCS, default 1, unparsable -> 1
    size_t nc_from_env = 1;
    if (const char *e = getenv("RTRACEMAXPROCS")) {
        char *end = nullptr;
        errno = 0;
        unsigned long long v = strtoull(e, &end, 10);
        if (*e && *e != '-' && end && *end == '\0' && errno == 0) nc_from_env = (size_t)v;
    }

    std::string width = "1024", height = "1024", ssp = "1", numcores = "1", output;
    std::string device = "0", devices = "1", traversal = "skip", level = "8", gather = "", scene_file = "", rccl_stand_in = "";
    bool have_output = false, stats = false, strict64 = false, timings = false;
    auto take = [&](int &i, const std::string &arg, const char *name, std::string &dst) -> bool {
        const std::string flag = std::string("--") + name;
        if (arg == flag) {
            if (i + 1 >= argc) { fprintf(stderr, "error: The argument '%s <value>' requires a value but none was supplied\n", flag.c_str()); exit(1); }
            dst = argv[++i];
            return true;
        }
        if (arg.compare(0, flag.size() + 1, flag + "=") == 0) { dst = arg.substr(flag.size() + 1); return true; }
        return false;
    };
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "-h" || a == "--help") { fputs(USAGE, stdout); return 0; }
        if (a == "-V" || a == "--version") { puts("rtrace 0.2.0"); return 0; }
        if (a == "--stats") { stats = true; continue; }
        if (a == "--strict-64") { strict64 = true; continue; }
        if (a == "--timings") { timings = true; continue; }
        if (take(i, a, "width", width) || take(i, a, "height", height) || take(i, a, "samples-per-pixel", ssp) ||
            take(i, a, "num-cores", numcores) || take(i, a, "device", device) || take(i, a, "devices", devices) ||
            take(i, a, "traversal", traversal) || take(i, a, "level", level) || take(i, a, "gather", gather) || take(i, a, "scene", scene_file))
            continue;
#ifdef RT_TEST_HOOKS
        // tests/c/rtrace_test (linked against the -DRT_TEST_HOOKS library): --rccl-stand-in <library> sends the gather through that library
        // instead of librccl.so with all N ranks on --device, so that the N > 1 path runs on a one-GPU box (tests/c/fake_rccl.cpp)
        if (take(i, a, "rccl-stand-in", rccl_stand_in)) continue;
#endif
        if (a.size() > 1 && a[0] == '-' && a != "-") {
            fprintf(stderr, "error: Found argument '%s' which wasn't expected, or isn't valid in this context\n\nFor more information try --help\n", a.c_str());
            return 1;
        }
        if (have_output) {
            fprintf(stderr, "error: Found argument '%s' which wasn't expected, or isn't valid in this context\n", a.c_str());
            return 1;
        }
        output = a;
        have_output = true;
    }
    if (!have_output || output.empty()) {                         // .required(true).empty_values(false)  main.rs:51-54
        fprintf(stderr, "error: The following required arguments were not provided:\n    <output>\n\nUSAGE:\n    rtrace [OPTIONS] <output>\n\nFor more information try --help\n");
        return 1;
    }

    // main.rs:56-61: the command line only wins when it asks for more than one core
    const size_t num_cores = parse_or_panic<size_t>(numcores);
    const size_t pool_size = num_cores > 1 ? num_cores : nc_from_env;
    if (pool_size == 0) panic("assertion failed: num_threads >= 1");            // ThreadPool::new(0) asserts

    // main.rs:63-76
    FileOrAnyWriter sink;
    if (output != "-") {
        if (!ends_with_tga_extension(output)) {
            printf("Output file '%s' must have the tga extension, e.g. %s\n", output.c_str(), with_tga_extension(output).c_str());
            return 0;                                             // `return;` from main: status 0, as in the reference
        }
        sink.f = fopen(output.c_str(), "wb");
        if (!sink.f) panic(std::string("called `Result::unwrap()` on an `Err` value: ") + strerror(errno));
        sink.is_file = true;
    } else {
        sink.f = stdout;
    }

    // main.rs:78-82 (u16 fields)
    const RenderOptions options{ parse_or_panic<uint16_t>(width), parse_or_panic<uint16_t>(height), parse_or_panic<uint16_t>(ssp) };

    int status = 0;
    try {
        // Arc::new(Default::default())  main.rs:23 -- or, not in the reference, a sphere list with an automatically built hierarchy
        const double t_args = ms_since(t_main);
        Clock::time_point t0 = Clock::now();
        const Scene scene = scene_file.empty() ? Scene::with_level(parse_or_panic<uint32_t>(level)) : Scene::from_file(scene_file);
        const double t_host_scene = ms_since(t0);
        double t_runtime = 0.0, t_context = 0.0;
        if (timings) {                                            // the first runtime call of the process: HIP initialisation on its own
            t0 = Clock::now();
            int n_visible = 0;
            (void)rt_device_count(&n_visible);
            t_runtime = ms_since(t0);
            // ... and the first call that needs the DEVICE (a pinned allocation of one page: the writer makes its own with the same call a
            // moment later).  What it does NOT bring about is the runtime's first hardware queue (~19 ms): that comes with the first stream
            // or the first launch of the process -- rt_scene_create's stream -- and is reported apart below (first_queue_ms, from
            // rt_scene_setup_cost; making and dropping a stream here instead would add 4 ms to the process).
            t0 = Clock::now();
            void *page = nullptr;
            if (rt_host_alloc(4096, &page) == RT_OK) (void)rt_host_free(page);
            t_context = ms_since(t0);
        }
        t0 = Clock::now();
        Backend be;
        be.strict_64 = strict64;
        be.want_stats = stats;
        if (traversal == "flat") be.traversal = RT_TRAVERSAL_FLAT;
        else if (traversal == "skip") be.traversal = RT_TRAVERSAL_SKIP;
        else { fprintf(stderr, "error: --traversal must be skip or flat\n"); return 1; }
        const int dev0 = parse_or_panic<int>(device), ndev = parse_or_panic<int>(devices);
        if (ndev < 1) { fprintf(stderr, "error: --devices must be >= 1\n"); return 1; }
        if (!gather.empty() && gather != "rccl" && gather != "host") { fprintf(stderr, "error: --gather must be rccl or host\n"); return 1; }
        // more than one GPU: the native gather (rt_gang: ncclCommInitAll + one ncclGather per frame); `--gather rccl` takes
        // that path with a single GPU too (a one-rank communicator), `--gather host` keeps the per-device host copies
        if (gather == "rccl" || (gather.empty() && ndev > 1)) {
#ifdef RT_TEST_HOOKS
            if (!rccl_stand_in.empty() && rt_debug_rccl_library(rccl_stand_in.c_str()) != RT_OK)
                throw std::runtime_error(std::string("--rccl-stand-in: ") + rt_last_error_message());
#endif
            std::vector<int> ids;
            for (int d = 0; d < ndev; ++d) ids.push_back(rccl_stand_in.empty() ? dev0 + d : dev0);
            rt_status gst = RT_OK;
            be.gang = DeviceGang::try_create(scene, ids, &gst);
            if (!be.gang && gst == RT_ERR_UNSUPPORTED && gather.empty()) {
                // no RCCL on this machine: the per-device host copies still work
                fprintf(stderr, "rtrace: RCCL is not available (%s); gathering the buckets through host memory instead\n", rt_last_error_message());
                for (int d = 0; d < ndev; ++d) be.devices.push_back(std::make_shared<DeviceScene>(scene, dev0 + d));
            } else if (!be.gang) {
                throw std::runtime_error(std::string("rt_gang_create: ") + rt_strerror(gst) + " -- " + rt_last_error_message());
            }
        } else {
            for (int d = 0; d < ndev; ++d) be.devices.push_back(std::make_shared<DeviceScene>(scene, dev0 + d));
        }

        const double t_device_scene = ms_since(t0);
        t0 = Clock::now();
        ThreadPool pool(pool_size);
        RenderStats st;
        double t_render = 0.0;
        {
            PPMStdoutRGBABufferWriter writer(true, sink);                                // main.rs:84-87
            st = Renderer::render(options, be, writer, pool);
            t_render = ms_since(t0);
            t0 = Clock::now();
        }                                                                                // Drop writes the final image
        const double t_drop_write = ms_since(t0);
        if (sink.is_file) { fclose(sink.f); sink.f = nullptr; } else fflush(stdout);
        if (timings) {
            double setup_ms = 0.0, queue_ms = 0.0;                // of the first device's rt_scene_create
            if (!be.devices.empty()) (void)rt_scene_setup_cost(be.devices[0]->handle(), &setup_ms, &queue_ms);
            fprintf(stderr, "{\"args_ms\": %.3f, \"host_scene_ms\": %.3f, \"runtime_init_ms\": %.3f, \"device_context_ms\": %.3f, \"device_scene_ms\": %.3f, \"first_queue_ms\": %.3f, "
                            "\"device_scene_own_ms\": %.3f, \"render_and_first_write_ms\": %.3f, \"drop_write_ms\": %.3f, \"main_to_here_ms\": %.3f}\n",
                    t_args, t_host_scene, t_runtime, t_context, t_device_scene, queue_ms, t_device_scene - queue_ms, t_render, t_drop_write, ms_since(t_main));
        }
        if (stats)
            fprintf(stderr, "primary %llu hits %llu shadow %llu occluded %llu item_tests %llu bound_tests %llu device_ms %.3f\n",
                    (unsigned long long)st.primary, (unsigned long long)st.hits, (unsigned long long)st.shadow,
                    (unsigned long long)st.occluded, (unsigned long long)st.sphere_tests, (unsigned long long)st.bound_tests, st.device_ms);
        // process::exit(0)  main.rs:89 -- like it, without unwinding main's locals: the image is complete and closed, and tearing the
        // device scene, its worker thread and the GPU runtime down in order costs a one-shot caller some 100 ms it has no use for
        // (bench.py make_image: process start -> exit is what `time make image` shows, /root/reference/Makefile:6-7)
        fflush(stderr);
        _exit(0);
    } catch (const std::exception &e) {
        fprintf(stderr, "thread 'main' panicked at '%s'\n", e.what());
        status = 101;
    }
    if (sink.is_file && sink.f) fclose(sink.f); else fflush(stdout);
    return status;                                                // process::exit(0)  main.rs:89
}
