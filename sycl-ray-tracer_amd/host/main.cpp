// main.cpp — the `raytracer` CLI with the reference's flags (src/main.cpp:8-77) on the MI355X path.
//
//   -d,--max-depth N       (10)       -s,--sample-count N  (32)      scene_path (./assets/sponza.glb)
//   -w,--wavefront         -m,--megakernel        neither => wavefront
// Extensions (the reference hard-codes 1920x1080, one device, out.png):
//   --width N --height N   --device N   --devices A,B,..   --out FILE   --bvh {sah,lbvh}   --rr N   --quiet
//   --schedule {default,per-sample,per-bounce,per-bounce-fused}   which of the wavefront renderer's schedules renders the frame
//                          (rt_renderer_set_schedule; per-bounce = the reference's own: src/render_wavefront.cpp:396-417)
// Prints the same lines as the reference (Loading scene, loader chatter, Sample n, the three statistics
// lines benchmark.py scrapes, Writing image to disk) and writes the PNG.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "renderer.h"

static void usage(const char* argv0) {
    std::printf("Usage: %s [OPTIONS] [scene_path]\n\n"
                "Options:\n"
                "  -h,--help                   Print this help message and exit\n"
                "  -d,--max-depth UINT         Max depth\n"
                "  -s,--sample-count UINT      Sample count\n"
                "  -w,--wavefront              Use wavefront renderer\n"
                "  -m,--megakernel             Use megakernel renderer\n"
                "  --width UINT --height UINT  Image size (default 1920x1080)\n"
                "  --device INT                HIP device (default 0)\n"
                "  --devices A,B,...           tile the frame over these HIP devices (8-row strips, one host thread each)\n"
                "  --out FILE                  Output PNG (default out.png)\n"
                "  --bvh sah|lbvh              BVH builder (default sah)\n"
                "  --rr UINT                   Russian roulette from this bounce on (default 0 = off, as the reference)\n"
                "  --schedule NAME             wavefront schedule: default (one launch per frame), per-sample, per-bounce (a launch pair per\n"
                "                              bounce with compaction in between, the reference's), per-bounce-fused (one kernel per bounce)\n"
                "  --quiet                     No loader chatter\n"
                "\nThe camera must lie within 100 scene scales of the scene's bounds (scale = largest extent or coordinate): farther out the\n"
                "conservative box culling of the closest-hit query no longer holds and the frame is refused with an error, not rendered wrong.\n",
                argv0);
}

int main(int argc, const char* argv[]) {
    // HIP serves a process's streams from GPU_MAX_HW_QUEUES hardware queues, 4 by default: the per-bounce schedules' three stream lanes plus the
    // frame's stream want more, and the library (which never changes the environment) would otherwise resolve them down to two lanes. This is
    // the HOST's setting, made here before the first HIP call of the process (rt_stats.hw_queues reports what the library saw).
    setenv("GPU_MAX_HW_QUEUES", "8", 0);
    uint32_t max_depth = 10, sample_count = 32, rr = 0;
    std::string scene_path = "./assets/sponza.glb", out_path = "out.png";
    bool use_wavefront = false, use_megakernel = false, quiet = false;
    int32_t width = 1920, height = 1080;
    int device = 0, bvh = RT_BVH_DEFAULT;
    std::vector<int> devices;
    bool have_scene = false;
    rt_schedule schedule{0u, 0u, 0u, -1, 0u, 0u, -1, 0u, 0u, -1}; // the library's default
    bool schedule_given = false;

    auto need = [&](int& i) -> const char* {
        if (i + 1 >= argc) {
            std::fprintf(stderr, "%s: 1 required TEXT missing\n", argv[i]);
            std::exit(106);
        }
        return argv[++i];
    };
    auto to_u32 = [&](const char* flag, const char* v) -> uint32_t {
        char* end = nullptr;
        unsigned long x = std::strtoul(v, &end, 10);
        if (!*v || *end || v[0] == '-') {
            std::fprintf(stderr, "Could not convert: %s = %s\n", flag, v);
            std::exit(104);
        }
        return (uint32_t)x;
    };
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "-h" || a == "--help") { usage(argv[0]); return 0; }
        else if (a == "-d" || a == "--max-depth") max_depth = to_u32("--max-depth", need(i));
        else if (a.rfind("--max-depth=", 0) == 0) max_depth = to_u32("--max-depth", a.c_str() + 12);
        else if (a == "-s" || a == "--sample-count") sample_count = to_u32("--sample-count", need(i));
        else if (a.rfind("--sample-count=", 0) == 0) sample_count = to_u32("--sample-count", a.c_str() + 15);
        else if (a == "-w" || a == "--wavefront") use_wavefront = true;
        else if (a == "-m" || a == "--megakernel") use_megakernel = true;
        else if (a == "--width") width = (int32_t)to_u32("--width", need(i));
        else if (a == "--height") height = (int32_t)to_u32("--height", need(i));
        else if (a == "--device") device = (int)to_u32("--device", need(i));
        else if (a == "--devices") {
            const std::string v = need(i);
            size_t pos = 0;
            while (pos <= v.size()) {
                const size_t comma = v.find(',', pos);
                const std::string tok = v.substr(pos, comma == std::string::npos ? std::string::npos : comma - pos);
                devices.push_back((int)to_u32("--devices", tok.c_str()));
                if (comma == std::string::npos) break;
                pos = comma + 1;
            }
        }
        else if (a == "--out") out_path = need(i);
        else if (a == "--rr") rr = to_u32("--rr", need(i));
        else if (a == "--quiet") quiet = true;
        else if (a == "--schedule") {
            const std::string v = need(i);
            schedule_given = true;
            if (v == "default") schedule_given = false;
            else if (v == "per-sample") schedule.samples_per_launch = 1;
            else if (v == "per-bounce") schedule.finish_depth = RT_SCHED_ALL_BOUNCES;
            else if (v == "per-bounce-fused") schedule.finish_depth = RT_SCHED_ALL_BOUNCES, schedule.fused_bounce = 1;
            else { std::fprintf(stderr, "--schedule: expected default, per-sample, per-bounce or per-bounce-fused\n"); return 105; }
        }
        else if (a == "--bvh") {
            const std::string v = need(i);
            if (v == "sah") bvh = RT_BVH_SAH;
            else if (v == "lbvh") bvh = RT_BVH_LBVH;
            else { std::fprintf(stderr, "--bvh: expected sah or lbvh\n"); return 105; }
        } else if (!a.empty() && a[0] == '-') {
            std::fprintf(stderr, "The following argument was not expected: %s\nRun with --help for more information.\n", a.c_str());
            return 109;
        } else if (!have_scene) {
            scene_path = a;
            have_scene = true;
        } else {
            std::fprintf(stderr, "The following argument was not expected: %s\n", a.c_str());
            return 109;
        }
    }
    if (!use_wavefront && !use_megakernel) use_wavefront = true; // src/main.cpp:26-28

    std::printf("Loading scene: %s\n", scene_path.c_str());
    try {
        const int n_dev = rt_device_count();
        if (n_dev <= 0) throw std::runtime_error(std::string("no HIP device: ") + rt_last_error());
        for (int d : devices)
            if (d >= n_dev) throw std::runtime_error("--devices: HIP device " + std::to_string(d) + " does not exist");
        if (!devices.empty()) device = devices[0];
        if (devices.size() > 1) {
            std::printf("Running on devices:");
            for (int d : devices) std::printf(" %d", d);
            std::printf(" of %d (gfx950 path, frame tiled in 8-row strips)\n", n_dev);
        } else
            std::printf("Running on device: HIP device %d of %d (gfx950 path)\n", device, n_dev); // src/app.hpp:51-54
        std::vector<uint8_t> image_buf((size_t)width * (size_t)height * 4);
        raytracer::Scene scene(scene_path, device, bvh, !quiet);
        raytracer::Camera camera({width, height}, scene.camera_position, scene.camera_direction, scene.camera_focal_length);
        std::unique_ptr<raytracer::IRenderer> renderer;
        if (use_megakernel) {
            auto* r = new raytracer::MegakernelRenderer({width, height}, image_buf.data(), max_depth, sample_count);
            r->out_path = out_path;
            r->russian_roulette = rr;
            r->devices = devices;
            renderer.reset(r);
        } else {
            auto* r = new raytracer::WavefrontRenderer({width, height}, image_buf.data(), max_depth, sample_count);
            r->out_path = out_path;
            r->russian_roulette = rr;
            r->devices = devices;
            if (schedule_given) r->schedule = schedule, r->has_schedule = true;
            renderer.reset(r);
        }
        renderer->render_frame(camera, scene);
    } catch (const std::exception& e) {
        std::printf("Caught exception: %s\n", e.what());
        return 1;
    }
    return 0;
}
