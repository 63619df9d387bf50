// renderer.h — C++ host mirror of the reference's renderer plugin surface, over the C ABI of
// librt_mi355x.so. Same names and argument meaning as the reference; SYCL/Embree types are replaced by
// plain ones. Errors of the C ABI become std::runtime_error (main() turns them into a message + non-zero
// exit, where the reference calls std::terminate: src/main.cpp:71-74).
//
//   raytracer::Camera               == src/camera.hpp:65-106
//   raytracer::Scene                == src/scene.hpp:64-100 (loader in scene_loader.cpp)
//   raytracer::IRenderer            == src/render.hpp:11-18
//   raytracer::MegakernelRenderer   == src/render_megakernel.hpp:10-22, render_frame src/render_megakernel.cpp:75-187
//   raytracer::WavefrontRenderer    == src/render_wavefront.hpp:40-76,  render_frame src/render_wavefront.cpp:396-431
#pragma once
#include <array>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rt_mi355x.h"
#include "png.h"
#include "scene_loader.h"

namespace raytracer {

inline void rt_check(int status) {
    if (status != RT_OK) throw std::runtime_error(std::string("librt_mi355x: ") + rt_last_error());
}

struct Camera {
    rt_camera c{};
    // Camera(img_size, cam_center, cam_dir, focal_length): src/camera.hpp:74-106
    Camera(std::array<int32_t, 2> img_size, const float center[3], const float dir[3], float focal_length) {
        rt_check(rt_camera_init(&c, img_size[0], img_size[1], center, dir, focal_length));
    }
};

struct Scene {
    rthost::LoadedScene data;
    rt_scene* handle = nullptr;
    int device = 0, bvh_kind = RT_BVH_DEFAULT;
    mutable std::map<int, rt_scene*> replicas; // device -> scene replica (multi-GPU frames), built on first use
    float camera_position[3], camera_direction[3], camera_focal_length;

    // Scene(app, filepath): src/scene.cpp:54-129. `device` replaces App's SYCL device.
    Scene(const std::string& filepath, int device = 0, int bvh_kind = RT_BVH_DEFAULT, bool verbose = true)
        : data(rthost::load_glb(filepath, verbose)), device(device), bvh_kind(bvh_kind) {
        for (int k = 0; k < 3; ++k) camera_position[k] = data.camera_position[k], camera_direction[k] = data.camera_direction[k];
        camera_focal_length = data.camera_focal_length;
        const rt_scene_desc d = data.desc();
        rt_check(rt_scene_create(&d, device, bvh_kind, &handle));
    }
    Scene(const Scene&) = delete;
    Scene& operator=(const Scene&) = delete;
    ~Scene() {
        for (auto& kv : replicas) rt_scene_destroy(kv.second);
        rt_scene_destroy(handle);
    }
    // the scene on `dev`: the primary handle or a replica (every GPU of a tiled frame holds the whole scene + BVH)
    rt_scene* on_device(int dev) const {
        if (dev == device) return handle;
        auto it = replicas.find(dev);
        if (it != replicas.end()) return it->second;
        const rt_scene_desc d = data.desc();
        rt_scene* h = nullptr;
        rt_check(rt_scene_create(&d, dev, bvh_kind, &h));
        replicas[dev] = h;
        return h;
    }
};

struct IRenderer {
    virtual void render_frame(const Camera& camera, const Scene& scene) = 0;
    virtual ~IRenderer() {}
};

// Shared body of the two renderers: they differ only in the `kind` handed to the C ABI.
struct HipRendererBase : public IRenderer {
    std::array<int32_t, 2> img_size;
    uint8_t* image; // caller-owned RGBA8 buffer, W*H*4 (the reference's malloc_shared image: src/main.cpp:39-46)
    const uint32_t max_depth, sample_count;
    rt_renderer* handle = nullptr;
    const Scene* bound = nullptr;
    int kind;
    std::string out_path = "out.png"; // src/util.hpp:27
    uint32_t russian_roulette = 0;    // extension, 0 = off (rt_renderer_set_russian_roulette)
    // Extension: which of the wavefront renderer's schedules renders the frame (rt_renderer_set_schedule; same frame bit for bit). The
    // reference has one, a launch per bounce (src/render_wavefront.cpp:396-417): finish_depth = RT_SCHED_ALL_BOUNCES. Ignored by the megakernel.
    rt_schedule schedule{0u, 0u, 0u, -1, 0u, 0u, -1, 0u, 0u, -1};
    bool has_schedule = false;
    // Extension: more than one entry tiles the frame over these HIP devices in THIS process, one host thread per
    // tile (interleaved 8-row strips, tile k -> devices[k]; the same device may appear more than once). Every tile is
    // rendered into its renderer's own device buffer; rt_frame_gather then brings the strips to devices[0] with one
    // grouped ncclGather over xGMI and de-interleaves them there: the frame touches host memory once, for the PNG.
    // (The one-process-per-GPU form of the same gather is rtamd/dist.py + bench.py.)
    std::vector<int> devices;
    std::vector<rt_renderer*> tile_handles;
    rt_comm* comm = nullptr;
    rt_stats last{};

    HipRendererBase(int kind, std::array<int32_t, 2> img_size, uint8_t* image, uint32_t max_depth, uint32_t sample_count)
        : img_size(img_size), image(image), max_depth(max_depth), sample_count(sample_count), kind(kind) {}
    ~HipRendererBase() override {
        for (rt_renderer* h : tile_handles) rt_renderer_destroy(h);
        rt_comm_destroy(comm);
        rt_renderer_destroy(handle);
    }

    // one frame over devices.size() tiles: returns with `image` assembled and `last` = summed rays / wall time
    void render_tiled(const Camera& camera, const Scene& scene) {
        const uint32_t G = (uint32_t)devices.size();
        if (tile_handles.empty() || bound != &scene) {
            for (rt_renderer* h : tile_handles) rt_renderer_destroy(h);
            tile_handles.assign(G, nullptr);
            for (uint32_t k = 0; k < G; ++k) {
                rt_check(rt_renderer_create(kind, scene.on_device(devices[k]), img_size[0], img_size[1], max_depth, sample_count,
                                            RT_SEED_DEFAULT, &tile_handles[k]));
                rt_check(rt_renderer_set_tile(tile_handles[k], k, G, 8));
                if (has_schedule) rt_check(rt_renderer_set_schedule(tile_handles[k], &schedule));
            }
            bound = &scene;
        }
        if (!comm) {
            rt_check(rt_comm_create((int)G, devices.data(), &comm));
            std::printf("Tile gather: %s\n", rt_comm_uses_rccl(comm) ? "RCCL ncclGather over xGMI" : "device copies (a device is listed twice: no RCCL rank per tile)");
        }
        std::vector<rt_stats> st(G);
        std::vector<std::string> err(G);
        std::vector<std::thread> threads;
        const auto t0 = std::chrono::high_resolution_clock::now();
        for (uint32_t k = 0; k < G; ++k) {
            threads.emplace_back([&, k]() {
                rt_renderer* h = tile_handles[k];
                if (rt_renderer_set_russian_roulette(h, russian_roulette) != RT_OK ||
                    rt_render_frame_begin(h, &camera.c, nullptr, rt_renderer_tile_u8(h), nullptr) != RT_OK ||
                    rt_render_frame_end(h, &st[k]) != RT_OK)
                    err[k] = rt_last_error(); // rt_last_error is per thread
            });
        }
        for (auto& t : threads) t.join();
        for (uint32_t k = 0; k < G; ++k)
            if (!err[k].empty()) throw std::runtime_error("librt_mi355x (tile " + std::to_string(k) + "): " + err[k]);
        rt_check(rt_frame_gather(comm, tile_handles.data(), nullptr, image, 0, 0)); // strips -> root GPU -> full frame -> host, once
        const double wall = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
        last = rt_stats{};
        for (uint32_t k = 0; k < G; ++k) {
            last.rays += st[k].rays;
            last.launches += st[k].launches;
            last.device_ms = std::max(last.device_ms, st[k].device_ms);
        }
        last.seconds = wall;
    }

    void render_frame(const Camera& camera, const Scene& scene) override {
        if (devices.size() > 1) {
            if (kind == RT_RENDERER_WAVEFRONT)
                for (uint32_t s = 0; s < sample_count; ++s) std::printf("Sample %u\n", s);
            render_tiled(camera, scene);
            report_and_write(last.seconds); // tiles run concurrently: the frame time is the wall time of the slowest
            return;
        }
        if (!handle || bound != &scene) { // the ray queues belong to a scene's device: created on first use
            rt_renderer_destroy(handle);
            handle = nullptr;
            rt_check(rt_renderer_create(kind, scene.handle, img_size[0], img_size[1], max_depth, sample_count, RT_SEED_DEFAULT, &handle));
            if (has_schedule) rt_check(rt_renderer_set_schedule(handle, &schedule));
            bound = &scene;
        }
        rt_check(rt_renderer_set_russian_roulette(handle, russian_roulette));
        if (kind == RT_RENDERER_WAVEFRONT)
            for (uint32_t s = 0; s < sample_count; ++s) std::printf("Sample %u\n", s); // src/render_wavefront.cpp:402
        rt_check(rt_render_frame(handle, &camera.c, nullptr, image, &last));
        report_and_write(last.device_ms * 1e-3);
    }

    void report_and_write(double secs) {
        const double rays_per_sec = secs > 0 ? (double)last.rays / secs : 0.0;
        // the three lines benchmark.py scrapes (src/render_wavefront.cpp:425-427, benchmark.py:49-55)
        std::printf("Time measured: %.6f seconds\n", secs);
        std::printf("Total rays: %llu\n", (unsigned long long)last.rays);
        std::printf("Rays/sec: %.2fM\n", rays_per_sec / 1000000.0);
        if (has_schedule && devices.size() <= 1) // what ran (rt_stats): launches of the traversal kernels of the frame
            std::printf("Schedule: %u stream lanes, launches: extend %u, shade %u, shoot %u, finish %u\n", last.stream_lanes, last.launches_by_kernel[RT_K_WF_EXTEND],
                        last.launches_by_kernel[RT_K_WF_SHADE], last.launches_by_kernel[RT_K_WF_SHOOT], last.launches_by_kernel[RT_K_WF_FINISH]);
        std::printf("Writing image to disk\n");
        if (!rthost::write_png_rgba8(out_path.c_str(), (uint32_t)img_size[0], (uint32_t)img_size[1], image, (size_t)img_size[0] * 4)) {
            std::printf("Failed to write image to disk.\n"); // src/util.hpp:27-30
            throw std::runtime_error("cannot write " + out_path);
        }
    }
};

struct MegakernelRenderer : public HipRendererBase {
    MegakernelRenderer(std::array<int32_t, 2> img_size, uint8_t* image, uint32_t max_depth, uint32_t sample_count)
        : HipRendererBase(RT_RENDERER_MEGAKERNEL, img_size, image, max_depth, sample_count) {}
};

struct WavefrontRenderer : public HipRendererBase {
    WavefrontRenderer(std::array<int32_t, 2> img_size, uint8_t* image, uint32_t max_depth, uint32_t sample_count)
        : HipRendererBase(RT_RENDERER_WAVEFRONT, img_size, image, max_depth, sample_count) {}
};

} // namespace raytracer
