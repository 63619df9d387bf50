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
#include <cstdint>
#include <cstdio>
#include <memory>
#include <stdexcept>
#include <string>
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
    float camera_position[3], camera_direction[3], camera_focal_length;

    // Scene(app, filepath): src/scene.cpp:54-129. `device` replaces App's SYCL device.
    Scene(const std::string& filepath, int device = 0, int bvh_kind = RT_BVH_DEFAULT, bool verbose = true)
        : data(rthost::load_glb(filepath, verbose)) {
        for (int k = 0; k < 3; ++k) camera_position[k] = data.camera_position[k], camera_direction[k] = data.camera_direction[k];
        camera_focal_length = data.camera_focal_length;
        const rt_scene_desc d = data.desc();
        rt_check(rt_scene_create(&d, device, bvh_kind, &handle));
    }
    Scene(const Scene&) = delete;
    Scene& operator=(const Scene&) = delete;
    ~Scene() { rt_scene_destroy(handle); }
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
    rt_stats last{};

    HipRendererBase(int kind, std::array<int32_t, 2> img_size, uint8_t* image, uint32_t max_depth, uint32_t sample_count)
        : img_size(img_size), image(image), max_depth(max_depth), sample_count(sample_count), kind(kind) {}
    ~HipRendererBase() override { rt_renderer_destroy(handle); }

    void render_frame(const Camera& camera, const Scene& scene) override {
        if (!handle || bound != &scene) { // the ray queues belong to a scene's device: created on first use
            rt_renderer_destroy(handle);
            handle = nullptr;
            rt_check(rt_renderer_create(kind, scene.handle, img_size[0], img_size[1], max_depth, sample_count, RT_SEED_DEFAULT, &handle));
            bound = &scene;
        }
        rt_check(rt_renderer_set_russian_roulette(handle, russian_roulette));
        if (kind == RT_RENDERER_WAVEFRONT)
            for (uint32_t s = 0; s < sample_count; ++s) std::printf("Sample %u\n", s); // src/render_wavefront.cpp:402
        rt_check(rt_render_frame(handle, &camera.c, nullptr, image, &last));
        const double secs = last.device_ms * 1e-3;
        const double rays_per_sec = secs > 0 ? (double)last.rays / secs : 0.0;
        // the three lines benchmark.py scrapes (src/render_wavefront.cpp:425-427, benchmark.py:49-55)
        std::printf("Time measured: %.6f seconds\n", secs);
        std::printf("Total rays: %llu\n", (unsigned long long)last.rays);
        std::printf("Rays/sec: %.2fM\n", rays_per_sec / 1000000.0);
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
