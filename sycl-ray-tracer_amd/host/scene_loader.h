// scene_loader.h — GLB scene loader + texture baker of the host adapter.
//
// Mirrors raytracer::Scene (src/scene.hpp:64-100, src/scene.cpp:54-129,148-162,164-442,444-510) and
// ImageManager (src/image_manager.hpp:39-100) without SYCL, Embree, glm or tinygltf: the result is
// the flat rt_scene_desc the C ABI consumes. float32 operation order equals rtamd/scenes.py so both
// loaders produce identical bits for the same file.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/rt_mi355x.h"

namespace rthost {

struct LoadedScene {
    std::string name;
    std::vector<float> positions, normals, uvs;
    std::vector<uint32_t> indices, tri_instance;
    std::vector<rt_instance> instances;
    std::vector<rt_material> materials;
    std::vector<uint8_t> textures; // n_layers * 512*512*4
    uint32_t n_layers = 0;
    float sky[3] = {0.5f, 0.7f, 1.0f};       // Scene::sky_color default (src/scene.hpp:76)
    bool has_camera = false;
    float camera_position[3] = {0, 0, 0};
    float camera_direction[3] = {0, 0, -1};
    float camera_focal_length = 1.0f;

    rt_scene_desc desc() const; // pointers into this object
};

// Throws std::runtime_error (the reference throws from Scene::Scene too: src/scene.cpp:68-70).
LoadedScene load_glb(const std::string& path, bool verbose = false);

// Every glTF image becomes one 512x512 RGBA8 layer (ImageManager::upload_image, :39-74). Images that are
// already 512x512 are copied; others are resampled in linear light (area average down, bilinear up). The
// reference calls stbir_resize_uint8_srgb (third party, unpinned): results for non-512 images may differ
// in low bits.
void resize_to_layer(const uint8_t* rgba, uint32_t w, uint32_t h, uint8_t* out512);

} // namespace rthost
