// fuzz_images.cpp — runs every file named on the command line through the host loader: *.glb through load_glb,
// anything else through the image path (PNG / JPEG decode + texture baker); reports how many were accepted. Built with -fsanitize=address,undefined by
// tests/test_host_sanitizers.py and fed mutated images: a malformed texture must end in an exception, never in a
// memory error.
#include <cstdint>
#include <cstdio>
#include <exception>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

#include "jpeg.h"
#include "scene_loader.h"

int main(int argc, char** argv) {
    int ok = 0, rejected = 0;
    std::vector<uint8_t> layer((size_t)RT_TEX_SIZE * RT_TEX_SIZE * 4);
    for (int i = 1; i < argc; ++i) {
        const std::string path = argv[i];
        if (path.size() > 4 && path.compare(path.size() - 4, 4, ".glb") == 0) { // whole loader: container, JSON, accessors, textures
            try {
                const rthost::LoadedScene sc = rthost::load_glb(path, false);
                const rt_scene_desc d = sc.desc();
                if (d.n_triangles && (!d.positions || !d.indices)) return 4;
                ++ok;
            } catch (const std::exception&) {
                ++rejected;
            }
            continue;
        }
        std::ifstream f(argv[i], std::ios::binary);
        std::vector<uint8_t> bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        try {
            uint32_t w = 0, h = 0;
            std::vector<uint8_t> rgba;
            rthost::decode_image_rgba8(bytes.data(), bytes.size(), w, h, rgba);
            if (rgba.size() != (size_t)w * h * 4) return 3;
            if ((uint64_t)w * h <= (1u << 22)) rthost::resize_to_layer(rgba.data(), w, h, layer.data());
            ++ok;
        } catch (const std::exception&) {
            ++rejected;
        }
    }
    std::printf("decoded %d rejected %d\n", ok, rejected);
    return 0;
}
