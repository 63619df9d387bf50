/* ref_stb.c — builds the REFERENCE's own image decoder and resizer into oracle/_ref/libref_stb.so, so that the
 * tests can pin this repo's PNG/JPEG decoders and texture baker against exactly what the reference runs:
 *   tinygltf::LoadImageData -> stbi_load_from_memory(bytes, size, &w, &h, &comp, 4)   (deps/include/tiny_gltf.h:2603-2634)
 *   ImageManager::upload_image -> stbir_resize_uint8_srgb(..., 512, 512, 0, STBIR_RGBA)  (src/image_manager.hpp:52-62)
 *
 * TEST INFRASTRUCTURE ONLY. The two headers are the reference's vendored third-party files; they are compiled
 * from where they lie under /root/reference (never copied into this repo) by `make -C oracle ref`, which is a no-op
 * when /root/reference is absent (the GPU box). The rest of the reference (SYCL + Embree) stays unbuildable here.
 * This file only contains the exported wrappers. */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#define STB_IMAGE_IMPLEMENTATION
#define STB_IMAGE_STATIC
#include "stb_image.h"
#define STB_IMAGE_RESIZE_IMPLEMENTATION
#define STB_IMAGE_RESIZE_STATIC
#include "stb_image_resize2.h"

/* RGBA8 decode as tinygltf asks for it. Returns 0 and the size on success; -1 undecodable; -2 `cap` too small. */
int ref_decode_rgba8(const uint8_t* bytes, int size, int* w, int* h, uint8_t* out, size_t cap) {
    int comp = 0;
    unsigned char* data = stbi_load_from_memory(bytes, size, w, h, &comp, 4);
    if (!data) return -1;
    const size_t n = (size_t)(*w) * (size_t)(*h) * 4;
    if (n > cap) {
        stbi_image_free(data);
        return -2;
    }
    memcpy(out, data, n);
    stbi_image_free(data);
    return 0;
}

/* The texture baker's resize to one 512x512 RGBA8 layer. */
int ref_resize_to_layer(const uint8_t* rgba, int w, int h, uint8_t* out512) {
    return stbir_resize_uint8_srgb(rgba, w, h, 0, out512, 512, 512, 0, STBIR_RGBA) == out512 ? 0 : -1;
}
