// host_capi.cpp — a small C surface over the host loader / PNG codec so the tests can drive them through
// ctypes (librt_host.so). Not part of the rendering boundary (that is include/rt_mi355x.h).
#include <cstring>
#include <string>

#include "jpeg.h"
#include "png.h"
#include "scene_loader.h"

using namespace rthost;

extern "C" {

struct rth_scene {
    LoadedScene sc;
};

static thread_local std::string g_err;
const char* rth_last_error(void) { return g_err.c_str(); }

int rth_load_glb(const char* path, rth_scene** out) {
    if (!path || !out) { g_err = "null argument"; return -1; }
    try {
        rth_scene* s = new rth_scene{load_glb(path, false)};
        *out = s;
        return 0;
    } catch (const std::exception& e) {
        g_err = e.what();
        *out = nullptr;
        return -1;
    }
}
void rth_scene_free(rth_scene* s) { delete s; }
void rth_scene_desc(const rth_scene* s, rt_scene_desc* out) { *out = s->sc.desc(); }
void rth_scene_camera(const rth_scene* s, float pos[3], float dir[3], float* focal, int* has_camera) {
    std::memcpy(pos, s->sc.camera_position, 12), std::memcpy(dir, s->sc.camera_direction, 12);
    *focal = s->sc.camera_focal_length;
    *has_camera = s->sc.has_camera ? 1 : 0;
}

int rth_write_png(const char* path, uint32_t w, uint32_t h, const uint8_t* rgba) {
    return write_png_rgba8(path, w, h, rgba, (size_t)w * 4) ? 0 : -1;
}
// decodes into `out` (capacity bytes); returns 0 and sets w/h, or -1
int rth_decode_png(const uint8_t* data, size_t size, uint32_t* w, uint32_t* h, uint8_t* out, size_t capacity) {
    try {
        std::vector<uint8_t> rgba;
        decode_png_rgba8(data, size, *w, *h, rgba);
        if (rgba.size() > capacity) { g_err = "output buffer too small"; return -1; }
        std::memcpy(out, rgba.data(), rgba.size());
        return 0;
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
}
// PNG or JPEG by signature, as the loader decodes embedded textures
int rth_decode_image(const uint8_t* data, size_t size, uint32_t* w, uint32_t* h, uint8_t* out, size_t capacity) {
    try {
        std::vector<uint8_t> rgba;
        decode_image_rgba8(data, size, *w, *h, rgba);
        if (rgba.size() > capacity) { g_err = "output buffer too small"; return -1; }
        std::memcpy(out, rgba.data(), rgba.size());
        return 0;
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
}
void rth_resize_to_layer(const uint8_t* rgba, uint32_t w, uint32_t h, uint8_t* out512) { resize_to_layer(rgba, w, h, out512); }

} // extern "C"
