// texture_resize.cpp — the texture baker's resize to one 512x512 RGBA8 layer.
//
// The reference calls stbir_resize_uint8_srgb(data, w, h, 0, out, 512, 512, 0, STBIR_RGBA)
// (src/image_manager.hpp:52-62; stb_image_resize2, an un-pinned vendored header). This is a restatement of what
// that call does with its defaults, written from the algorithm, not from the code:
//   * colour channels are converted sRGB -> linear, alpha is linear (a / 255);
//   * per axis: shrinking uses the Mitchell-Netravali cubic (B = C = 1/3) stretched by 1/scale and evaluated in
//     OUTPUT pixel units (weight = k(x_out - x_in * scale) * scale); enlarging (or equal size) uses the Catmull-Rom
//     cubic evaluated in INPUT pixel units; support 2 either way; each output pixel's weights are normalised to
//     sum 1; weights that fall outside the image are added to the edge pixel (clamp);
//   * non-premultiplied alpha: colour and colour*alpha are both filtered; the result is (colour*alpha)/alpha, or the
//     plain filtered colour where the filtered alpha vanishes;
//   * back to sRGB / alpha*255 + 0.5, clamped.
// Checked against the reference's own stb build in tests/test_ref_stb.py: identical for
// 512x512 inputs, within 1 LSB elsewhere (the last-bit differences come from stb's table-driven float -> sRGB8
// conversion and its SIMD summation order, neither of which is restated here).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "scene_loader.h"

namespace rthost {
namespace {

float cubic_catmullrom(float x) {
    x = std::fabs(x);
    if (x < 1.0f) return 1.0f - x * x * (2.5f - 1.5f * x);
    if (x < 2.0f) return 2.0f - x * (4.0f + x * (0.5f * x - 2.5f));
    return 0.0f;
}
float cubic_mitchell(float x) {
    x = std::fabs(x);
    if (x < 1.0f) return (16.0f + x * x * (21.0f * x - 36.0f)) / 18.0f;
    if (x < 2.0f) return (32.0f + x * (-60.0f + x * (36.0f - 7.0f * x))) / 18.0f;
    return 0.0f;
}

struct Taps {
    int first = 0;            // first input pixel (already inside the image)
    std::vector<float> w;     // weights of pixels first, first+1, ...
};

// weights of every output pixel of one axis
std::vector<Taps> axis_taps(int in, int out) {
    std::vector<Taps> taps((size_t)out);
    const float scale = (float)out / (float)in, inv_scale = (float)in / (float)out;
    std::vector<int> lo((size_t)out, 0), hi((size_t)out, -1);
    std::vector<std::vector<float>> raw((size_t)out); // weights of pixels lo..hi, possibly outside the image
    if (out >= in) { // Catmull-Rom, evaluated around the output pixel's centre in input space
        const float radius = 2.0f * scale; // in output pixels
        for (int n = 0; n < out; ++n) {
            const float centre = (float)n + 0.5f;
            const float in_centre = centre * inv_scale;
            const int first = (int)std::floor((centre - radius) * inv_scale + 0.5f);
            const int last = (int)std::floor((centre + radius) * inv_scale - 0.5f);
            lo[n] = first, hi[n] = last;
            for (int i = first; i <= last; ++i) raw[n].push_back(cubic_catmullrom(in_centre - ((float)i + 0.5f)));
        }
    } else { // Mitchell, every input pixel spreads over the output pixels within 2 output pixels of it
        const float radius = 2.0f / scale; // in input pixels
        const int reach = (int)std::ceil(radius) + 1;
        for (int i = -reach; i < in + reach; ++i) {
            const float in_centre = (float)i + 0.5f;
            const float out_of_in = in_centre * scale;
            int first = (int)std::floor((in_centre - radius) * scale + 0.5f);
            int last = (int)std::floor((in_centre + radius) * scale - 0.5f);
            first = std::max(first, 0), last = std::min(last, out - 1);
            for (int n = first; n <= last; ++n) {
                const float wgt = cubic_mitchell(((float)n + 0.5f) - out_of_in) * scale;
                if (raw[n].empty()) lo[n] = i;
                // input pixels arrive in ascending order, so each output pixel's list stays contiguous
                while (lo[n] + (int)raw[n].size() < i) raw[n].push_back(0.0f);
                raw[n].push_back(wgt);
                hi[n] = i;
            }
        }
    }
    for (int n = 0; n < out; ++n) {
        float total = 0.0f;
        for (float v : raw[n]) total += v;
        if (total != 0.0f && total != 1.0f) {
            const float s = 1.0f / total;
            for (float& v : raw[n]) v *= s;
        }
        // clamp: fold the weights of pixels outside [0, in) onto the edge pixels
        Taps& t = taps[(size_t)n];
        const int a = std::min(std::max(lo[n], 0), in - 1), b = std::min(std::max(hi[n], 0), in - 1);
        t.first = a;
        t.w.assign((size_t)(b - a + 1), 0.0f);
        for (int i = lo[n]; i <= hi[n]; ++i) t.w[(size_t)(std::min(std::max(i, 0), in - 1) - a)] += raw[n][(size_t)(i - lo[n])];
    }
    return taps;
}

struct SrgbTables {
    float to_linear[256];
    float threshold[255]; // linear value at which the rounded sRGB code steps from v to v+1
    SrgbTables() {
        for (int c = 0; c < 256; ++c) {
            const double x = c / 255.0;
            to_linear[c] = (float)(x <= 0.04045 ? x / 12.92 : std::pow((x + 0.055) / 1.055, 2.4));
        }
        for (int v = 0; v < 255; ++v) {
            const double x = (v + 0.5) / 255.0;
            threshold[v] = (float)(x <= 0.04045 ? x / 12.92 : std::pow((x + 0.055) / 1.055, 2.4));
        }
    }
    uint8_t encode(float lin) const { // correctly rounded in sRGB code space; NaN -> 0
        if (!(lin > 0.0f)) return 0;
        return (uint8_t)(std::upper_bound(threshold, threshold + 255, lin) - threshold);
    }
};
const SrgbTables& tables() {
    static const SrgbTables t;
    return t;
}

} // namespace

void resize_to_layer(const uint8_t* rgba, uint32_t w, uint32_t h, uint8_t* out) {
    constexpr int N = RT_TEX_SIZE;
    if (w == (uint32_t)N && h == (uint32_t)N) { // scale 1 on both axes: the cubic's taps are (0, 1, 0, 0)
        std::memcpy(out, rgba, (size_t)N * N * 4);
        return;
    }
    const SrgbTables& T = tables();
    const std::vector<Taps> tx = axis_taps((int)w, N), ty = axis_taps((int)h, N);
    // 7 floats per pixel: R G B A (plain) and R*A G*A B*A
    constexpr int C = 7;
    std::vector<float> rows((size_t)h * N * C); // after the horizontal pass
    std::vector<float> line((size_t)w * C);
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t* src = rgba + (size_t)y * w * 4;
        for (uint32_t x = 0; x < w; ++x) {
            float* p = &line[(size_t)x * C];
            const float a = (float)src[4 * x + 3] * (1.0f / 255.0f);
            p[0] = T.to_linear[src[4 * x]], p[1] = T.to_linear[src[4 * x + 1]], p[2] = T.to_linear[src[4 * x + 2]], p[3] = a;
            p[4] = p[0] * a, p[5] = p[1] * a, p[6] = p[2] * a;
        }
        float* dst = &rows[(size_t)y * N * C];
        for (int n = 0; n < N; ++n) {
            const Taps& t = tx[(size_t)n];
            float acc[C] = {0, 0, 0, 0, 0, 0, 0};
            for (size_t k = 0; k < t.w.size(); ++k) {
                const float* p = &line[((size_t)t.first + k) * C];
                for (int c = 0; c < C; ++c) acc[c] += p[c] * t.w[k];
            }
            std::memcpy(dst + (size_t)n * C, acc, sizeof(acc));
        }
    }
    for (int m = 0; m < N; ++m) {
        const Taps& t = ty[(size_t)m];
        for (int n = 0; n < N; ++n) {
            float acc[C] = {0, 0, 0, 0, 0, 0, 0};
            for (size_t k = 0; k < t.w.size(); ++k) {
                const float* p = &rows[(((size_t)t.first + k) * N + (size_t)n) * C];
                for (int c = 0; c < C; ++c) acc[c] += p[c] * t.w[k];
            }
            uint8_t* d = out + ((size_t)m * N + (size_t)n) * 4;
            const float alpha = acc[3];
            float r = acc[0], g = acc[1], b = acc[2];
            if (!(alpha < 7.5231638e-37f)) { // 2^-120, below which the plain filtered colour is kept
                const float ia = 1.0f / alpha;
                r = acc[4] * ia, g = acc[5] * ia, b = acc[6] * ia;
            }
            d[0] = T.encode(r), d[1] = T.encode(g), d[2] = T.encode(b);
            const float a255 = alpha * 255.0f + 0.5f;
            d[3] = (uint8_t)(a255 < 0.0f ? 0.0f : a255 > 255.0f ? 255.0f : a255);
        }
    }
}

} // namespace rthost
