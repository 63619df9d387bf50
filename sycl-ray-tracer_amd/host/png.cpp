// png.cpp — PNG decode (embedded glTF textures) and encode (out.png) on zlib.
//
// write_png_rgba8 stands in for stbi_write_png("out.png", w, h, 4, buf, w*4) (src/util.hpp:27);
// decode_png_rgba8 stands in for tinygltf's stb_image decode to 4 channels (deps/include/tiny_gltf.h:2610).
// Bit depth 8 or 16, colour types 0/2/3/4/6 (JPEG textures: jpeg.cpp).
#include "png.h"

#include <zlib.h>

#include <cstdio>
#include <cstring>
#include <stdexcept>

namespace rthost {
namespace {

uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
void put_be32(std::vector<uint8_t>& v, uint32_t x) {
    v.push_back((uint8_t)(x >> 24)), v.push_back((uint8_t)(x >> 16)), v.push_back((uint8_t)(x >> 8)), v.push_back((uint8_t)x);
}
int paeth(int a, int b, int c) {
    int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}
void chunk(std::vector<uint8_t>& out, const char type[4], const std::vector<uint8_t>& data) {
    put_be32(out, (uint32_t)data.size());
    size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), data.begin(), data.end());
    put_be32(out, (uint32_t)crc32(0L, out.data() + start, (uInt)(out.size() - start)));
}

} // namespace

bool is_png(const uint8_t* data, size_t size) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    return size >= 8 && !std::memcmp(data, sig, 8);
}

void decode_png_rgba8(const uint8_t* data, size_t size, uint32_t& width, uint32_t& height, std::vector<uint8_t>& rgba) {
    if (!is_png(data, size)) throw std::runtime_error("image is not a PNG (only PNG textures are supported)");
    size_t p = 8;
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte, trns;
    while (p + 12 <= size) {
        uint32_t len = be32(data + p);
        const char* type = (const char*)data + p + 4;
        const uint8_t* body = data + p + 8;
        if (p + 12 + (size_t)len > size) throw std::runtime_error("PNG: truncated chunk");
        if (!std::memcmp(type, "IHDR", 4)) {
            w = be32(body), h = be32(body + 4);
            depth = body[8], ctype = body[9], interlace = body[12];
        } else if (!std::memcmp(type, "PLTE", 4)) {
            plte.assign(body, body + len);
        } else if (!std::memcmp(type, "tRNS", 4)) {
            trns.assign(body, body + len);
        } else if (!std::memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), body, body + len);
        } else if (!std::memcmp(type, "IEND", 4)) {
            break;
        }
        p += 12 + (size_t)len;
    }
    if (!w || !h || w > 16384 || h > 16384) throw std::runtime_error("PNG: bad dimensions");
    if (interlace > 1) throw std::runtime_error("PNG: unknown interlace method");
    if (depth != 8 && depth != 16) throw std::runtime_error("PNG: only bit depth 8 or 16 is supported");
    int channels = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!channels) throw std::runtime_error("PNG: bad colour type");
    if (ctype == 3 && depth != 8) throw std::runtime_error("PNG: palette images must be 8 bit");
    const size_t bpp = (size_t)channels * (depth / 8);
    // A non-interlaced image is ONE pass over every pixel; an Adam7 image is seven passes, each a reduced image of its own — pixels
    // (x0 + i * dx, y0 + j * dy) of the full one — with its own filtered scanlines (empty passes have none)
    struct Pass { uint32_t x0, y0, dx, dy; };
    static const Pass adam7[7] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
    static const Pass whole = {0, 0, 1, 1};
    const int n_pass = interlace ? 7 : 1;
    size_t raw_size = 0;
    for (int k = 0; k < n_pass; ++k) {
        const Pass& ps = interlace ? adam7[k] : whole;
        const uint32_t pw = (w - ps.x0 + ps.dx - 1) / ps.dx, ph = (h - ps.y0 + ps.dy - 1) / ps.dy;
        if (w > ps.x0 && h > ps.y0 && pw && ph) raw_size += (bpp * pw + 1) * (size_t)ph;
    }
    std::vector<uint8_t> raw(raw_size);
    uLongf raw_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size()) != Z_OK || raw_len != raw.size())
        throw std::runtime_error("PNG: zlib inflate failed");
    rgba.assign((size_t)w * h * 4, 255);
    const uint8_t* line = raw.data();
    for (int k = 0; k < n_pass; ++k) {
        const Pass& ps = interlace ? adam7[k] : whole;
        if (!(w > ps.x0 && h > ps.y0)) continue;
        const uint32_t pw = (w - ps.x0 + ps.dx - 1) / ps.dx, ph = (h - ps.y0 + ps.dy - 1) / ps.dy;
        if (!pw || !ph) continue;
        const size_t stride = bpp * pw;
        std::vector<uint8_t> cur(stride), prev(stride, 0);
        for (uint32_t py = 0; py < ph; ++py, line += stride + 1) {
            const int filter = line[0];
            for (size_t i = 0; i < stride; ++i) {
                int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0, x = line[1 + i];
                switch (filter) {
                case 0: break;
                case 1: x += a; break;
                case 2: x += b; break;
                case 3: x += (a + b) / 2; break;
                case 4: x += paeth(a, b, c); break;
                default: throw std::runtime_error("PNG: bad filter");
                }
                cur[i] = (uint8_t)x;
            }
            const uint32_t y = ps.y0 + py * ps.dy;
            for (uint32_t px = 0; px < pw; ++px) {
                const uint32_t xx = ps.x0 + px * ps.dx;
                const uint8_t* s = cur.data() + bpp * px;
                uint8_t* d = rgba.data() + ((size_t)y * w + xx) * 4;
                auto ch = [&](int kk) { return depth == 8 ? s[kk] : s[2 * kk]; }; // 16 bit: high byte
                switch (ctype) {
                case 0: d[0] = d[1] = d[2] = ch(0); break;
                case 2: d[0] = ch(0), d[1] = ch(1), d[2] = ch(2); break;
                case 3: {
                    size_t idx = s[0];
                    if (idx * 3 + 2 < plte.size()) d[0] = plte[idx * 3], d[1] = plte[idx * 3 + 1], d[2] = plte[idx * 3 + 2];
                    if (idx < trns.size()) d[3] = trns[idx];
                    break;
                }
                case 4: d[0] = d[1] = d[2] = ch(0), d[3] = ch(1); break;
                case 6: d[0] = ch(0), d[1] = ch(1), d[2] = ch(2), d[3] = ch(3); break;
                }
            }
            prev.swap(cur);
        }
    }
    width = w, height = h;
}

std::vector<uint8_t> encode_png_rgba8(uint32_t w, uint32_t h, const uint8_t* rgba, size_t stride_bytes) {
    std::vector<uint8_t> raw;
    raw.reserve(((size_t)w * 4 + 1) * h);
    for (uint32_t y = 0; y < h; ++y) {
        raw.push_back(0); // filter: none
        raw.insert(raw.end(), rgba + stride_bytes * y, rgba + stride_bytes * y + (size_t)w * 4);
    }
    uLongf clen = compressBound((uLong)raw.size());
    std::vector<uint8_t> comp(clen);
    if (compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), 6) != Z_OK) throw std::runtime_error("PNG: deflate failed");
    comp.resize(clen);
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, w), put_be32(ihdr, h);
    ihdr.push_back(8), ihdr.push_back(6), ihdr.push_back(0), ihdr.push_back(0), ihdr.push_back(0);
    chunk(out, "IHDR", ihdr);
    chunk(out, "IDAT", comp);
    chunk(out, "IEND", {});
    return out;
}

bool write_png_rgba8(const char* path, uint32_t w, uint32_t h, const uint8_t* rgba, size_t stride_bytes) {
    std::vector<uint8_t> png;
    try {
        png = encode_png_rgba8(w, h, rgba, stride_bytes);
    } catch (const std::exception&) {
        return false;
    }
    FILE* f = std::fopen(path, "wb");
    if (!f) return false;
    const bool ok = std::fwrite(png.data(), 1, png.size(), f) == png.size();
    return std::fclose(f) == 0 && ok;
}

} // namespace rthost
