// png.h — PNG decode/encode for the host adapter (textures in, out.png out).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace rthost {

bool is_png(const uint8_t* data, size_t size);
// Decodes to tightly packed RGBA8 (row 0 first). Throws std::runtime_error on unsupported input.
void decode_png_rgba8(const uint8_t* data, size_t size, uint32_t& width, uint32_t& height, std::vector<uint8_t>& rgba);
std::vector<uint8_t> encode_png_rgba8(uint32_t w, uint32_t h, const uint8_t* rgba, size_t stride_bytes);
// == stbi_write_png(path, w, h, 4, rgba, stride) (src/util.hpp:27). Returns false on failure.
bool write_png_rgba8(const char* path, uint32_t w, uint32_t h, const uint8_t* rgba, size_t stride_bytes);

} // namespace rthost
