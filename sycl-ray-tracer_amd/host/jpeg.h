// jpeg.h — JPEG decode for embedded glTF textures (baseline, extended-sequential and progressive Huffman frames).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace rthost {

bool is_jpeg(const uint8_t* data, size_t size);
// Decodes to tightly packed RGBA8 (alpha 255), row 0 first. Throws std::runtime_error on unsupported input.
void decode_jpeg_rgba8(const uint8_t* data, size_t size, uint32_t& width, uint32_t& height, std::vector<uint8_t>& rgba);

// PNG or JPEG, chosen by signature (as the reference's decoder does, whatever the glTF mimeType says)
void decode_image_rgba8(const uint8_t* data, size_t size, uint32_t& width, uint32_t& height, std::vector<uint8_t>& rgba);

} // namespace rthost
