// PNG reader for glTF texture images: plain and interlaced (Adam7), bit depths 1-16, all five
// colour types, palette transparency (tRNS); output is always RGBA8.  Inflate comes from zlib.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace lsg {

// Throws std::runtime_error with a message naming what is wrong or unsupported.
void decodePNG(const uint8_t* data, size_t size, std::vector<uint8_t>& rgba, uint32_t& width, uint32_t& height);

// 8-bit RGB (channels = 3) or RGBA (4) pixels, row 0 first, as a PNG file image (one IDAT chunk, no interlace).
void encodePNG(const uint8_t* pixels, uint32_t width, uint32_t height, int channels, std::vector<uint8_t>& out);

}  // namespace lsg
