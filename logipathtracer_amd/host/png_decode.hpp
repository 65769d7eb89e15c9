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

}  // namespace lsg
