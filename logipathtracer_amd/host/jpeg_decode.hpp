// JPEG reader for glTF texture images: Huffman-coded DCT frames, sequential (baseline and extended, SOF0 /
// SOF1) and progressive (SOF2), at 8 bits per sample, one component (grey) or three (YCbCr, or RGB when an
// Adobe marker says so), any sampling factors, interleaved and non-interleaved scans, restart intervals.
// Arithmetic-coded, lossless, hierarchical, 12-bit and four-component files are refused by name.  Output is RGBA8.
//
// JPEG leaves the inverse DCT and the chroma interpolation to the decoder within an error bound, so two
// decoders differ by a level or two per sample; the rules used here are fixed (see jpeg_decode.cpp) so
// that the same file gives the same texels on every host.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace lsg {

// Throws std::runtime_error with a message naming what is wrong or unsupported.
void decodeJPEG(const uint8_t* data, size_t size, std::vector<uint8_t>& rgba, uint32_t& width, uint32_t& height);

}  // namespace lsg
