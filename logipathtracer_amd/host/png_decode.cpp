#include "png_decode.hpp"

#include <zlib.h>

#include <cstring>
#include <stdexcept>

namespace lsg {

namespace {

uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

int paeth(int a, int b, int c) {
  int p = a + b - c;
  int pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
  if (pa <= pb && pa <= pc) return a;
  return pb <= pc ? b : c;
}

}  // namespace

void decodePNG(const uint8_t* data, size_t size, std::vector<uint8_t>& rgba, uint32_t& width, uint32_t& height) {
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  if (size < 8 || std::memcmp(data, sig, 8) != 0) throw std::runtime_error("PNG: bad signature");
  size_t pos = 8;
  bool haveHeader = false;
  uint32_t w = 0, h = 0;
  int depth = 0, colorType = 0;
  bool interlaced = false;
  std::vector<uint8_t> idat, palette, trns;
  bool ended = false;
  while (!ended && pos + 12 <= size) {
    const uint32_t len = be32(data + pos);
    const uint8_t* type = data + pos + 4;
    if (len > size - pos - 12) throw std::runtime_error("PNG: chunk exceeds file");
    const uint8_t* body = data + pos + 8;
    if (std::memcmp(type, "IHDR", 4) == 0) {
      if (len != 13) throw std::runtime_error("PNG: bad IHDR");
      w = be32(body);
      h = be32(body + 4);
      depth = body[8];
      colorType = body[9];
      if (body[10] != 0 || body[11] != 0) throw std::runtime_error("PNG: unknown compression or filter method");
      if (body[12] > 1) throw std::runtime_error("PNG: unknown interlace method");
      interlaced = body[12] == 1;
      haveHeader = true;
    } else if (std::memcmp(type, "PLTE", 4) == 0) {
      palette.assign(body, body + len);
    } else if (std::memcmp(type, "tRNS", 4) == 0) {
      trns.assign(body, body + len);
    } else if (std::memcmp(type, "IDAT", 4) == 0) {
      idat.insert(idat.end(), body, body + len);
    } else if (std::memcmp(type, "IEND", 4) == 0) {
      ended = true;
    }
    pos += 12 + (size_t)len;
  }
  if (!haveHeader || idat.empty()) throw std::runtime_error("PNG: missing IHDR or IDAT");
  if (w == 0 || h == 0 || w > 32768u || h > 32768u) throw std::runtime_error("PNG: image size outside 1..32768");
  int channels;
  switch (colorType) {
    case 0: channels = 1; break;
    case 2: channels = 3; break;
    case 3: channels = 1; break;
    case 4: channels = 2; break;
    case 6: channels = 4; break;
    default: throw std::runtime_error("PNG: unknown colour type");
  }
  const bool depthOk = (colorType == 0 && (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) ||
                       (colorType == 3 && (depth == 1 || depth == 2 || depth == 4 || depth == 8)) ||
                       ((colorType == 2 || colorType == 4 || colorType == 6) && (depth == 8 || depth == 16));
  if (!depthOk) throw std::runtime_error("PNG: bit depth not allowed for the colour type");
  if (colorType == 3 && palette.size() < 3) throw std::runtime_error("PNG: palette image without PLTE");

  const size_t bitsPerPixel = (size_t)channels * (size_t)depth;
  const size_t bpp = (bitsPerPixel + 7) / 8;  // filter distance in bytes
  // The image as a sequence of passes: one for a plain file, seven reduced images for an interlaced one
  // (Adam7: pass p holds the pixels x0 + i*dx, y0 + j*dy).  Every pass is a run of filtered scan lines of its own.
  struct Pass { uint32_t x0, y0, dx, dy, w, h; size_t rowBytes; };
  std::vector<Pass> passes;
  if (!interlaced) {
    passes.push_back(Pass{0, 0, 1, 1, w, h, ((size_t)w * bitsPerPixel + 7) / 8});
  } else {
    static const uint32_t X0[7] = {0, 4, 0, 2, 0, 1, 0}, Y0[7] = {0, 0, 4, 0, 2, 0, 1};
    static const uint32_t DX[7] = {8, 8, 4, 4, 2, 2, 1}, DY[7] = {8, 8, 8, 4, 4, 2, 2};
    for (int p = 0; p < 7; ++p) {
      const uint32_t pw = w > X0[p] ? (w - X0[p] + DX[p] - 1) / DX[p] : 0u;
      const uint32_t ph = h > Y0[p] ? (h - Y0[p] + DY[p] - 1) / DY[p] : 0u;
      if (pw && ph) passes.push_back(Pass{X0[p], Y0[p], DX[p], DY[p], pw, ph, ((size_t)pw * bitsPerPixel + 7) / 8});
    }
  }
  size_t rawBytes = 0;
  for (const Pass& ps : passes) rawBytes += (ps.rowBytes + 1) * (size_t)ps.h;
  std::vector<uint8_t> raw(rawBytes);
  z_stream zs;
  std::memset(&zs, 0, sizeof zs);
  if (inflateInit(&zs) != Z_OK) throw std::runtime_error("PNG: inflateInit failed");
  zs.next_in = idat.data();
  zs.avail_in = (uInt)idat.size();
  zs.next_out = raw.data();
  zs.avail_out = (uInt)raw.size();
  const int zr = inflate(&zs, Z_FINISH);
  const size_t produced = raw.size() - zs.avail_out;
  inflateEnd(&zs);
  if ((zr != Z_STREAM_END && zr != Z_OK && zr != Z_BUF_ERROR) || produced != raw.size())
    throw std::runtime_error("PNG: compressed data is damaged or has the wrong length");

  rgba.assign((size_t)w * h * 4, 255);
  const int maxv = (1 << (depth > 8 ? 8 : depth)) - 1;
  size_t passStart = 0;
  for (const Pass& ps : passes) {
    const size_t rowBytes = ps.rowBytes;
    // undo the scan-line filters in place (the line above the first one of a pass counts as zeros)
    std::vector<uint8_t> zeroRow(rowBytes, 0);
    for (uint32_t y = 0; y < ps.h; ++y) {
      uint8_t* line = raw.data() + passStart + (rowBytes + 1) * (size_t)y;
      const int filter = line[0];
      uint8_t* cur = line + 1;
      const uint8_t* up = y ? line - rowBytes : zeroRow.data();
      for (size_t i = 0; i < rowBytes; ++i) {
        const int a = i >= bpp ? cur[i - bpp] : 0;
        const int b = up[i];
        const int c = i >= bpp ? up[i - bpp] : 0;
        int v = cur[i];
        switch (filter) {
          case 0: break;
          case 1: v += a; break;
          case 2: v += b; break;
          case 3: v += (a + b) >> 1; break;
          case 4: v += paeth(a, b, c); break;
          default: throw std::runtime_error("PNG: unknown scan-line filter");
        }
        cur[i] = (uint8_t)v;
      }
    }
    for (uint32_t y = 0; y < ps.h; ++y) {
      const uint8_t* cur = raw.data() + passStart + (rowBytes + 1) * (size_t)y + 1;
      for (uint32_t x = 0; x < ps.w; ++x) {
        int s[4] = {0, 0, 0, 0};  // samples reduced to 8 bits (16-bit: high byte)
        for (int k = 0; k < channels; ++k) {
          const size_t idx = (size_t)x * channels + k;
          if (depth == 16) s[k] = cur[idx * 2];
          else if (depth == 8) s[k] = cur[idx];
          else {
            const size_t bit = idx * (size_t)depth;
            s[k] = (cur[bit >> 3] >> (8 - depth - (int)(bit & 7))) & maxv;
          }
        }
        uint8_t* o = &rgba[((size_t)(ps.y0 + y * ps.dy) * w + (ps.x0 + x * ps.dx)) * 4];
        switch (colorType) {
          case 0: {
            const int g = depth < 8 ? s[0] * 255 / maxv : s[0];
            o[0] = o[1] = o[2] = (uint8_t)g;
            break;
          }
          case 2: o[0] = (uint8_t)s[0]; o[1] = (uint8_t)s[1]; o[2] = (uint8_t)s[2]; break;
          case 3: {
            const size_t pi = (size_t)s[0];
            if (pi * 3 + 2 >= palette.size()) throw std::runtime_error("PNG: palette index out of range");
            o[0] = palette[pi * 3]; o[1] = palette[pi * 3 + 1]; o[2] = palette[pi * 3 + 2];
            if (pi < trns.size()) o[3] = trns[pi];
            break;
          }
          case 4: o[0] = o[1] = o[2] = (uint8_t)s[0]; o[3] = (uint8_t)s[1]; break;
          case 6: o[0] = (uint8_t)s[0]; o[1] = (uint8_t)s[1]; o[2] = (uint8_t)s[2]; o[3] = (uint8_t)s[3]; break;
        }
      }
    }
    passStart += (rowBytes + 1) * (size_t)ps.h;
  }
  width = w;
  height = h;
}

void encodePNG(const uint8_t* pixels, uint32_t width, uint32_t height, int channels, std::vector<uint8_t>& out) {
  if (!pixels || width == 0 || height == 0 || (channels != 3 && channels != 4))
    throw std::runtime_error("PNG: nothing to encode");
  const size_t rowBytes = (size_t)width * (size_t)channels;
  std::vector<uint8_t> raw((rowBytes + 1) * (size_t)height);
  for (uint32_t y = 0; y < height; ++y) {  // filter type 0 (none) in front of every scan line
    raw[(rowBytes + 1) * y] = 0;
    std::memcpy(&raw[(rowBytes + 1) * y + 1], pixels + rowBytes * y, rowBytes);
  }
  uLongf packedSize = compressBound((uLong)raw.size());
  std::vector<uint8_t> packed(packedSize);
  if (compress2(packed.data(), &packedSize, raw.data(), (uLong)raw.size(), 6) != Z_OK)
    throw std::runtime_error("PNG: compression failed");
  packed.resize(packedSize);
  auto be32 = [](std::vector<uint8_t>& v, uint32_t x) {
    v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x);
  };
  auto chunk = [&](const char* tag, const std::vector<uint8_t>& data) {
    be32(out, (uint32_t)data.size());
    const size_t start = out.size();
    out.insert(out.end(), tag, tag + 4);
    out.insert(out.end(), data.begin(), data.end());
    be32(out, (uint32_t)crc32(0L, out.data() + start, (uInt)(out.size() - start)));
  };
  static const uint8_t signature[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n'};
  out.assign(signature, signature + 8);
  std::vector<uint8_t> ihdr;
  be32(ihdr, width);
  be32(ihdr, height);
  ihdr.push_back(8);                                    // bit depth
  ihdr.push_back(channels == 4 ? 6 : 2);                // colour type: RGBA / RGB
  ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);  // deflate, adaptive filtering, not interlaced
  chunk("IHDR", ihdr);
  chunk("IDAT", packed);
  chunk("IEND", {});
}

}  // namespace lsg
