#include "jpeg_decode.hpp"

#include <cmath>
#include <cstring>
#include <stdexcept>
#include <string>

// Fixed rules where the JPEG standard leaves a choice:
//  * inverse DCT: the separable definition evaluated in double precision, + 128, rounded half up, clamped;
//  * chroma planes with half the resolution are interpolated with the triangle filter (3/4 of the nearer
//    sample, 1/4 of the farther one, per direction; edges repeat), other integer ratios repeat samples;
//  * YCbCr -> RGB with the JFIF matrix in 16.16 fixed point, rounded half up.

namespace lsg {

namespace {

[[noreturn]] void bad(const std::string& what) { throw std::runtime_error("JPEG: " + what); }

const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// Canonical Huffman table (ITU-T T.81 annex C and F.2.2.3).
struct Huffman {
  bool present = false;
  uint8_t vals[256];
  int mincode[17], maxcode[17], valptr[17];  // per code length 1..16; maxcode -1 = no code of that length

  void build(const uint8_t counts[16], const uint8_t* symbols, int nSymbols) {
    std::memcpy(vals, symbols, (size_t)nSymbols);
    int code = 0, k = 0;
    for (int len = 1; len <= 16; ++len) {
      const int n = counts[len - 1];
      valptr[len] = k;
      mincode[len] = code;
      maxcode[len] = n ? code + n - 1 : -1;
      if (n && code + n > (1 << len)) bad("Huffman table assigns more codes than a length can hold");
      code = (code + n) << 1;
      k += n;
    }
    present = true;
  }
};

// Entropy-coded segment: bytes with 0xFF 0x00 stuffing; any other marker ends the data (zeros follow).
struct BitReader {
  const uint8_t* p;
  const uint8_t* end;
  uint32_t acc = 0;
  int n = 0;
  bool atMarker = false;
  int padded = 0;  // zero bytes fed behind the end of the entropy-coded data

  // Behind the end of the data (a marker, or the end of the file) the reader feeds zero bits: a well-formed scan
  // never consumes them, it only looks up to four bytes ahead.  A scan that goes on decoding from them is cut
  // off after eight bytes' worth, so that the work of a scan stays proportional to the bytes it really has.
  void fill() {
    while (n <= 24) {
      uint32_t b = 0;
      if (atMarker || p >= end) {
        if (++padded > 8) bad("entropy-coded data ends before the scan does");
      } else {
        b = *p;
        if (b == 0xFF) {
          if (p + 1 < end && p[1] == 0x00) {
            p += 2;
          } else {
            atMarker = true;
            b = 0;
          }
        } else {
          ++p;
        }
      }
      acc |= b << (24 - n);
      n += 8;
    }
  }
  int bit() {
    if (n == 0) fill();
    const int v = (int)(acc >> 31);
    acc <<= 1;
    --n;
    return v;
  }
  int bits(int k) {  // k <= 16
    if (k == 0) return 0;
    if (n < k) fill();
    const int v = (int)(acc >> (32 - k));
    acc <<= k;
    n -= k;
    return v;
  }
  int decode(const Huffman& h) {
    int code = 0;
    for (int len = 1; len <= 16; ++len) {
      code = (code << 1) | bit();
      if (h.maxcode[len] >= 0 && code <= h.maxcode[len] && code >= h.mincode[len])
        return h.vals[h.valptr[len] + code - h.mincode[len]];
    }
    bad("bit sequence that is no Huffman code");
  }
  // restart marker RSTm expected here: drop the bits of the finished interval and step over the marker
  void restart(int m) {
    acc = 0;
    n = 0;
    atMarker = false;
    padded = 0;
    while (p < end && *p != 0xFF) ++p;        // (a well-formed file has nothing here)
    while (p + 1 < end && p[1] == 0xFF) ++p;  // fill bytes
    if (p + 1 >= end || p[1] != (uint8_t)(0xD0 + m)) bad("restart marker missing or out of sequence");
    p += 2;
  }
};

inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

struct Component {
  int id = 0, h = 1, v = 1, tq = 0;
  int planeW = 0, planeH = 0;  // whole MCUs
  int compW = 0, compH = 0;    // samples that belong to the image
  std::vector<uint8_t> plane;
  std::vector<int> coefs;  // progressive frames: 64 coefficients per block of the plane, natural order, not yet dequantised
  int pred = 0;
  int td = 0, ta = 0;
  bool decoded = false;
};

struct Idct {
  double c[8][8];  // c[x][u] = C(u)/2 * cos((2x+1) u pi / 16)
  Idct() {
    const double pi = 3.14159265358979323846;
    for (int x = 0; x < 8; ++x)
      for (int u = 0; u < 8; ++u) c[x][u] = (u == 0 ? std::sqrt(0.5) : 1.0) * 0.5 * std::cos((2 * x + 1) * u * pi / 16.0);
  }
  // coef: dequantised coefficients, natural order [v][u]; out: 8 rows of `stride`
  void run(const int coef[64], uint8_t* out, int stride) const {
    double tmp[64];
    for (int v = 0; v < 8; ++v)
      for (int x = 0; x < 8; ++x) {
        double s = 0.0;
        for (int u = 0; u < 8; ++u) s += c[x][u] * (double)coef[8 * v + u];
        tmp[8 * v + x] = s;
      }
    for (int y = 0; y < 8; ++y)
      for (int x = 0; x < 8; ++x) {
        double s = 0.0;
        for (int v = 0; v < 8; ++v) s += c[y][v] * tmp[8 * v + x];
        const double r = std::floor(s + 128.5);
        out[y * stride + x] = (uint8_t)(r < 0.0 ? 0.0 : (r > 255.0 ? 255.0 : r));
      }
  }
};

inline uint8_t clamp8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

}  // namespace

void decodeJPEG(const uint8_t* data, size_t size, std::vector<uint8_t>& rgba, uint32_t& width, uint32_t& height) {
  if (size < 4 || data[0] != 0xFF || data[1] != 0xD8) bad("bad signature");
  static const Idct idct;
  uint16_t quant[4][64];
  bool haveQuant[4] = {false, false, false, false};
  Huffman dc[4], ac[4];
  std::vector<Component> comps;
  int X = 0, Y = 0, hmax = 1, vmax = 1, mcux = 0, mcuy = 0;
  int restartInterval = 0;
  int adobeTransform = -1;
  bool haveFrame = false, ended = false, progressive = false;
  int scans = 0;

  size_t pos = 2;
  while (!ended) {
    // next marker (fill bytes 0xFF may precede it)
    while (pos < size && data[pos] != 0xFF) ++pos;
    while (pos < size && data[pos] == 0xFF) ++pos;
    if (pos >= size) bad("file ends before the end-of-image marker");
    const int marker = data[pos++];
    if (marker == 0xD9) {
      ended = true;
      break;
    }
    if (marker == 0x00 || marker == 0x01 || (marker >= 0xD0 && marker <= 0xD7)) continue;  // stuffing, stand-alone markers
    if (pos + 2 > size) bad("truncated marker segment");
    const size_t len = ((size_t)data[pos] << 8) | data[pos + 1];
    if (len < 2 || pos + len > size) bad("marker segment exceeds the file");
    const uint8_t* seg = data + pos + 2;
    const size_t segLen = len - 2;
    pos += len;

    if (marker == 0xDB) {  // quantisation tables
      size_t o = 0;
      while (o < segLen) {
        const int pq = seg[o] >> 4, tq = seg[o] & 15;
        ++o;
        if (tq > 3 || pq > 1) bad("bad quantisation table header");
        const size_t need = pq ? 128 : 64;
        if (o + need > segLen) bad("truncated quantisation table");
        for (int k = 0; k < 64; ++k) {
          quant[tq][k] = pq ? (uint16_t)((seg[o + 2 * k] << 8) | seg[o + 2 * k + 1]) : seg[o + k];
        }
        o += need;
        haveQuant[tq] = true;
      }
    } else if (marker == 0xC4) {  // Huffman tables
      size_t o = 0;
      while (o < segLen) {
        if (o + 17 > segLen) bad("truncated Huffman table");
        const int tc = seg[o] >> 4, th = seg[o] & 15;
        if (tc > 1 || th > 3) bad("bad Huffman table header");
        int total = 0;
        for (int k = 0; k < 16; ++k) total += seg[o + 1 + k];
        if (total > 256 || o + 17 + (size_t)total > segLen) bad("truncated Huffman table");
        (tc ? ac[th] : dc[th]).build(seg + o + 1, seg + o + 17, total);
        o += 17 + (size_t)total;
      }
    } else if (marker == 0xC0 || marker == 0xC1 || marker == 0xC2) {  // frame header, Huffman-coded DCT
      if (haveFrame) bad("more than one frame");
      progressive = marker == 0xC2;
      if (segLen < 6) bad("truncated frame header");
      if (seg[0] != 8) bad("only 8 bits per sample are supported");
      Y = (seg[1] << 8) | seg[2];
      X = (seg[3] << 8) | seg[4];
      const int nf = seg[5];
      if (X == 0 || Y == 0) bad("image without size (DNL marker) is not supported");
      if (X > 32768 || Y > 32768) bad("image size outside 1..32768");
      // Memory and time are tied to the size of the file, not to what a header claims: 64 Mi pixels at most, and
      // no more than 4096 pixels per byte of file (real files hold tens of pixels per byte at their smallest).
      if ((long long)X * Y > (64ll << 20)) bad("image of more than 64 Mi pixels");
      if ((long long)X * Y > 4096ll * (long long)size) bad("frame header claims more pixels than a file of this size can hold");
      if (nf != 1 && nf != 3) bad(nf == 4 ? "four-component (CMYK) images are not supported" : "bad component count");
      if (segLen < 6 + 3 * (size_t)nf) bad("truncated frame header");
      comps.resize((size_t)nf);
      for (int i = 0; i < nf; ++i) {
        Component& c = comps[(size_t)i];
        c.id = seg[6 + 3 * i];
        c.h = seg[7 + 3 * i] >> 4;
        c.v = seg[7 + 3 * i] & 15;
        c.tq = seg[8 + 3 * i];
        if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) bad("bad component description");
        hmax = c.h > hmax ? c.h : hmax;
        vmax = c.v > vmax ? c.v : vmax;
      }
      mcux = (X + 8 * hmax - 1) / (8 * hmax);
      mcuy = (Y + 8 * vmax - 1) / (8 * vmax);
      for (Component& c : comps) {
        if (hmax % c.h || vmax % c.v) bad("fractional sampling ratios are not supported");
        c.planeW = mcux * c.h * 8;
        c.planeH = mcuy * c.v * 8;
        c.compW = (X * c.h + hmax - 1) / hmax;
        c.compH = (Y * c.v + vmax - 1) / vmax;
        c.plane.assign((size_t)c.planeW * c.planeH, 128);
        if (progressive) c.coefs.assign((size_t)c.planeW * c.planeH, 0);
      }
      haveFrame = true;
    } else if ((marker >= 0xC3 && marker <= 0xCF) && marker != 0xC4 && marker != 0xC8 && marker != 0xCC) {
      bad("lossless, hierarchical and arithmetic-coded JPEG are not supported");
    } else if (marker == 0xCC) {
      bad("arithmetic-coded JPEG is not supported");
    } else if (marker == 0xDD) {
      if (segLen < 2) bad("truncated restart interval");
      restartInterval = (seg[0] << 8) | seg[1];
    } else if (marker == 0xEE) {  // Adobe: says whether three components are YCbCr (1) or RGB (0)
      if (segLen >= 12 && std::memcmp(seg, "Adobe", 5) == 0) adobeTransform = seg[11];
    } else if (marker == 0xDA) {  // scan
      if (!haveFrame) bad("scan before the frame header");
      if (++scans > 256) bad("more than 256 scans");  // progressive files in the wild have about ten
      if (segLen < 1) bad("truncated scan header");
      const int ns = seg[0];
      if (ns < 1 || ns > (int)comps.size() || segLen < 1 + 2 * (size_t)ns + 3) bad("bad scan header");
      std::vector<Component*> sc;
      for (int i = 0; i < ns; ++i) {
        const int cs = seg[1 + 2 * i];
        Component* found = nullptr;
        for (Component& c : comps)
          if (c.id == cs) found = &c;
        if (!found) bad("scan names an unknown component");
        if (!progressive && found->decoded) bad("component coded twice");
        found->td = seg[2 + 2 * i] >> 4;
        found->ta = seg[2 + 2 * i] & 15;
        if (found->td > 3 || found->ta > 3) bad("bad Huffman table selector");
        if (!haveQuant[found->tq]) bad("component uses a quantisation table that was not defined");
        found->pred = 0;
        sc.push_back(found);
      }
      const int ss = seg[1 + 2 * ns], se = seg[2 + 2 * ns], ah = seg[3 + 2 * ns] >> 4, al = seg[3 + 2 * ns] & 15;
      if (!progressive) {
        if (ss != 0 || se != 63 || ah != 0 || al != 0) bad("scan parameters of a progressive file in a sequential frame");
      } else {
        if (ss > se || se > 63 || al > 13 || (ah != 0 && ah != al + 1)) bad("bad progressive scan parameters");
        if (ss == 0 && se != 0) bad("progressive scan mixes DC and AC coefficients");
        if (ss > 0 && ns != 1) bad("progressive AC scan with more than one component");
      }
      for (Component* c : sc) {  // the tables this kind of scan decodes with
        const bool needDc = !progressive || (ss == 0 && ah == 0), needAc = !progressive || ss > 0;
        if ((needDc && !dc[c->td].present) || (needAc && !ac[c->ta].present))
          bad("scan uses a Huffman table that was not defined");
      }

      BitReader br{data + pos, data + size};
      int coef[64];
      auto block = [&](Component& c, int bx, int by) {
        std::memset(coef, 0, sizeof coef);
        const uint16_t* q = quant[c.tq];
        const int t = br.decode(dc[c.td]);
        if (t > 11) bad("DC difference of more than 11 bits");
        c.pred += t ? extend(br.bits(t), t) : 0;
        if (c.pred < -32767 || c.pred > 32767) bad("DC value out of range");
        coef[0] = c.pred * (int)q[0];  // |pred| < 2^15, q < 2^16: fits
        for (int k = 1; k < 64;) {
          const int rs = br.decode(ac[c.ta]);
          const int r = rs >> 4, s = rs & 15;
          if (s == 0) {
            if (r != 15) break;  // end of block
            k += 16;
            continue;
          }
          k += r;
          if (k > 63) bad("coefficient index beyond the block");
          coef[kZigzag[k]] = extend(br.bits(s), s) * (int)q[k];  // |value| < 2^15, q < 2^16: fits
          ++k;
        }
        idct.run(coef, &c.plane[(size_t)by * 8 * c.planeW + (size_t)bx * 8], c.planeW);
      };
      // Progressive scans (ITU-T T.81 annex G) add to the stored coefficients of a block: a band ss..se of the
      // zigzag order, either its first `8 - al` high bits (ah == 0) or one more bit of precision (ah > 0).
      int eobrun = 0;  // blocks still covered by the last end-of-band run
      auto progressiveBlock = [&](Component& c, int bx, int by) {
        int* cf = &c.coefs[((size_t)by * (c.planeW / 8) + (size_t)bx) * 64];
        const int p1 = 1 << al, m1 = -(1 << al);
        auto refine = [&](int& v) {  // one more bit for a coefficient that is already non-zero
          if (br.bit() && (v & p1) == 0) v += v >= 0 ? p1 : m1;
        };
        if (ss == 0) {
          if (ah == 0) {
            const int t = br.decode(dc[c.td]);
            if (t > 11) bad("DC difference of more than 11 bits");
            c.pred += t ? extend(br.bits(t), t) : 0;
            if (c.pred < -32767 || c.pred > 32767) bad("DC value out of range");
            cf[0] = c.pred * (1 << al);
          } else if (br.bit()) {
            cf[0] |= p1;
          }
          return;
        }
        int k = ss;
        if (ah == 0) {
          if (eobrun > 0) {
            --eobrun;
            return;
          }
          while (k <= se) {
            const int rs = br.decode(ac[c.ta]);
            const int r = rs >> 4, sz = rs & 15;
            if (sz == 0) {
              if (r < 15) {  // end of band for this block and 2^r - 1 + (r more bits) blocks after it
                eobrun = (1 << r) - 1;
                if (r) eobrun += br.bits(r);
                break;
              }
              k += 16;
              continue;
            }
            k += r;
            if (k > se) bad("coefficient index beyond the band");
            cf[kZigzag[k]] = extend(br.bits(sz), sz) * (1 << al);
            ++k;
          }
          return;
        }
        if (eobrun == 0) {
          while (k <= se) {
            const int rs = br.decode(ac[c.ta]);
            int r = rs >> 4;
            const int sz = rs & 15;
            int value = 0;
            if (sz == 0) {
              if (r < 15) {
                eobrun = 1 << r;
                if (r) eobrun += br.bits(r);
                break;
              }
            } else {
              if (sz != 1) bad("refinement scan with a coefficient of more than one bit");
              value = br.bit() ? p1 : m1;
            }
            // pass over coefficients that are already non-zero (each takes a correction bit) and over r zeros
            while (k <= se) {
              int& v = cf[kZigzag[k]];
              if (v != 0) {
                refine(v);
              } else {
                if (r == 0) break;
                --r;
              }
              ++k;
            }
            if (value != 0) {
              if (k > se) bad("coefficient index beyond the band");
              cf[kZigzag[k]] = value;
            }
            ++k;
          }
        }
        if (eobrun > 0) {  // the rest of the band holds no new coefficients: correction bits only
          for (; k <= se; ++k) {
            int& v = cf[kZigzag[k]];
            if (v != 0) refine(v);
          }
          --eobrun;
        }
      };
      auto codedBlock = [&](Component& c, int bx, int by) {
        if (progressive) progressiveBlock(c, bx, by);
        else block(c, bx, by);
      };
      long long units, done = 0;
      int nextRst = 0;
      auto maybeRestart = [&]() {
        ++done;
        if (restartInterval && done % restartInterval == 0 && done < units) {
          br.restart(nextRst);
          nextRst = (nextRst + 1) & 7;
          eobrun = 0;
          for (Component* c : sc) c->pred = 0;
        }
      };
      if (ns == 1) {  // non-interleaved: the component's own blocks, row by row
        Component& c = *sc[0];
        const int bw = (c.compW + 7) / 8, bh = (c.compH + 7) / 8;
        units = (long long)bw * bh;
        for (int by = 0; by < bh; ++by)
          for (int bx = 0; bx < bw; ++bx) {
            codedBlock(c, bx, by);
            maybeRestart();
          }
      } else {
        units = (long long)mcux * mcuy;
        for (int my = 0; my < mcuy; ++my)
          for (int mx = 0; mx < mcux; ++mx) {
            for (Component* c : sc)
              for (int v = 0; v < c->v; ++v)
                for (int h = 0; h < c->h; ++h) codedBlock(*c, mx * c->h + h, my * c->v + v);
            maybeRestart();
          }
      }
      for (Component* c : sc) c->decoded = true;
      // go on behind the entropy-coded data: the marker search above finds the marker that ended it
      pos = (size_t)(br.p - data);
    }
    // every other segment (APPn, COM, ...) is skipped
  }
  if (!haveFrame) bad("no frame header");
  for (const Component& c : comps)
    if (!c.decoded) bad("a component has no scan");
  if (progressive) {
    for (Component& c : comps) {
      const uint16_t* q = quant[c.tq];
      int natural[64];  // the table in natural order
      for (int k = 0; k < 64; ++k) natural[kZigzag[k]] = q[k];
      const int bw = c.planeW / 8, bh = c.planeH / 8;
      int coef[64];
      for (int by = 0; by < bh; ++by)
        for (int bx = 0; bx < bw; ++bx) {
          const int* cf = &c.coefs[((size_t)by * bw + (size_t)bx) * 64];
          for (int k = 0; k < 64; ++k) {
            const int v = cf[k] < -32767 ? -32767 : (cf[k] > 32767 ? 32767 : cf[k]);  // (only damaged files exceed this)
            coef[k] = v * natural[k];
          }
          idct.run(coef, &c.plane[(size_t)by * 8 * c.planeW + (size_t)bx * 8], c.planeW);
        }
    }
  }

  // full-resolution sample of component c at pixel (x, y)
  auto sample = [&](const Component& c, int x, int y) -> int {
    const int fx = hmax / c.h, fy = vmax / c.v;
    auto at = [&](int sx, int sy) -> int {
      sx = sx < 0 ? 0 : (sx >= c.compW ? c.compW - 1 : sx);
      sy = sy < 0 ? 0 : (sy >= c.compH ? c.compH - 1 : sy);
      return c.plane[(size_t)sy * c.planeW + sx];
    };
    const int sx = x / fx, sy = y / fy;
    const int nx = fx == 2 ? ((x & 1) ? sx + 1 : sx - 1) : sx;
    const int ny = fy == 2 ? ((y & 1) ? sy + 1 : sy - 1) : sy;
    if (fx == 2 && fy == 2) {
      const int a = 3 * at(sx, sy) + at(sx, ny), b = 3 * at(nx, sy) + at(nx, ny);
      return (3 * a + b + 8) >> 4;
    }
    if (fx == 2) return (3 * at(sx, sy) + at(nx, sy) + 2) >> 2;
    if (fy == 2) return (3 * at(sx, sy) + at(sx, ny) + 2) >> 2;
    return at(sx, sy);
  };

  const bool rgbDirect = comps.size() == 3 && (adobeTransform == 0 || (adobeTransform < 0 && comps[0].id == 'R' &&
                                                                       comps[1].id == 'G' && comps[2].id == 'B'));
  rgba.assign((size_t)X * Y * 4, 255);
  for (int y = 0; y < Y; ++y)
    for (int x = 0; x < X; ++x) {
      uint8_t* o = &rgba[((size_t)y * X + x) * 4];
      if (comps.size() == 1) {
        o[0] = o[1] = o[2] = (uint8_t)sample(comps[0], x, y);
      } else if (rgbDirect) {
        o[0] = (uint8_t)sample(comps[0], x, y);
        o[1] = (uint8_t)sample(comps[1], x, y);
        o[2] = (uint8_t)sample(comps[2], x, y);
      } else {
        const int yy = sample(comps[0], x, y), cb = sample(comps[1], x, y) - 128, cr = sample(comps[2], x, y) - 128;
        // floor((v + 2^15) / 2^16) also for negative v
        auto round16 = [](int v) { return (v + 32768) >= 0 ? (v + 32768) / 65536 : -((65535 - (v + 32768)) / 65536); };
        o[0] = clamp8(yy + round16(91881 * cr));
        o[1] = clamp8(yy + round16(-22554 * cb - 46802 * cr));
        o[2] = clamp8(yy + round16(116130 * cb));
      }
    }
  width = (uint32_t)X;
  height = (uint32_t)Y;
}

}  // namespace lsg
