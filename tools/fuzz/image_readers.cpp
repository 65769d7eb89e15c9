// In-process mutation fuzzer for the PNG and JPEG readers (ASan/UBSan build, tools/fuzz/Makefile):
//   ASAN_OPTIONS=detect_leaks=0 ./image_readers <seed> <mutations per file> <file.png|file.jpg>...
// Every decode must either
// succeed or throw; anything else aborts under the sanitizers.
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <fstream>
#include <iterator>
#include <random>
#include <string>
#include <vector>
#include "png_decode.hpp"
#include "jpeg_decode.hpp"
static std::vector<uint8_t> slurp(const char* p) { std::ifstream f(p, std::ios::binary); return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), {}); }
int main(int argc, char** argv) {
  unsigned seed = std::atoi(argv[1]);
  int iters = std::atoi(argv[2]);
  std::mt19937 rng(seed);
  long ok = 0, thrown = 0;
  for (int a = 3; a < argc; ++a) {
    std::vector<uint8_t> base = slurp(argv[a]);
    bool png = std::string(argv[a]).find(".png") != std::string::npos;
    for (int it = 0; it < iters; ++it) {
      std::vector<uint8_t> d = base;
      int nm = 1 + rng() % 4;
      for (int k = 0; k < nm; ++k) {
        unsigned kind = rng() % 10;
        if (d.empty()) break;
        size_t pos = rng() % d.size();
        if (kind < 5) d[pos] = (uint8_t)rng();
        else if (kind < 7) d[pos] ^= (uint8_t)(1u << (rng() % 8));
        else if (kind == 7) d.resize(pos);
        else if (kind == 8) { size_t n = 1 + rng() % 8; d.insert(d.begin() + pos, n, (uint8_t)rng()); }
        else { size_t n = 1 + rng() % 16; if (pos + n < d.size()) d.erase(d.begin() + pos, d.begin() + pos + n); }
      }
      // favour the headers: half of the cases mutate within the first 700 bytes
      if (!d.empty() && (rng() & 1)) d[rng() % std::min<size_t>(d.size(), 700)] = (uint8_t)rng();
      std::vector<uint8_t> rgba; uint32_t w = 0, h = 0;
      try {
        if (png) lsg::decodePNG(d.data(), d.size(), rgba, w, h); else lsg::decodeJPEG(d.data(), d.size(), rgba, w, h);
        if (rgba.size() != (size_t)w * h * 4) { std::printf("SIZE MISMATCH %s it %d\n", argv[a], it); return 2; }
        ++ok;
      } catch (const std::exception&) { ++thrown; }
    }
  }
  std::printf("decoded %ld, rejected %ld\n", ok, thrown);
  return 0;
}
