// The reference seeds each frame with two values of the C library's rand() after
// srand(time(0)) (src/RendererPT.cpp:13,584-585).  This is glibc's default generator
// (TYPE_3 additive feedback, degree 31, separation 3) restated so that a given
// srand() seed yields the same frame seeds on any host, without touching the
// process-wide rand() state.
#pragma once
#include <cstdint>

class GlibcRand {
 public:
  explicit GlibcRand(uint32_t seed = 1) { srand(seed); }

  void srand(uint32_t seed) {
    if (seed == 0) seed = 1;
    int32_t word = (int32_t)seed;
    r_[0] = (uint32_t)word;
    for (int i = 1; i < 31; ++i) {
      // 16807 * word mod (2^31 - 1) without overflow (Schrage)
      int32_t hi = word / 127773, lo = word % 127773;
      word = 16807 * lo - 2836 * hi;
      if (word < 0) word += 2147483647;
      r_[i] = (uint32_t)word;
    }
    front_ = 3;
    rear_ = 0;
    for (int i = 0; i < 310; ++i) (void)next();
  }

  // Same value sequence as glibc rand(): 31-bit results.
  uint32_t rand() { return next(); }

 private:
  uint32_t next() {
    r_[front_] += r_[rear_];
    uint32_t out = r_[front_] >> 1;
    front_ = (front_ + 1) % 31;
    rear_ = (rear_ + 1) % 31;
    return out;
  }
  uint32_t r_[31];
  int front_ = 3, rear_ = 0;
};
