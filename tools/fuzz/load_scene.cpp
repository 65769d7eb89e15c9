// Loads a glTF file with both host BVH builders through the host C API; prints "ok" or the error message.
// Built with sanitizers by tools/fuzz/Makefile and driven by tools/fuzz/gltf_mutate.py.
#include <cstdio>
struct lpt_ctx;
extern "C" { void* lpth_scene_load(const char*, int, lpt_ctx*); void lpth_scene_free(void*); const char* lpth_last_error(); }
int main(int argc, char** argv) {
  for (int b = 0; b < 2; ++b) {
    void* s = lpth_scene_load(argv[1], b, nullptr);
    if (!s) std::printf("err %.100s\n", lpth_last_error()); else { std::printf("ok\n"); lpth_scene_free(s); }
  }
  return 0;
}
