// Headless command-line driver standing in for the interactive loop of the reference
// (src/Main.cpp): loads a glTF file, renders a number of samples per pixel with RendererPT
// and writes the accumulated radiance (PFM) and the display image (PPM and PNG).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "png_decode.hpp"
#include "renderer_pt.hpp"

static void writePFM(const std::string& path, const std::vector<float>& rgba, int w, int h, float scale) {
  FILE* f = std::fopen(path.c_str(), "wb");
  if (!f) throw std::runtime_error("cannot write " + path);
  std::fprintf(f, "PF\n%d %d\n-1.0\n", w, h);
  std::vector<float> row(3 * (size_t)w);
  for (int y = 0; y < h; ++y) {  // PFM stores the bottom row first = accumulation row 0
    for (int x = 0; x < w; ++x)
      for (int c = 0; c < 3; ++c) row[3 * x + c] = rgba[4 * ((size_t)y * w + x) + c] * scale;
    std::fwrite(row.data(), sizeof(float), row.size(), f);
  }
  std::fclose(f);
}

static void writePPM(const std::string& path, const std::vector<float>& rgba, int w, int h) {
  FILE* f = std::fopen(path.c_str(), "wb");
  if (!f) throw std::runtime_error("cannot write " + path);
  std::fprintf(f, "P6\n%d %d\n255\n", w, h);
  std::vector<unsigned char> row(3 * (size_t)w);
  for (int y = 0; y < h; ++y) {
    for (int x = 0; x < w; ++x)
      for (int c = 0; c < 3; ++c) {
        float v = rgba[4 * ((size_t)y * w + x) + c];
        v = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
        row[3 * x + c] = (unsigned char)(v * 255.0f + 0.5f);
      }
    std::fwrite(row.data(), 1, row.size(), f);
  }
  std::fclose(f);
}

static void writePNG(const std::string& path, const std::vector<float>& rgba, int w, int h) {
  std::vector<uint8_t> px(3 * (size_t)w * h), file;
  for (size_t i = 0; i < (size_t)w * h; ++i)
    for (int c = 0; c < 3; ++c) {
      float v = rgba[4 * i + c];
      v = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
      px[3 * i + c] = (unsigned char)(v * 255.0f + 0.5f);
    }
  lsg::encodePNG(px.data(), (uint32_t)w, (uint32_t)h, 3, file);
  FILE* f = std::fopen(path.c_str(), "wb");
  if (!f) throw std::runtime_error("cannot write " + path);
  std::fwrite(file.data(), 1, file.size(), f);
  std::fclose(f);
}

int main(int argc, char** argv) {
  std::string scenePath, out = "render";
  int width = 1920, height = 1080, spp = 64, bounces = 10, device = 0, bsdf = 0, pipeline = LPT_PIPELINE_AUTO, builder = 0;
  uint32_t seed = 0x5EED;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    auto next = [&]() -> const char* { if (i + 1 >= argc) { std::cerr << "missing value for " << a << "\n"; std::exit(2); } return argv[++i]; };
    if (a == "--width") width = std::atoi(next());
    else if (a == "--height") height = std::atoi(next());
    else if (a == "--spp") spp = std::atoi(next());
    else if (a == "--bounces") bounces = std::atoi(next());
    else if (a == "--seed") seed = (uint32_t)std::strtoul(next(), nullptr, 0);
    else if (a == "--device") device = std::atoi(next());
    else if (a == "--basic-bsdf") bsdf = 1;
    else if (a == "--wavefront") pipeline = 0;
    else if (a == "--megakernel") pipeline = 1;
    else if (a == "--persistent") pipeline = 2;
    else if (a == "--hybrid") pipeline = LPT_PIPELINE_HYBRID;
    else if (a == "--builder") builder = std::atoi(next());
    else if (a == "--out") out = next();
    else if (a[0] != '-') scenePath = a;
    else { std::cerr << "unknown option " << a << "\n"; return 2; }
  }
  if (scenePath.empty()) {
    std::cerr << "usage: lpt_render scene.gltf [--width W --height H --spp N --bounces B --seed S --device D --basic-bsdf --wavefront | --megakernel | --persistent | --hybrid --builder 0|1|2|3 --out prefix]\n";
    return 2;
  }
  try {
    lsg::GLTFLoader loader;
    std::vector<lsg::Ref<lsg::Scene>> scenes = loader.load(scenePath);
    RendererConfiguration config;
    config.renderScale = 1;
    config.device = device;
    RendererPT renderer(cppglfw::Window("lpt_render", width, height), config);
    renderer.setSeed(seed);
    renderer.setMaxTraceDepth(bounces);
    renderer.setBsdfMode(bsdf);
    renderer.setPipeline(pipeline);
    renderer.setMeshBuilder((PTSceneConverter::MeshBuilder)builder);
    renderer.setVerbose(true);
    renderer.loadScene(scenes[0]);
    renderer.drawFrames((uint32_t)spp);
    std::vector<float> accum, display;
    renderer.readAccumulation(accum);
    renderer.resolve(display);
    writePFM(out + ".pfm", accum, width, height, 1.0f / (float)spp);
    writePPM(out + ".ppm", display, width, height);
    writePNG(out + ".png", display, width, height);
    std::cout << "wrote " << out << ".pfm, " << out << ".ppm and " << out << ".png (" << renderer.samples() << " spp)\n";
  } catch (const std::exception& e) {
    std::cerr << "error: " << e.what() << "\n";
    return 1;
  }
  return 0;
}
