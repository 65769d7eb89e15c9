// Flat C entry points over the C++ host classes, for bindings (ctypes) and tests.
// Every function catches C++ exceptions and reports them through an error string.
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "glibc_rand.hpp"
#include "png_decode.hpp"
#include "pt_scene_converter.hpp"
#include "renderer_pt.hpp"

namespace {

thread_local std::string g_error;

struct SceneHandle {
  lsg::Ref<lsg::Scene> scene;
  std::unique_ptr<PTSceneConverter> converter;
};

struct RendererHandle {
  std::unique_ptr<RendererPT> renderer;
  lsg::Ref<lsg::Scene> scene;
  // loadScene on a thread of its own, as src/Main.cpp:45 runs it
  std::thread loadThread;
  std::atomic<bool> loadRunning{false};  // set by the starter, cleared by the thread as its last step
  int loadStatus = 0;                    // written by the thread before it clears loadRunning, read after a join
  std::string loadError;
  ~RendererHandle() {
    if (loadThread.joinable()) loadThread.join();
  }
};

template <typename F>
int guarded(F&& f) {
  try {
    f();
    return 0;
  } catch (const std::exception& e) {
    g_error = e.what();
  } catch (...) {
    g_error = "unknown C++ exception";
  }
  return -1;
}

}  // namespace

extern "C" {

const char* lpth_last_error() { return g_error.c_str(); }

// ---- images ---------------------------------------------------------------------------------

// Writes 8-bit RGB (channels 3) or RGBA (4) pixels, row 0 first, to `path` as a PNG file.
int lpth_write_png(const char* path, const uint8_t* pixels, uint32_t width, uint32_t height, int channels) {
  return guarded([&] {
    std::vector<uint8_t> file;
    lsg::encodePNG(pixels, width, height, channels, file);
    FILE* f = std::fopen(path, "wb");
    if (!f) throw std::runtime_error(std::string("cannot write ") + path);
    const size_t n = std::fwrite(file.data(), 1, file.size(), f);
    std::fclose(f);
    if (n != file.size()) throw std::runtime_error(std::string("short write to ") + path);
  });
}

// ---- scene conversion (PTSceneConverter) ------------------------------------------------

// builder: 0 SplitBVH (host), 1 BinnedSAH (host), 2 LBVH, 3 PLOC (GPU; need ctx).  ctx may be NULL
// for host-only conversion; when given, the buffers are also uploaded to it.
void* lpth_scene_load(const char* gltf_path, int builder, lpt_ctx* ctx) {
  SceneHandle* h = nullptr;
  int rc = guarded([&] {
    auto handle = std::make_unique<SceneHandle>();
    lsg::GLTFLoader loader;
    auto scenes = loader.load(gltf_path);
    handle->scene = scenes.at(0);
    handle->converter = std::make_unique<PTSceneConverter>(ctx);
    handle->converter->setMeshBuilder((PTSceneConverter::MeshBuilder)builder);
    {  // builder parameters for comparison runs (defaults otherwise)
      lsg::bvh::BuildOptions o;
      bool any = false;
      if (const char* e = std::getenv("LPT_BVH_CHILD_ORDER")) { o.childOrder = std::atoi(e); any = true; }
      if (const char* e = std::getenv("LPT_BVH_LEAF")) { o.maxLeafSize = o.minLeafSize = (uint32_t)std::atoi(e); any = true; }
      if (const char* e = std::getenv("LPT_BVH_MIN_LEAF")) { o.minLeafSize = (uint32_t)std::atoi(e); any = true; }
      if (const char* e = std::getenv("LPT_BVH_TRAVERSAL_COST")) { o.traversalCost = (float)std::atof(e); any = true; }
      if (const char* e = std::getenv("LPT_BVH_BINS")) { o.bins = (uint32_t)std::atoi(e); any = true; }
      if (const char* e = std::getenv("LPT_BVH_SPLIT_ALPHA")) { o.splitAlpha = (float)std::atof(e); any = true; }
      if (any) handle->converter->setBuildOptions(o);
    }
    handle->converter->loadScene(handle->scene);
    h = handle.release();
  });
  return rc == 0 ? h : nullptr;
}

void lpth_scene_free(void* scene) { delete (SceneHandle*)scene; }

// counts = {objects, object BVH nodes, vertices, mesh BVH nodes, cameras}
void lpth_scene_counts(void* scene, uint32_t counts[5]) {
  auto* h = (SceneHandle*)scene;
  counts[0] = (uint32_t)h->converter->getObjectDataBuffer().size();
  counts[1] = (uint32_t)h->converter->getObjectBvhNodesBuffer().size();
  counts[2] = (uint32_t)h->converter->getVerticesBuffer().size();
  counts[3] = (uint32_t)h->converter->getMeshBvhNodesBuffer().size();
  counts[4] = (uint32_t)h->converter->getCameras().size();
}
const void* lpth_scene_objects(void* scene) { return ((SceneHandle*)scene)->converter->getObjectDataBuffer().data(); }
const void* lpth_scene_object_bvh(void* scene) { return ((SceneHandle*)scene)->converter->getObjectBvhNodesBuffer().data(); }
const void* lpth_scene_vertices(void* scene) { return ((SceneHandle*)scene)->converter->getVerticesBuffer().data(); }
const void* lpth_scene_mesh_bvh(void* scene) { return ((SceneHandle*)scene)->converter->getMeshBvhNodesBuffer().data(); }

// Textures in the order the object records index them.  info = {width, height, magFilter,
// addressModeU, addressModeV}; the returned pointer addresses width*height*4 RGBA8 bytes.
uint32_t lpth_scene_texture_count(void* scene) { return (uint32_t)((SceneHandle*)scene)->converter->getTextures().size(); }
const void* lpth_scene_texture(void* scene, uint32_t index, uint32_t info[5]) {
  const auto& tex = ((SceneHandle*)scene)->converter->getTextures();
  if (index >= tex.size()) return nullptr;
  info[0] = tex[index].image->width();
  info[1] = tex[index].image->height();
  info[2] = tex[index].magFilter;
  info[3] = tex[index].addressModeU;
  info[4] = tex[index].addressModeV;
  return tex[index].image->rawPixelData();
}

// World matrix (column-major) and vertical field of view of camera `index`.
int lpth_scene_camera(void* scene, uint32_t index, float world_matrix[16], float* fov_y) {
  return guarded([&] {
    auto* h = (SceneHandle*)scene;
    const auto& cams = h->converter->getCameras();
    if (index >= cams.size()) throw std::runtime_error("camera index out of range");
    lsg::mat4 m;
    if (auto tr = cams[index]->getComponent<lsg::Transform>()) m = tr->worldMatrix();
    std::memcpy(world_matrix, m.m, sizeof m.m);
    *fov_y = cams[index]->getComponent<lsg::PerspectiveCamera>()->fov();
  });
}

// ---- renderer (RendererPT) ---------------------------------------------------------------

void* lpth_renderer_create(int width, int height, float render_scale, int device) {
  RendererHandle* h = nullptr;
  int rc = guarded([&] {
    auto handle = std::make_unique<RendererHandle>();
    RendererConfiguration config("lpt", width, height, render_scale);
    config.device = device;
    handle->renderer = std::make_unique<RendererPT>(cppglfw::Window("lpt", width, height), config);
    h = handle.release();
  });
  return rc == 0 ? h : nullptr;
}

void lpth_renderer_destroy(void* r) { delete (RendererHandle*)r; }

int lpth_renderer_load_gltf(void* r, const char* path, int builder) {
  return guarded([&] {
    auto* h = (RendererHandle*)r;
    lsg::GLTFLoader loader;
    h->scene = loader.load(path).at(0);
    h->renderer->setMeshBuilder((PTSceneConverter::MeshBuilder)builder);
    h->renderer->loadScene(h->scene);
  });
}

// Starts loading on a second host thread and returns at once (src/Main.cpp:45); the caller may spin
// lpth_renderer_draw_frame meanwhile, which does nothing until the scene is in place
// (src/RendererPT.cpp:608-612).  lpth_renderer_load_wait joins the thread and returns its status.
int lpth_renderer_load_gltf_async(void* r, const char* path, int builder) {
  return guarded([&] {
    auto* h = (RendererHandle*)r;
    if (h->loadRunning.load()) throw std::runtime_error("a load is already running");
    if (h->loadThread.joinable()) h->loadThread.join();  // a finished load nobody has waited for
    h->loadStatus = 0;
    h->loadError.clear();
    std::string file(path);
    h->loadRunning = true;
    h->loadThread = std::thread([h, file, builder] {
      try {
        lsg::GLTFLoader loader;
        lsg::Ref<lsg::Scene> scene = loader.load(file).at(0);
        h->renderer->setMeshBuilder((PTSceneConverter::MeshBuilder)builder);
        h->renderer->loadScene(scene);
        h->scene = scene;  // (the previous scene is let go of only once the new one is in place)
      } catch (const std::exception& e) {
        h->loadStatus = -1;
        h->loadError = e.what();
      } catch (...) {
        h->loadStatus = -1;
        h->loadError = "unknown C++ exception";
      }
      h->loadRunning = false;
    });
  });
}
int lpth_renderer_load_wait(void* r) {
  auto* h = (RendererHandle*)r;
  if (h->loadThread.joinable()) h->loadThread.join();
  if (h->loadStatus != 0) g_error = h->loadError;
  return h->loadStatus;
}
int lpth_renderer_scene_loaded(void* r) { return ((RendererHandle*)r)->renderer->sceneLoaded() ? 1 : 0; }

int lpth_renderer_configure(void* r, uint32_t seed, int max_bounces, int bsdf_mode, int pipeline, int frames_in_flight) {
  return guarded([&] {
    auto* h = (RendererHandle*)r;
    h->renderer->setSeed(seed);
    h->renderer->setMaxTraceDepth(max_bounces);
    h->renderer->setBsdfMode(bsdf_mode);
    h->renderer->setPipeline(pipeline, frames_in_flight);
  });
}

int lpth_renderer_set_deferred_frames(void* r, uint32_t n) {
  return guarded([&] { ((RendererHandle*)r)->renderer->setDeferredFrames(n); });
}
int lpth_renderer_flush(void* r) {
  return guarded([&] { ((RendererHandle*)r)->renderer->flush(); });
}
int lpth_renderer_draw_frame(void* r) {
  return guarded([&] { ((RendererHandle*)r)->renderer->drawFrame(); });
}
int lpth_renderer_draw_frames(void* r, uint32_t n) {
  return guarded([&] { ((RendererHandle*)r)->renderer->drawFrames(n); });
}
int lpth_renderer_resize(void* r, int width, int height) {
  return guarded([&] { ((RendererHandle*)r)->renderer->resize(width, height); });
}
// axis 0..2 = translate X/Y/Z, 3..5 = rotate X/Y/Z (the key bindings of src/Main.cpp:57-93)
int lpth_renderer_move_camera(void* r, int axis, float amount) {
  return guarded([&] {
    auto tr = ((RendererHandle*)r)->renderer->cameraTransform();
    if (!tr) throw std::runtime_error("no scene loaded");
    switch (axis) {
      case 0: tr->translateX(amount); break;
      case 1: tr->translateY(amount); break;
      case 2: tr->translateZ(amount); break;
      case 3: tr->rotateX(amount); break;
      case 4: tr->rotateY(amount); break;
      case 5: tr->rotateZ(amount); break;
      default: throw std::runtime_error("bad axis");
    }
  });
}
uint32_t lpth_renderer_samples(void* r) { return ((RendererHandle*)r)->renderer->samples(); }
lpt_ctx* lpth_renderer_device(void* r) { return ((RendererHandle*)r)->renderer->device(); }
int lpth_renderer_read(void* r, float* rgba, size_t n_floats) {
  return guarded([&] {
    std::vector<float> img;
    ((RendererHandle*)r)->renderer->readAccumulation(img);
    if (img.size() != n_floats) throw std::runtime_error("destination size differs from the render size");
    std::memcpy(rgba, img.data(), img.size() * sizeof(float));
  });
}
int lpth_renderer_resolve(void* r, float* rgba, size_t n_floats) {
  return guarded([&] {
    std::vector<float> img;
    ((RendererHandle*)r)->renderer->resolve(img);
    if (img.size() != n_floats) throw std::runtime_error("destination size differs from the render size");
    std::memcpy(rgba, img.data(), img.size() * sizeof(float));
  });
}

// ---- frame-seed generator -----------------------------------------------------------------

// First n values of rand() after srand(seed), as the C library of the reference's platform
// produces them.
void lpth_glibc_rand(uint32_t seed, uint32_t* out, uint32_t n) {
  GlibcRand g(seed);
  for (uint32_t i = 0; i < n; ++i) out[i] = g.rand();
}

}  // extern "C"
