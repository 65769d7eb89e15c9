// RendererPT: host driver of the path-tracing compute path, with the public methods of
// the reference class (include/RendererPT.h:13-19) and its per-frame contract
// (src/RendererPT.cpp:574-612): a camera change restarts accumulation, every frame gets
// a fresh pair of seeds and adds one sample per pixel.
#pragma once
#include <atomic>
#include <chrono>
#include <mutex>
#include <vector>

#include "glibc_rand.hpp"
#include "pt_scene_converter.hpp"
#include "renderer_core.hpp"

class RendererPT : public RendererCore {
 public:
  RendererPT(const cppglfw::Window& window, const RendererConfiguration& configuration);

  // May run on another thread than drawFrame (src/Main.cpp:45); drawFrame does nothing
  // until loading has finished.  The reference guards this with one atomic flag only
  // (include/RendererPT.h:87); here the flag says "nothing to draw" and a mutex keeps the two
  // threads out of the frame queue and the device context at the same time: loadScene holds it
  // from its first to its last step, every other entry point for its own duration, and a
  // drawFrame that finds it taken by a load returns at once.
  void loadScene(const lsg::Ref<lsg::Scene>& scene) override;

  void drawFrame() override;

  // ---- additions for headless, reproducible use -------------------------------------
  // n consecutive frames with the same per-frame contract, submitted as one batch.
  void drawFrames(uint32_t n);
  // Seed of the frame-seed generator; the reference uses srand(time(0)) (src/RendererPT.cpp:13).
  void setSeed(uint32_t seed) { rand_.srand(seed); }
  // MAX_TRACE_DEPTH and the BSDF family are compile-time choices in the reference shader
  // (shaders/path_tracing.comp:20,22).
  void setMaxTraceDepth(int bounces);
  void setBsdfMode(int mode);
  void setPipeline(int pipeline, int framesInFlight = 0);
  // Deferred submission for headless loops: drawFrame() keeps its per-frame contract (camera check, seeds,
  // sample count) but queues the frame; the queue is rendered as ONE batch when it holds `n` frames or when
  // anything looks at the image or changes what later frames depend on (readAccumulation, resolve, resize,
  // loadScene, the option setters).  A batch keeps the GPU filled where single frames pay a launch tail each:
  // the same pixels, about 2.7x the throughput at 1080p.  n = 1 (default) renders inside every drawFrame(),
  // as the reference's loop does (src/RendererCore.cpp:388-398).
  void setDeferredFrames(uint32_t n);
  // Renders the frames queued by deferred submission, if any.
  void flush();
  void setMeshBuilder(PTSceneConverter::MeshBuilder b) { sceneConverter_.setMeshBuilder(b); }
  void setVerbose(bool v) { verbose_ = v; sceneConverter_.setVerbose(v); }
  // Accumulated radiance, width*height RGBA32F, alpha = number of samples; row 0 first.
  void readAccumulation(std::vector<float>& rgba);
  // Display image of shaders/tex_to_quad.frag (tone-mapped, gamma-corrected, y-flipped).
  void resolve(std::vector<float>& rgba);
  // Samples accumulated per pixel so far.
  uint32_t samples() const { return sampleCount - 1; }
  // false from the start of loadScene until it has finished (loadScene may run on a second thread while
  // the draw loop spins, src/Main.cpp:45,96); drawFrame does nothing meanwhile
  bool sceneLoaded() const { return sceneLoaded_.load(); }
  const PTSceneConverter& sceneConverter() const { return sceneConverter_; }
  const lsg::Ref<lsg::Transform>& cameraTransform() const { return selectedCameraTransform_; }

 protected:
  void initializeAccumulationTexture();
  void onSwapChainRecreate() override;
  void preDraw() override;
  void postDraw() override;
  void submitFrame() override;

 private:
  // include/RendererPT.h:51-61; std140 block of shaders/path_tracing.comp:83-87
  using PathTracerUBO = lpt_ubo;
  void applyOptions();

  PathTracerUBO ubo_;
  std::vector<PathTracerUBO> pending_;

  uint32_t sampleCount = 1;
  float invSampleCount = 1;

  int maxTraceDepth_ = 10;
  int bsdfMode_ = LPT_BSDF_HEITZ;
  int pipeline_ = LPT_PIPELINE_AUTO;
  int framesInFlight_ = 0;
  bool verbose_ = false;
  bool batching_ = false;
  uint32_t deferredFrames_ = 1;

  GlibcRand rand_;
  std::chrono::time_point<std::chrono::high_resolution_clock> startTime_;

  std::atomic<bool> sceneLoaded_{false};
  // guards pending_, the frame counters and every call on device_; recursive: the setters, drawFrames and the
  // readers flush the queue from inside
  std::recursive_mutex frameMutex_;
  using Lock = std::lock_guard<std::recursive_mutex>;
  // takes frameMutex_ unless a scene is being loaded; false = nothing to draw now
  bool lockForDrawing(std::unique_lock<std::recursive_mutex>& lock);
  lsg::Ref<lsg::Transform> selectedCameraTransform_;
  PTSceneConverter sceneConverter_;
};
