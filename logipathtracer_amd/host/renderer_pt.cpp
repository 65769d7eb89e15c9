#include "renderer_pt.hpp"

#include <cstring>
#include <ctime>
#include <iostream>
#include <stdexcept>
#include <thread>

RendererPT::RendererPT(const cppglfw::Window& window, const RendererConfiguration& configuration)
    : RendererCore(window, configuration), rand_((uint32_t)std::time(nullptr)), sceneConverter_(device_) {
  std::memset(&ubo_, 0, sizeof ubo_);
  initializeAccumulationTexture();
  applyOptions();
}

void RendererPT::initializeAccumulationTexture() {
  check(lpt_resize(device_, (int)renderWidth_, (int)renderHeight_), "lpt_resize");
}

void RendererPT::applyOptions() {
  check(lpt_set_options(device_, maxTraceDepth_, bsdfMode_, pipeline_, framesInFlight_), "lpt_set_options");
}

void RendererPT::setDeferredFrames(uint32_t n) {
  Lock lock(frameMutex_);
  flush();
  deferredFrames_ = n == 0 ? 1u : n;
}

void RendererPT::flush() {
  Lock lock(frameMutex_);
  if (batching_ || pending_.empty()) return;
  // the queue is emptied first: a failing batch must not be submitted again by the next flush
  std::vector<PathTracerUBO> frames;
  frames.swap(pending_);
  check(lpt_render_frames(device_, frames.data(), (int)frames.size()), "lpt_render_frames");
}

void RendererPT::setMaxTraceDepth(int bounces) {
  Lock lock(frameMutex_);
  flush();  // frames queued so far were drawn under the old options
  maxTraceDepth_ = bounces;
  applyOptions();
}

void RendererPT::setBsdfMode(int mode) {
  Lock lock(frameMutex_);
  flush();
  bsdfMode_ = mode;
  applyOptions();
}

void RendererPT::setPipeline(int pipeline, int framesInFlight) {
  Lock lock(frameMutex_);
  flush();
  pipeline_ = pipeline;
  framesInFlight_ = framesInFlight;
  applyOptions();
}

void RendererPT::loadScene(const lsg::Ref<lsg::Scene>& scene) {
  sceneLoaded_ = false;  // from here on drawFrame does nothing (and does not wait for the mutex below)
  Lock lock(frameMutex_);  // a frame being drawn right now finishes first
  flush();  // frames of the previous scene still queued

  sceneConverter_.loadScene(scene);
  const auto& cameras = sceneConverter_.getCameras();
  if (cameras.empty()) {
    // the reference dereferences the missing camera here (src/RendererPT.cpp:36-41); fail cleanly instead
    sceneConverter_.reset();
    throw std::runtime_error("Loaded scene without cameras.");
  }

  selectedCameraTransform_ = cameras[0]->getComponent<lsg::Transform>();
  auto cameraData = cameras[0]->getComponent<lsg::PerspectiveCamera>();
  ubo_.fovY = cameraData->fov();
  const lsg::mat4 world = selectedCameraTransform_->worldMatrix();
  std::memcpy(ubo_.cameraWorldMatrix, world.m, sizeof ubo_.cameraWorldMatrix);
  ubo_.reset = 1u;
  sampleCount = 1;

  sceneLoaded_ = true;
}

void RendererPT::onSwapChainRecreate() {
  Lock lock(frameMutex_);
  // a new target size restarts accumulation (src/RendererPT.cpp:225-232); frames still queued belonged to
  // the old image and are dropped with it
  pending_.clear();
  initializeAccumulationTexture();
  sampleCount = 1;
  ubo_.reset = 1u;
}

void RendererPT::preDraw() {
  // src/RendererPT.cpp:574-592.  The accumulation image starts out undefined, so the
  // very first sample also has to store rather than add.
  if (selectedCameraTransform_->isWorldMatrixDirty()) {
    const lsg::mat4 world = selectedCameraTransform_->worldMatrix();
    std::memcpy(ubo_.cameraWorldMatrix, world.m, sizeof ubo_.cameraWorldMatrix);
    ubo_.reset = 1u;
    sampleCount = 1;
  } else {
    ubo_.reset = sampleCount == 1 ? 1u : 0u;
  }

  invSampleCount = 1.0f / (float)sampleCount;
  ubo_.seed[0] = rand_.rand();
  ubo_.seed[1] = rand_.rand();

  if (sampleCount == 1) startTime_ = std::chrono::high_resolution_clock::now();
}

void RendererPT::submitFrame() {
  if (batching_ || deferredFrames_ > 1) {
    pending_.push_back(ubo_);
    if (!batching_ && pending_.size() >= deferredFrames_) flush();
    return;
  }
  check(lpt_render_frame(device_, &ubo_), "lpt_render_frame");
}

void RendererPT::postDraw() {
  // src/RendererPT.cpp:594-607
  sampleCount++;
  if (verbose_ && sampleCount % 10 == 0) {
    std::cout << "Sample: " << sampleCount << std::endl;
    if (sampleCount % 100 == 0) {
      check(lpt_synchronize(device_), "lpt_synchronize");
      auto dt = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::high_resolution_clock::now() - startTime_).count() / 1000.0f;
      std::cout << "Samples per second: " << sampleCount / dt << std::endl;
    }
  }
}

bool RendererPT::lockForDrawing(std::unique_lock<std::recursive_mutex>& lock) {
  // A load holds the mutex for seconds and has cleared the flag before taking it: give up then, wait otherwise
  // (another thread reading the image or changing an option is done in a moment).
  while (!lock.try_lock()) {
    if (!sceneLoaded_) return false;
    std::this_thread::yield();
  }
  return sceneLoaded_;
}

void RendererPT::drawFrame() {
  if (!sceneLoaded_) return;
  std::unique_lock<std::recursive_mutex> lock(frameMutex_, std::defer_lock);
  if (lockForDrawing(lock)) RendererCore::drawFrame();
}

void RendererPT::drawFrames(uint32_t n) {
  if (!sceneLoaded_ || n == 0) return;
  std::unique_lock<std::recursive_mutex> lock(frameMutex_, std::defer_lock);
  if (!lockForDrawing(lock)) return;
  flush();
  batching_ = true;
  try {
    for (uint32_t i = 0; i < n; ++i) RendererCore::drawFrame();
  } catch (...) {
    batching_ = false;
    throw;
  }
  batching_ = false;
  flush();
}

void RendererPT::readAccumulation(std::vector<float>& rgba) {
  Lock lock(frameMutex_);
  flush();
  rgba.resize((size_t)renderWidth_ * renderHeight_ * 4);
  check(lpt_read_accum(device_, rgba.data()), "lpt_read_accum");
}

void RendererPT::resolve(std::vector<float>& rgba) {
  Lock lock(frameMutex_);
  flush();
  rgba.resize((size_t)renderWidth_ * renderHeight_ * 4);
  // the display pass divides by the number of samples accumulated so far
  float inv = samples() > 0 ? 1.0f / (float)samples() : 1.0f;
  check(lpt_resolve(device_, inv, rgba.data()), "lpt_resolve");
}
