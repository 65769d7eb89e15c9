#include "renderer_core.hpp"

#include <cmath>
#include <stdexcept>

RendererConfiguration::RendererConfiguration(std::string windowTitle, int32_t windowWidth, int32_t windowHeight,
                                             float renderScale, std::vector<const char*> instanceExtensions,
                                             std::vector<const char*> deviceExtensions,
                                             std::vector<const char*> validationLayers)
    : windowTitle(std::move(windowTitle)), windowWidth(windowWidth), windowHeight(windowHeight),
      renderScale(renderScale), instanceExtensions(std::move(instanceExtensions)),
      deviceExtensions(std::move(deviceExtensions)), validationLayers(std::move(validationLayers)) {}

RendererCore::RendererCore(cppglfw::Window window, const RendererConfiguration& configuration)
    : window_(std::move(window)), renderScale(configuration.renderScale) {
  int rc = lpt_create(configuration.device, &device_);
  if (rc != LPT_OK) throw std::runtime_error(std::string("Failed to create device context: ") + lpt_last_error(nullptr));
  renderWidth_ = (uint32_t)std::ceil((float)window_.width() * renderScale);
  renderHeight_ = (uint32_t)std::ceil((float)window_.height() * renderScale);
}

RendererCore::~RendererCore() { lpt_destroy(device_); }

void RendererCore::check(int status, const char* what) const {
  if (status != LPT_OK) throw std::runtime_error(std::string(what) + ": " + lpt_last_error(device_));
}

void RendererCore::drawFrame() {
  preDraw();
  submitFrame();
  ++currentFrame_;
  postDraw();
}

void RendererCore::resize(int32_t width, int32_t height) {
  window_.resize(width, height);
  recreateSwapChain();
}

void RendererCore::recreateSwapChain() {
  check(lpt_synchronize(device_), "lpt_synchronize");
  renderWidth_ = (uint32_t)std::ceil((float)window_.width() * renderScale);
  renderHeight_ = (uint32_t)std::ceil((float)window_.height() * renderScale);
  onSwapChainRecreate();
}

void RendererCore::preDraw() {}
void RendererCore::postDraw() {}
