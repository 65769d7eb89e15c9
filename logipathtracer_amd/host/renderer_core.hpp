// RendererCore / RendererConfiguration: the host base class of the reference
// (include/RendererCore.hpp:13-117) for headless use on an MI355X.  The public surface
// (drawFrame, loadScene, the preDraw/postDraw/onSwapChainRecreate hooks, the
// configuration fields) is kept; the Vulkan instance, surface, swapchain and command
// buffers are replaced by one device context of the C-ABI layer (include/lpt.h).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/lpt.h"
#include "lsg_min.hpp"

namespace cppglfw {
// Headless stand-in for the window the reference renders into (src/Main.cpp:29-30):
// it only carries the framebuffer size.
class Window {
 public:
  Window(std::string title = "Renderer", int32_t width = 1280, int32_t height = 720)
      : title_(std::move(title)), width_(width), height_(height) {}
  int32_t width() const { return width_; }
  int32_t height() const { return height_; }
  const std::string& title() const { return title_; }
  void resize(int32_t w, int32_t h) { width_ = w; height_ = h; }
  bool shouldClose() const { return false; }

 private:
  std::string title_;
  int32_t width_, height_;
};
}  // namespace cppglfw

// include/RendererCore.hpp:13-27.  The extension / layer lists are accepted and ignored
// (there is no Vulkan here); `device` is new: the HIP device the renderer runs on.
struct RendererConfiguration {
  explicit RendererConfiguration(std::string windowTitle = "Renderer", int32_t windowWidth = 1280,
                                 int32_t windowHeight = 720, float renderScale = 1,
                                 std::vector<const char*> instanceExtensions = {},
                                 std::vector<const char*> deviceExtensions = {},
                                 std::vector<const char*> validationLayers = {"VK_LAYER_LUNARG_standard_validation"});

  std::string windowTitle;
  int32_t windowWidth;
  int32_t windowHeight;
  float renderScale;
  std::vector<const char*> instanceExtensions;
  std::vector<const char*> deviceExtensions;
  std::vector<const char*> validationLayers;
  int device = 0;
};

class RendererCore {
 public:
  // Throws std::runtime_error if the device context cannot be created
  // (the reference throws from its Vulkan set-up, src/RendererCore.cpp:118,121).
  explicit RendererCore(cppglfw::Window window, const RendererConfiguration& configuration);
  virtual ~RendererCore();
  RendererCore(const RendererCore&) = delete;
  RendererCore& operator=(const RendererCore&) = delete;

  // preDraw -> submit one frame -> postDraw (src/RendererCore.cpp:373-412).
  virtual void drawFrame();

  virtual void loadScene(const lsg::Ref<lsg::Scene>& scene) = 0;

  // The render-target size changed (the reference learns this from an out-of-date
  // swapchain, src/RendererCore.cpp:407-411).
  void resize(int32_t width, int32_t height);

  lpt_ctx* device() const { return device_; }
  uint32_t renderWidth() const { return renderWidth_; }
  uint32_t renderHeight() const { return renderHeight_; }

 protected:
  void recreateSwapChain();
  virtual void onSwapChainRecreate() = 0;
  virtual void preDraw();
  virtual void postDraw();
  // Replaces graphicsQueue_.submit(mainCmdBuffers_[imageIndex]) (src/RendererCore.cpp:388-398).
  virtual void submitFrame() = 0;
  // Throws std::runtime_error with the context's message if status != 0.
  void check(int status, const char* what) const;

  cppglfw::Window window_;
  lpt_ctx* device_ = nullptr;
  float renderScale;
  uint32_t renderWidth_ = 0, renderHeight_ = 0;  // swapchainImageExtent_ * renderScale
  size_t currentFrame_ = 0;
};
