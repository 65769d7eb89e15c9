// PTSceneConverter: scene graph -> the four flattened buffers the path tracer consumes
// (objects, object BVH, de-indexed vertices in BVH leaf order, mesh BVHs).  Same role,
// method names and buffer layouts as include/PTSceneConverter.hpp of the reference; the
// Vulkan allocator / command pool / queue arguments are replaced by a device context of
// the C-ABI layer (include/lpt.h), which receives the buffers through lpt_upload_scene.
#pragma once
#include <cstdint>
#include <vector>

#include "../../include/lpt.h"
#include "lsg_bvh.hpp"
#include "lsg_min.hpp"

// include/PTSceneConverter.hpp:15-59 — byte layouts are those of the C-ABI structs.
using GPUObjectData = lpt_object;
using GPUVertex = lpt_vertex;
using GPUBVHNode = lpt_bvh_node;
static_assert(sizeof(GPUObjectData) == 208 && sizeof(GPUVertex) == 48 && sizeof(GPUBVHNode) == 48, "scene buffer layouts");

// include/GPUTexture.hpp:9-13 — image + sampler of one material texture, here as host data
// in the form lpt_upload_textures takes.
struct GPUTexture {
  lsg::Ref<lsg::Image> image;
  uint32_t magFilter = 1;                  // vk::Filter
  uint32_t addressModeU = 0, addressModeV = 0;  // vk::SamplerAddressMode
};

class PTSceneConverter {
 public:
  // Which builder stands in for lsg::bvh::SplitBVHBuilder at src/PTSceneConverter.cpp:97-98.
  enum class MeshBuilder {
    SplitBVH = 0,   // host SAH build with spatial splits
    BinnedSAH = 1,  // host SAH build, object splits only
    LBVH = 2,       // Morton-code build on the GPU (needs a device context)
    PLOC = 3        // GPU build by locally-ordered clustering: SAH-class quality (needs a device context)
  };

  // `device` may be null: buffers are then built on the host only.
  explicit PTSceneConverter(lpt_ctx* device = nullptr);

  void loadScene(const lsg::Ref<lsg::Scene>& scene);

  const std::vector<lsg::Ref<lsg::Object>>& getCameras() const;

  const std::vector<GPUObjectData>& getObjectDataBuffer() const;
  const std::vector<GPUBVHNode>& getObjectBvhNodesBuffer() const;
  const std::vector<GPUVertex>& getVerticesBuffer() const;
  const std::vector<GPUBVHNode>& getMeshBvhNodesBuffer() const;
  const std::vector<GPUTexture>& getTextures() const;

  void reset();

  void setMeshBuilder(MeshBuilder b) { meshBuilder_ = b; }
  void setBuildOptions(const lsg::bvh::BuildOptions& o) { buildOptions_ = o; }
  void setVerbose(bool v) { verbose_ = v; }

 protected:
  void copyToGPU();
  uint32_t copyTextureToGPU(const lsg::Ref<lsg::Texture>& texture);

 private:
  lpt_ctx* device_;
  MeshBuilder meshBuilder_ = MeshBuilder::SplitBVH;
  lsg::bvh::BuildOptions buildOptions_;
  bool verbose_ = false;

  std::vector<lsg::Ref<lsg::Object>> cameras_;
  std::vector<GPUObjectData> objectData_;
  std::vector<GPUBVHNode> objectBVHNodes_;
  std::vector<GPUVertex> vertices_;
  std::vector<GPUBVHNode> meshBVHNodes_;
  std::vector<GPUTexture> textures_;
};
