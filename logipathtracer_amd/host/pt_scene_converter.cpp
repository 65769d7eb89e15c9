#include "pt_scene_converter.hpp"

#include <algorithm>
#include <atomic>
#include <cstring>
#include <iostream>
#include <stdexcept>
#include <string>
#include <thread>

namespace {

constexpr uint32_t kNoTexture = 0xFFFFFFFFu;

GPUBVHNode packNode(const lsg::bvh::Node<float>& n) {
  GPUBVHNode g;
  std::memset(&g, 0, sizeof g);
  g.minCorner[0] = n.bounds.min().x; g.minCorner[1] = n.bounds.min().y; g.minCorner[2] = n.bounds.min().z;
  g.maxCorner[0] = n.bounds.max().x; g.maxCorner[1] = n.bounds.max().y; g.maxCorner[2] = n.bounds.max().z;
  g.isLeaf = n.is_leaf ? 1u : 0u;
  g.indices[0] = n.child_indices.x;
  g.indices[1] = n.child_indices.y;
  return g;
}

GPUVertex packVertex(const lsg::vec3& p, const lsg::vec3& n, const lsg::vec2& uv) {
  GPUVertex v;
  std::memset(&v, 0, sizeof v);
  v.position[0] = p.x; v.position[1] = p.y; v.position[2] = p.z;
  v.normal[0] = n.x; v.normal[1] = n.y; v.normal[2] = n.z;
  v.uv[0] = uv.x; v.uv[1] = uv.y;
  return v;
}

}  // namespace

PTSceneConverter::PTSceneConverter(lpt_ctx* device) : device_(device) {}

void PTSceneConverter::loadScene(const lsg::Ref<lsg::Scene>& scene) {
  reset();
  if (!scene) throw std::runtime_error("PTSceneConverter::loadScene: null scene");

  // One job per (object, sub-mesh): the object record is filled while walking the graph (which
  // also fixes the order of the textures); the mesh trees are then built on several host threads
  // and appended in the walk's order, so the buffers do not depend on the number of threads.
  struct Job {
    GPUObjectData od;
    lsg::mat4 world;
    lsg::Ref<lsg::Geometry> geometry;
    std::vector<GPUBVHNode> nodes;
    std::vector<GPUVertex> vertices;
    lsg::AABB<float> localBounds;
    std::string error;
  };
  std::vector<Job> jobs;

  auto visit = [&](const lsg::Ref<lsg::Object>& object) -> bool {
    lsg::mat4 world;  // identity when the object carries no transform
    if (auto tr = object->getComponent<lsg::Transform>()) world = tr->worldMatrix();

    if (object->getComponent<lsg::PerspectiveCamera>()) cameras_.push_back(object);

    auto mesh = object->getComponent<lsg::Mesh>();
    if (!mesh) return true;
    if (verbose_) std::cout << "Building BVH for object " << object->name() << "..." << std::endl;

    for (const auto& sub : mesh->subMeshes()) {
      auto material = lsg::dynamicRefCast<lsg::MetallicRoughnessMaterial>(sub->material());
      if (!material) {
        if (verbose_) std::cout << "Unknown material. Skipping submesh." << std::endl;
        continue;
      }
      Job job;
      GPUObjectData& od = job.od;
      std::memset(&od, 0, sizeof od);
      const lsg::mat4 inv = lsg::inverse(world);
      std::memcpy(od.worldMatrix, world.m, sizeof od.worldMatrix);
      std::memcpy(od.worldMatrixInverse, inv.m, sizeof od.worldMatrixInverse);
      const lsg::vec4& bc = material->baseColorFactor();
      od.baseColorFactor[0] = bc.x; od.baseColorFactor[1] = bc.y; od.baseColorFactor[2] = bc.z; od.baseColorFactor[3] = bc.w;
      const lsg::vec3& em = material->emissiveFactor();
      od.emissionFactor[0] = em.x; od.emissionFactor[1] = em.y; od.emissionFactor[2] = em.z;
      od.metallicFactor = material->metallicFactor();
      od.roughnessFactor = material->roughnessFactor();
      od.transmissionFactor = material->transmissionFactor();
      od.ior = material->ior();
      // one GPU texture per material slot and object, no sharing (src/PTSceneConverter.cpp:69-84)
      od.colorTexture = material->baseColorTex() ? copyTextureToGPU(material->baseColorTex()) : kNoTexture;
      od.emissionTexture = material->emissiveTex() ? copyTextureToGPU(material->emissiveTex()) : kNoTexture;
      od.metallicRoughnessTexture =
          material->metallicRoughnessTex() ? copyTextureToGPU(material->metallicRoughnessTex()) : kNoTexture;
      od.transmissionTexture =
          material->transmissionTexture() ? copyTextureToGPU(material->transmissionTexture()) : kNoTexture;
      od.normalTexture = material->normalTex() ? copyTextureToGPU(material->normalTex()) : kNoTexture;
      job.world = world;
      job.geometry = sub->geometry();
      jobs.push_back(std::move(job));
    }
    return true;
  };
  for (const auto& root : scene->children()) root->traverseDown(visit);

  // mesh tree + de-indexed, interleaved vertices in leaf order (src/PTSceneConverter.cpp:97-120)
  auto buildJob = [&](Job& job) {
    auto posAcc = job.geometry->getTrianglePositionAccessor();
    auto nrmAcc = job.geometry->getTriangleNormalAccessor();
    auto uvAcc = job.geometry->hasUv(0u) ? job.geometry->getTriangleUVAccessor(0u) : nullptr;
    const uint32_t nTris = (uint32_t)posAcc->count();
    std::vector<uint32_t> order;
    if (meshBuilder_ == MeshBuilder::LBVH || meshBuilder_ == MeshBuilder::PLOC) {
      if (!device_) throw std::runtime_error("PTSceneConverter: the GPU builders need a device context");
      std::vector<float> corners(9 * (size_t)nTris);
      for (uint32_t t = 0; t < nTris; ++t) {
        auto T = (*posAcc)[t];
        const lsg::vec3 c[3] = {T.a(), T.b(), T.c()};
        for (int k = 0; k < 3; ++k) {
          corners[9 * (size_t)t + 3 * k + 0] = c[k].x;
          corners[9 * (size_t)t + 3 * k + 1] = c[k].y;
          corners[9 * (size_t)t + 3 * k + 2] = c[k].z;
          job.localBounds.expand(c[k]);
        }
      }
      job.nodes.resize(nTris ? 2 * (size_t)nTris : 1);
      order.resize(nTris);
      uint32_t nNodes = 0;
      const int which = meshBuilder_ == MeshBuilder::PLOC ? LPT_BVH_PLOC : LPT_BVH_LBVH;
      int rc = lpt_build_mesh_bvh(device_, which, corners.data(), nTris, job.nodes.data(), &nNodes, order.data());
      if (rc != LPT_OK) throw std::runtime_error(std::string("lpt_build_mesh_bvh: ") + lpt_last_error(device_));
      job.nodes.resize(nNodes);
    } else {
      lsg::bvh::BuildOptions opt = buildOptions_;
      opt.spatialSplits = meshBuilder_ == MeshBuilder::SplitBVH;
      lsg::bvh::SplitBVHBuilder<float> builder(opt);
      auto bvh = builder.process(posAcc);
      for (const auto& node : bvh->getNodes()) job.nodes.push_back(packNode(node));
      order = bvh->getPrimitiveIndices();
      job.localBounds = bvh->getBounds();
    }
    job.vertices.reserve(3 * order.size());
    for (uint32_t idx : order) {
      auto P = (*posAcc)[idx];
      auto N = (*nrmAcc)[idx];
      if (uvAcc) {
        auto U = (*uvAcc)[idx];
        job.vertices.push_back(packVertex(P.a(), N.a(), U.a()));
        job.vertices.push_back(packVertex(P.b(), N.b(), U.b()));
        job.vertices.push_back(packVertex(P.c(), N.c(), U.c()));
      } else {
        job.vertices.push_back(packVertex(P.a(), N.a(), {}));
        job.vertices.push_back(packVertex(P.b(), N.b(), {}));
        job.vertices.push_back(packVertex(P.c(), N.c(), {}));
      }
    }
  };
  // the GPU builder uses one device context: serial; the host builders run one job per thread at a time
  const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  const unsigned nThreads = (meshBuilder_ == MeshBuilder::LBVH || meshBuilder_ == MeshBuilder::PLOC) ? 1u : std::min<unsigned>(hw, (unsigned)std::max<size_t>(1, jobs.size()));
  std::atomic<size_t> nextJob{0};
  auto worker = [&]() {
    for (size_t j; (j = nextJob.fetch_add(1)) < jobs.size();) {
      try {
        buildJob(jobs[j]);
      } catch (const std::exception& e) {
        jobs[j].error = e.what()[0] ? e.what() : "mesh conversion failed";
      }
    }
  };
  if (nThreads <= 1) {
    worker();
  } else {
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < nThreads; ++t) pool.emplace_back(worker);
    for (auto& t : pool) t.join();
  }

  std::vector<lsg::AABB<float>> worldBoxes;       // one per sub-mesh instance
  std::vector<GPUObjectData> pendingObjects;      // in scene-graph order, before BVH ordering
  for (Job& job : jobs) {
    if (!job.error.empty()) throw std::runtime_error(job.error);
    // element indices, not byte offsets (src/PTSceneConverter.cpp:87-88)
    job.od.bvhOffset = (uint32_t)meshBVHNodes_.size();
    job.od.verticesOffset = (uint32_t)vertices_.size();
    meshBVHNodes_.insert(meshBVHNodes_.end(), job.nodes.begin(), job.nodes.end());
    vertices_.insert(vertices_.end(), job.vertices.begin(), job.vertices.end());
    worldBoxes.push_back(job.localBounds.transform(job.world));
    pendingObjects.push_back(job.od);
    job.nodes = {};
    job.vertices = {};
  }

  if (pendingObjects.empty()) throw std::runtime_error("PTSceneConverter: the scene contains no renderable geometry");

  // object tree; objects are stored in the order of its leaves (src/PTSceneConverter.cpp:134-142)
  lsg::bvh::BuildOptions topOpt = buildOptions_;
  topOpt.maxLeafSize = topOpt.minLeafSize = 1;
  lsg::bvh::BVHBuilder<float> topBuilder(topOpt);
  auto top = topBuilder.process(worldBoxes);
  for (const auto& node : top->getNodes()) objectBVHNodes_.push_back(packNode(node));
  for (uint32_t idx : top->getPrimitiveIndices()) objectData_.push_back(pendingObjects[idx]);

  copyToGPU();
}

// src/PTSceneConverter.cpp:243-389.  The sampler's magnification filter and address modes are
// taken over as the reference does, including its use of wrappingW() for the V axis (:359);
// a texture without sampler gets linear filtering and repeat (:371-376).
uint32_t PTSceneConverter::copyTextureToGPU(const lsg::Ref<lsg::Texture>& texture) {
  lsg::Ref<lsg::Image> image = texture->image();
  if (!image || image->width() == 0 || image->height() == 0)
    throw std::runtime_error("PTSceneConverter: texture without image data");
  if (textures_.size() >= LPT_MAX_TEXTURES)
    throw std::runtime_error("PTSceneConverter: more than 512 textures (sampler2D textures[512])");
  GPUTexture gpuTexture;
  gpuTexture.image = image;
  if (lsg::Ref<lsg::Sampler> sampler = texture->sampler()) {
    gpuTexture.magFilter = sampler->magFilter();
    gpuTexture.addressModeU = sampler->wrappingU();
    gpuTexture.addressModeV = sampler->wrappingW();
  }
  textures_.push_back(gpuTexture);
  return (uint32_t)textures_.size() - 1u;
}

void PTSceneConverter::copyToGPU() {
  if (!device_) return;
  std::vector<lpt_texture> tex(textures_.size());
  for (size_t i = 0; i < tex.size(); ++i)
    tex[i] = lpt_texture{textures_[i].image->rawPixelData(), textures_[i].image->width(), textures_[i].image->height(),
                         textures_[i].magFilter, textures_[i].addressModeU, textures_[i].addressModeV};
  int rc = lpt_upload_textures(device_, tex.data(), (uint32_t)tex.size());
  if (rc != LPT_OK) throw std::runtime_error(std::string("lpt_upload_textures: ") + lpt_last_error(device_));
  rc = lpt_upload_scene(device_, objectData_.data(), (uint32_t)objectData_.size(), objectBVHNodes_.data(),
                        (uint32_t)objectBVHNodes_.size(), vertices_.data(), (uint32_t)vertices_.size(),
                        meshBVHNodes_.data(), (uint32_t)meshBVHNodes_.size());
  if (rc != LPT_OK) throw std::runtime_error(std::string("lpt_upload_scene: ") + lpt_last_error(device_));
}

const std::vector<lsg::Ref<lsg::Object>>& PTSceneConverter::getCameras() const { return cameras_; }
const std::vector<GPUObjectData>& PTSceneConverter::getObjectDataBuffer() const { return objectData_; }
const std::vector<GPUBVHNode>& PTSceneConverter::getObjectBvhNodesBuffer() const { return objectBVHNodes_; }
const std::vector<GPUVertex>& PTSceneConverter::getVerticesBuffer() const { return vertices_; }
const std::vector<GPUBVHNode>& PTSceneConverter::getMeshBvhNodesBuffer() const { return meshBVHNodes_; }
const std::vector<GPUTexture>& PTSceneConverter::getTextures() const { return textures_; }

void PTSceneConverter::reset() {
  cameras_.clear();
  objectData_.clear();
  objectBVHNodes_.clear();
  vertices_.clear();
  meshBVHNodes_.clear();
  textures_.clear();
}
