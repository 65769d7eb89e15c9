// Host BVH builders under the names the reference's scene converter uses
// (lsg::bvh::BVHBuilder / SplitBVHBuilder, src/PTSceneConverter.cpp:97-101,134-138).
// Both produce a binary tree in depth-first order: node 0 is the root, a node's
// first child follows it directly.  Leaves hold a half-open range into
// getPrimitiveIndices().
#pragma once
#include <cstdint>
#include <vector>

#include "lsg_min.hpp"

namespace lsg {
namespace bvh {

struct uvec2 {
  uint32_t x = 0, y = 0;
};

template <typename T>
struct Node {
  AABB<T> bounds;
  bool is_leaf = false;
  uvec2 child_indices;  // inner: child node indices; leaf: [first, last) into the primitive order
};

template <typename T>
class BVH {
 public:
  const std::vector<Node<T>>& getNodes() const { return nodes_; }
  const std::vector<uint32_t>& getPrimitiveIndices() const { return primitiveIndices_; }
  const AABB<T>& getBounds() const { return bounds_; }
  uint32_t depth() const;

  std::vector<Node<T>> nodes_;
  std::vector<uint32_t> primitiveIndices_;
  AABB<T> bounds_;
};

struct BuildOptions {
  // Leaves: a list of at most minLeafSize primitives always becomes a leaf, one of more than maxLeafSize never does;
  // in between the surface-area heuristic decides (split if the best object split costs less than testing them all).
  // Measured on MI355X with these defaults against leaves of exactly four: +2.6 % traced segments/s on the Cornell box,
  // +1.0 % on a 400 k-triangle terrain scene (leaves of exactly two: +4.2 % / -1.6 %).
  uint32_t maxLeafSize = 4;
  uint32_t minLeafSize = 1;
  uint32_t bins = 16;           // SAH bins per axis
  float traversalCost = 1.5f;   // cost of visiting an inner node, relative to one primitive test
  // Spatial splits (SplitBVHBuilder only): a spatial split is tried when the overlap of
  // the best object split exceeds this fraction of the root's surface area.
  float splitAlpha = 1e-5f;
  bool spatialSplits = true;
  // Which child is visited first by the (ray-independent) traversal order: 0 as split,
  // 1 the child with the larger box, 2 the child with the smaller box.
  int childOrder = 0;
};

// Binned surface-area-heuristic build over axis-aligned boxes (object splits only).
template <typename T>
class BVHBuilder {
 public:
  explicit BVHBuilder(BuildOptions opt = {}) : opt_(opt) {}
  Ref<BVH<T>> process(const std::vector<AABB<T>>& boxes);

 private:
  BuildOptions opt_;
};

// SAH build over triangles that may also split triangles spatially (Stich et al.,
// "Spatial Splits in Bounding Volume Hierarchies", 2009): a triangle straddling a split
// plane is referenced from both sides, so getPrimitiveIndices() may repeat indices.
template <typename T>
class SplitBVHBuilder {
 public:
  explicit SplitBVHBuilder(BuildOptions opt = {}) : opt_(opt) {}
  Ref<BVH<T>> process(const Ref<TriangleAccessor<vec3>>& triangles);

 private:
  BuildOptions opt_;
};

}  // namespace bvh
}  // namespace lsg
