#include "lsg_bvh.hpp"

#include "../../include/lpt.h"

#include <algorithm>
#include <cmath>
#include <limits>

namespace lsg {
namespace bvh {

namespace {

using Box = AABB<float>;

struct PrimRef {
  uint32_t prim;
  Box box;
};

inline float axisOf(const vec3& v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }
inline void setAxis(vec3& v, int a, float f) { (a == 0 ? v.x : (a == 1 ? v.y : v.z)) = f; }

inline Box intersectBoxes(const Box& a, const Box& b) {
  vec3 mn{std::fmax(a.min().x, b.min().x), std::fmax(a.min().y, b.min().y), std::fmax(a.min().z, b.min().z)};
  vec3 mx{std::fmin(a.max().x, b.max().x), std::fmin(a.max().y, b.max().y), std::fmin(a.max().z, b.max().z)};
  return Box(mn, mx);
}

struct Split {
  float cost = std::numeric_limits<float>::infinity();
  int axis = -1;
  float pos = 0.0f;  // object split: bin boundary index; spatial split: plane position
  int bin = 0;
  Box left, right;
};

constexpr int kMaxDepth = 40;
// Beyond kMaxDepth a node stops looking for good splits, but it still halves its list until it fits a
// leaf the device layer accepts.
constexpr uint32_t kHardLeafLimit = LPT_MAX_LEAF_PRIMITIVES;

class Builder {
 public:
  Builder(const BuildOptions& opt, const TriangleAccessor<vec3>* tris) : opt_(opt), tris_(tris) {
    bins_ = (int)std::max<uint32_t>(2, std::min<uint32_t>(opt.bins, 64));
    opt_.maxLeafSize = std::max<uint32_t>(1, std::min<uint32_t>(opt.maxLeafSize, kHardLeafLimit));
    opt_.minLeafSize = std::max<uint32_t>(1, std::min<uint32_t>(opt.minLeafSize, opt_.maxLeafSize));
  }

  Ref<BVH<float>> run(std::vector<PrimRef>&& refs) {
    auto out = makeRef<BVH<float>>();
    bvh_ = out.get();
    Box all;
    for (auto& r : refs) all.expand(r.box);
    out->bounds_ = all;
    rootArea_ = all.area();
    refBudget_ = refs.size() + refs.size() / 2 + 16;  // spatial splits may add at most 50 % references
    refTotal_ = refs.size();
    if (refs.empty()) {
      Node<float> n;
      n.bounds = Box({0, 0, 0}, {0, 0, 0});
      n.is_leaf = true;
      out->nodes_.push_back(n);
      return out;
    }
    build(std::move(refs), all, 0);
    return out;
  }

 private:
  BuildOptions opt_;
  const TriangleAccessor<vec3>* tris_;
  BVH<float>* bvh_ = nullptr;
  int bins_ = 16;
  float rootArea_ = 0.0f;
  size_t refBudget_ = 0, refTotal_ = 0;

  uint32_t makeLeaf(uint32_t idx, const std::vector<PrimRef>& refs) {
    Node<float>& n = bvh_->nodes_[idx];
    n.is_leaf = true;
    n.child_indices.x = (uint32_t)bvh_->primitiveIndices_.size();
    for (auto& r : refs) bvh_->primitiveIndices_.push_back(r.prim);
    n.child_indices.y = (uint32_t)bvh_->primitiveIndices_.size();
    return idx;
  }

  Split bestObjectSplit(const std::vector<PrimRef>& refs, const Box& nodeBox) const {
    Split best;
    Box cb;
    for (auto& r : refs) cb.expand(r.box.center());
    const float nodeArea = std::max(nodeBox.area(), 1e-30f);
    std::vector<Box> binBox(bins_);
    std::vector<uint32_t> binCnt(bins_);
    std::vector<float> rightArea(bins_);
    for (int axis = 0; axis < 3; ++axis) {
      float lo = axisOf(cb.min(), axis), hi = axisOf(cb.max(), axis);
      if (!(hi > lo)) continue;
      float scale = (float)bins_ / (hi - lo);
      for (int b = 0; b < bins_; ++b) { binBox[b].reset(); binCnt[b] = 0; }
      for (auto& r : refs) {
        int b = std::min(bins_ - 1, std::max(0, (int)((axisOf(r.box.center(), axis) - lo) * scale)));
        binBox[b].expand(r.box);
        ++binCnt[b];
      }
      Box acc;
      for (int b = bins_ - 1; b > 0; --b) {
        acc.expand(binBox[b]);
        rightArea[b] = acc.area();
      }
      acc.reset();
      uint32_t nl = 0;
      for (int b = 0; b + 1 < bins_; ++b) {
        if (binCnt[b]) acc.expand(binBox[b]);
        nl += binCnt[b];
        uint32_t nr = (uint32_t)refs.size() - nl;
        if (nl == 0 || nr == 0) continue;
        float cost = opt_.traversalCost + (acc.area() * (float)nl + rightArea[b + 1] * (float)nr) / nodeArea;
        if (cost < best.cost) {
          best.cost = cost;
          best.axis = axis;
          best.bin = b;
          best.pos = lo + (float)(b + 1) / scale;
        }
      }
    }
    if (best.axis >= 0) {  // recompute the two child boxes of the winner
      Box cb2;
      for (auto& r : refs) cb2.expand(r.box.center());
      float lo = axisOf(cb2.min(), best.axis), hi = axisOf(cb2.max(), best.axis);
      float scale = (float)bins_ / (hi - lo);
      for (auto& r : refs) {
        int b = std::min(bins_ - 1, std::max(0, (int)((axisOf(r.box.center(), best.axis) - lo) * scale)));
        (b <= best.bin ? best.left : best.right).expand(r.box);
      }
    }
    return best;
  }

  // Box of the part of triangle `prim` inside the slab lo <= axis <= hi, clipped to `within`.
  Box clipTriangle(uint32_t prim, int axis, float lo, float hi, const Box& within) const {
    auto T = (*tris_)[prim];
    vec3 poly[8], tmp[8];
    int n = 3;
    poly[0] = T.a(); poly[1] = T.b(); poly[2] = T.c();
    for (int pass = 0; pass < 2; ++pass) {
      const float plane = pass == 0 ? lo : hi;
      const float sign = pass == 0 ? 1.0f : -1.0f;
      int m = 0;
      for (int i = 0; i < n; ++i) {
        const vec3& p = poly[i];
        const vec3& q = poly[(i + 1) % n];
        float dp = sign * (axisOf(p, axis) - plane), dq = sign * (axisOf(q, axis) - plane);
        if (dp >= 0.0f) tmp[m++] = p;
        if ((dp > 0.0f && dq < 0.0f) || (dp < 0.0f && dq > 0.0f)) {
          float t = dp / (dp - dq);
          vec3 x = p + (q - p) * t;
          setAxis(x, axis, plane);
          tmp[m++] = x;
        }
      }
      n = m;
      for (int i = 0; i < n; ++i) poly[i] = tmp[i];
      if (n == 0) break;
    }
    Box b;
    for (int i = 0; i < n; ++i) b.expand(poly[i]);
    if (n > 0) {
      // the clip points are rounded; widen the other two axes by one ulp so the box still
      // encloses the exact clipped polygon
      vec3 mn = b.min(), mx = b.max();
      for (int a = 0; a < 3; ++a) {
        if (a == axis) continue;
        setAxis(mn, a, std::nextafterf(axisOf(mn, a), -INFINITY));
        setAxis(mx, a, std::nextafterf(axisOf(mx, a), INFINITY));
      }
      b = Box(mn, mx);
    }
    return intersectBoxes(b, within);
  }

  Split bestSpatialSplit(const std::vector<PrimRef>& refs, const Box& nodeBox) const {
    Split best;
    const float nodeArea = std::max(nodeBox.area(), 1e-30f);
    std::vector<Box> binBox(bins_);
    std::vector<uint32_t> enter(bins_), leave(bins_);
    std::vector<float> rightArea(bins_);
    for (int axis = 0; axis < 3; ++axis) {
      float lo = axisOf(nodeBox.min(), axis), hi = axisOf(nodeBox.max(), axis);
      if (!(hi > lo)) continue;
      float width = (hi - lo) / (float)bins_;
      float scale = 1.0f / width;
      for (int b = 0; b < bins_; ++b) { binBox[b].reset(); enter[b] = leave[b] = 0; }
      for (auto& r : refs) {
        int b0 = std::min(bins_ - 1, std::max(0, (int)((axisOf(r.box.min(), axis) - lo) * scale)));
        int b1 = std::min(bins_ - 1, std::max(b0, (int)((axisOf(r.box.max(), axis) - lo) * scale)));
        ++enter[b0];
        ++leave[b1];
        if (b0 == b1) {
          binBox[b0].expand(r.box);
        } else {
          for (int b = b0; b <= b1; ++b) {
            Box c = clipTriangle(r.prim, axis, lo + width * (float)b, lo + width * (float)(b + 1), r.box);
            if (c.min().x <= c.max().x && c.min().y <= c.max().y && c.min().z <= c.max().z) binBox[b].expand(c);
          }
        }
      }
      Box acc;
      for (int b = bins_ - 1; b > 0; --b) {
        acc.expand(binBox[b]);
        rightArea[b] = acc.area();
      }
      acc.reset();
      uint32_t nl = 0, nr = (uint32_t)refs.size();
      for (int b = 0; b + 1 < bins_; ++b) {
        acc.expand(binBox[b]);
        nl += enter[b];
        nr -= leave[b];
        if (nl == 0 || nr == 0) continue;
        float cost = opt_.traversalCost + (acc.area() * (float)nl + rightArea[b + 1] * (float)nr) / nodeArea;
        if (cost < best.cost) {
          best.cost = cost;
          best.axis = axis;
          best.bin = b;
          best.pos = lo + width * (float)(b + 1);
        }
      }
    }
    return best;
  }

  uint32_t build(std::vector<PrimRef>&& refs, const Box& box, int depth) {
    const uint32_t idx = (uint32_t)bvh_->nodes_.size();
    bvh_->nodes_.emplace_back();
    bvh_->nodes_[idx].bounds = box;
    const uint32_t n = (uint32_t)refs.size();
    if (n <= opt_.minLeafSize || (depth >= kMaxDepth && n <= kHardLeafLimit)) return makeLeaf(idx, refs);

    Split obj = bestObjectSplit(refs, box);
    // Up to maxLeafSize primitives may stay together: they do when no object split is cheaper than testing them all
    // (surface-area heuristic, costs in units of one primitive test).
    if (n <= opt_.maxLeafSize && !(obj.axis >= 0 && obj.cost < (float)n)) return makeLeaf(idx, refs);
    Split spa;
    if (tris_ && opt_.spatialSplits && refTotal_ < refBudget_ && depth < kMaxDepth - 8) {
      float overlap = 0.0f;
      if (obj.axis >= 0) overlap = intersectBoxes(obj.left, obj.right).area();
      if (obj.axis < 0 || overlap / std::max(rootArea_, 1e-30f) > opt_.splitAlpha) spa = bestSpatialSplit(refs, box);
    }

    std::vector<PrimRef> L, R;
    Box lb, rb;
    if (spa.axis >= 0 && spa.cost < obj.cost) {
      for (auto& r : refs) {
        float mn = axisOf(r.box.min(), spa.axis), mx = axisOf(r.box.max(), spa.axis);
        if (mx <= spa.pos) {
          L.push_back(r);
          lb.expand(r.box);
        } else if (mn >= spa.pos) {
          R.push_back(r);
          rb.expand(r.box);
        } else {
          Box cl = clipTriangle(r.prim, spa.axis, axisOf(box.min(), spa.axis), spa.pos, r.box);
          Box cr = clipTriangle(r.prim, spa.axis, spa.pos, axisOf(box.max(), spa.axis), r.box);
          bool okL = cl.min().x <= cl.max().x && cl.min().y <= cl.max().y && cl.min().z <= cl.max().z;
          bool okR = cr.min().x <= cr.max().x && cr.min().y <= cr.max().y && cr.min().z <= cr.max().z;
          if (okL) { L.push_back({r.prim, cl}); lb.expand(cl); }
          if (okR) { R.push_back({r.prim, cr}); rb.expand(cr); }
          if (!okL && !okR) { L.push_back(r); lb.expand(r.box); }
          if (okL && okR) ++refTotal_;
        }
      }
      // a spatial split that does not separate anything would recurse forever
      if (L.empty() || R.empty() || (L.size() == n && R.size() == n)) {
        L.clear();
        R.clear();
        lb.reset();
        rb.reset();
      }
    }
    if (L.empty() && R.empty() && obj.axis >= 0) {
      Box cb;
      for (auto& r : refs) cb.expand(r.box.center());
      float lo = axisOf(cb.min(), obj.axis), hi = axisOf(cb.max(), obj.axis);
      float scale = (float)bins_ / (hi - lo);
      for (auto& r : refs) {
        int b = std::min(bins_ - 1, std::max(0, (int)((axisOf(r.box.center(), obj.axis) - lo) * scale)));
        if (b <= obj.bin) { L.push_back(r); lb.expand(r.box); } else { R.push_back(r); rb.expand(r.box); }
      }
    }
    if (L.empty() || R.empty()) {
      // all centroids coincide (or binning failed): split the list in half
      L.clear(); R.clear(); lb.reset(); rb.reset();
      for (uint32_t i = 0; i < n; ++i) {
        if (i < n / 2) { L.push_back(refs[i]); lb.expand(refs[i].box); } else { R.push_back(refs[i]); rb.expand(refs[i].box); }
      }
    }
    std::vector<PrimRef>().swap(refs);
    // The path tracer walks child y before child x, whatever the ray (shaders/path_tracing.comp:
    // 159-171 pushes x then y and pops y).  childOrder decides which half becomes y.
    bool swapHalves = false;
    if (opt_.childOrder == 1) swapHalves = lb.area() > rb.area();       // larger box visited first
    else if (opt_.childOrder == 2) swapHalves = lb.area() < rb.area();  // smaller box visited first
    if (swapHalves) {
      std::swap(L, R);
      std::swap(lb, rb);
    }
    // the half visited first is stored right behind its parent
    uint32_t yi = build(std::move(R), rb, depth + 1);
    uint32_t xi = build(std::move(L), lb, depth + 1);
    Node<float>& node = bvh_->nodes_[idx];
    node.is_leaf = false;
    node.child_indices.x = xi;
    node.child_indices.y = yi;
    return idx;
  }
};

}  // namespace

template <>
uint32_t BVH<float>::depth() const {
  if (nodes_.empty()) return 0;
  uint32_t best = 0;
  std::vector<std::pair<uint32_t, uint32_t>> st{{0u, 1u}};
  while (!st.empty()) {
    auto [i, d] = st.back();
    st.pop_back();
    best = std::max(best, d);
    if (!nodes_[i].is_leaf) {
      st.push_back({nodes_[i].child_indices.x, d + 1});
      st.push_back({nodes_[i].child_indices.y, d + 1});
    }
  }
  return best;
}

template <>
Ref<BVH<float>> BVHBuilder<float>::process(const std::vector<AABB<float>>& boxes) {
  std::vector<PrimRef> refs(boxes.size());
  for (size_t i = 0; i < boxes.size(); ++i) refs[i] = {(uint32_t)i, boxes[i]};
  BuildOptions o = opt_;
  o.spatialSplits = false;
  return Builder(o, nullptr).run(std::move(refs));
}

template <>
Ref<BVH<float>> SplitBVHBuilder<float>::process(const Ref<TriangleAccessor<vec3>>& triangles) {
  const size_t n = triangles->count();
  std::vector<PrimRef> refs(n);
  for (size_t i = 0; i < n; ++i) {
    auto T = (*triangles)[i];
    Box b;
    b.expand(T.a());
    b.expand(T.b());
    b.expand(T.c());
    refs[i] = {(uint32_t)i, b};
  }
  return Builder(opt_, triangles.get()).run(std::move(refs));
}

}  // namespace bvh
}  // namespace lsg
