// Linear BVH construction on the GPU for one triangle mesh:
//   triangle boxes + mesh bounds -> 30-bit Morton codes of the box centres ->
//   radix sort (rocPRIM) -> radix-tree hierarchy (Karras, "Maximizing Parallelism in the
//   Construction of BVHs, Octrees, and k-d Trees", 2012) -> bottom-up box fitting.
// The fitted tree is then written out on the host in depth-first order as interchange
// nodes (include/lpt.h: lpt_bvh_node), with subtrees of at most kLeafSize triangles
// collapsed into leaves.  Stands in for lsg::bvh::SplitBVHBuilder at
// src/PTSceneConverter.cpp:97-101 of the reference.
#include <hip/hip_runtime.h>
#include <string.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "lpt_internal.h"

namespace {

constexpr uint32_t kLeafSize = 4;

#define LB_HIP(ctx, call)                                                                      \
  do {                                                                                         \
    hipError_t e_ = (call);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      release();                                                                               \
      return lpt_internal_fail(ctx, LPT_ERR_HIP, (std::string(#call) + ": " + hipGetErrorString(e_)).c_str()); \
    }                                                                                          \
  } while (0)

// Order-preserving float <-> uint map, so box corners can be reduced with integer atomics.
__device__ __forceinline__ uint32_t ordered(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unordered(uint32_t u) {
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
}

// Per-triangle boxes; mesh bounds reduced per wave, then one atomic per wave and component.
__global__ void k_tri_boxes(const float* __restrict__ pos, uint32_t n, float* __restrict__ boxMin,
                            float* __restrict__ boxMax, uint32_t* bounds /* [0..2] min, [3..5] max, ordered */) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  if (i < n) {
    const float* p = pos + 9 * (size_t)i;
    for (int k = 0; k < 3; ++k) {
      mn[k] = fminf(fminf(p[k], p[3 + k]), p[6 + k]);
      mx[k] = fmaxf(fmaxf(p[k], p[3 + k]), p[6 + k]);
      boxMin[3 * (size_t)i + k] = mn[k];
      boxMax[3 * (size_t)i + k] = mx[k];
    }
  }
  for (int k = 0; k < 3; ++k) {
    float a = mn[k], b = mx[k];
    for (int off = 32; off > 0; off >>= 1) {
      a = fminf(a, __shfl_xor(a, off));
      b = fmaxf(b, __shfl_xor(b, off));
    }
    if ((threadIdx.x & 63) == 0) {
      atomicMin(&bounds[k], ordered(a));
      atomicMax(&bounds[3 + k], ordered(b));
    }
  }
}

__device__ __forceinline__ uint32_t spread3(uint32_t v) {  // 10 bits -> every third bit
  v = (v * 0x00010001u) & 0xFF0000FFu;
  v = (v * 0x00000101u) & 0x0F00F00Fu;
  v = (v * 0x00000011u) & 0xC30C30C3u;
  v = (v * 0x00000005u) & 0x49249249u;
  return v;
}

__global__ void k_morton(const float* __restrict__ boxMin, const float* __restrict__ boxMax, uint32_t n,
                         const uint32_t* __restrict__ bounds, uint32_t* __restrict__ codes, uint32_t* __restrict__ ids) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t q[3];
  for (int k = 0; k < 3; ++k) {
    float lo = unordered(bounds[k]), hi = unordered(bounds[3 + k]);
    float c = 0.5f * (boxMin[3 * (size_t)i + k] + boxMax[3 * (size_t)i + k]);
    float ext = hi - lo;
    float t = ext > 0.0f ? (c - lo) / ext : 0.0f;
    t = fminf(fmaxf(t * 1024.0f, 0.0f), 1023.0f);
    q[k] = (uint32_t)t;
  }
  codes[i] = (spread3(q[0]) << 2) | (spread3(q[1]) << 1) | spread3(q[2]);
  ids[i] = i;
}

// Length of the common prefix of keys i and j; equal codes are told apart by position.
__device__ __forceinline__ int prefix(const uint32_t* __restrict__ codes, int n, int i, int j) {
  if (j < 0 || j >= n) return -1;
  uint32_t a = codes[i], b = codes[j];
  if (a == b) return 32 + __clz((uint32_t)i ^ (uint32_t)j);
  return __clz(a ^ b);
}

// One thread per internal node i of the radix tree over n sorted keys.
// Node ids: internal i -> i, leaf j -> (n-1)+j.
__global__ void k_hierarchy(const uint32_t* __restrict__ codes, int n, uint32_t* __restrict__ left,
                            uint32_t* __restrict__ right, uint32_t* __restrict__ lo, uint32_t* __restrict__ hi) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n - 1) return;
  int d = (prefix(codes, n, i, i + 1) - prefix(codes, n, i, i - 1)) > 0 ? 1 : -1;
  int dmin = prefix(codes, n, i, i - d);
  int lmax = 2;
  while (prefix(codes, n, i, i + lmax * d) > dmin) lmax *= 2;
  int l = 0;
  for (int t = lmax / 2; t >= 1; t /= 2)
    if (prefix(codes, n, i, i + (l + t) * d) > dmin) l += t;
  int j = i + l * d;
  int dnode = prefix(codes, n, i, j);
  int s = 0;
  for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
    if (prefix(codes, n, i, i + (s + t) * d) > dnode) s += t;
    if (t <= 1) break;
  }
  int gamma = i + s * d + min(d, 0);
  int first = min(i, j), last = max(i, j);
  left[i] = (first == gamma) ? (uint32_t)(n - 1 + gamma) : (uint32_t)gamma;
  right[i] = (last == gamma + 1) ? (uint32_t)(n - 1 + gamma + 1) : (uint32_t)(gamma + 1);
  lo[i] = (uint32_t)first;
  hi[i] = (uint32_t)last;
}

__global__ void k_leaf_boxes(const float* __restrict__ boxMin, const float* __restrict__ boxMax,
                             const uint32_t* __restrict__ sortedIds, uint32_t n, float* __restrict__ nodeMin,
                             float* __restrict__ nodeMax, uint32_t* __restrict__ ready) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  uint32_t t = sortedIds[j];
  size_t o = 3 * (size_t)(n - 1 + j);
  for (int k = 0; k < 3; ++k) {
    nodeMin[o + k] = boxMin[3 * (size_t)t + k];
    nodeMax[o + k] = boxMax[3 * (size_t)t + k];
  }
  ready[n - 1 + j] = 1u;
}

// One bottom-up sweep: an internal node whose children were complete BEFORE this launch
// gets its box.  Launch boundaries order the writes of one sweep before the reads of the
// next, so no in-kernel hand-off between workgroups is needed.
__global__ void k_fit_sweep(const uint32_t* __restrict__ left, const uint32_t* __restrict__ right, uint32_t nInternal,
                            const uint32_t* __restrict__ readyIn, uint32_t* __restrict__ readyOut,
                            float* __restrict__ nodeMin, float* __restrict__ nodeMax, uint32_t* remaining) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  bool pending = false;
  if (i < nInternal && !readyIn[i]) {
    uint32_t l = left[i], r = right[i];
    if (readyIn[l] && readyIn[r]) {
      for (int k = 0; k < 3; ++k) {
        nodeMin[3 * (size_t)i + k] = fminf(nodeMin[3 * (size_t)l + k], nodeMin[3 * (size_t)r + k]);
        nodeMax[3 * (size_t)i + k] = fmaxf(nodeMax[3 * (size_t)l + k], nodeMax[3 * (size_t)r + k]);
      }
      readyOut[i] = 1u;
    } else {
      pending = true;
    }
  }
  unsigned long long b = __ballot(pending);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(remaining, (uint32_t)__popcll(b));
}

__global__ void k_copy_flags(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

void put_box(lpt_bvh_node& g, const float* mn, const float* mx) {
  for (int k = 0; k < 3; ++k) {
    g.minCorner[k] = mn[k];
    g.maxCorner[k] = mx[k];
  }
}

}  // namespace

extern "C" int lpt_build_mesh_bvh(lpt_ctx* ctx, int builder, const float* positions, uint32_t n,
                                  lpt_bvh_node* nodesOut, uint32_t* nNodesOut, uint32_t* primOrderOut) {
  if (!ctx) return LPT_ERR_INVALID;
  if (builder != LPT_BVH_LBVH) return lpt_internal_fail(ctx, LPT_ERR_INVALID, "unknown BVH builder");
  if (!nodesOut || !nNodesOut || (n && (!positions || !primOrderOut)))
    return lpt_internal_fail(ctx, LPT_ERR_INVALID, "null argument");
  if (n >= (1u << 30)) return lpt_internal_fail(ctx, LPT_ERR_INVALID, "too many triangles");

  std::memset(nodesOut, 0, sizeof(lpt_bvh_node));
  if (n == 0) {  // a single empty leaf
    nodesOut[0].isLeaf = 1u;
    *nNodesOut = 1;
    return LPT_OK;
  }

  hipStream_t stream = lpt_internal_stream(ctx);
  float *dPos = nullptr, *dBoxMin = nullptr, *dBoxMax = nullptr, *dNodeMin = nullptr, *dNodeMax = nullptr;
  uint32_t *dBounds = nullptr, *dCodes = nullptr, *dIds = nullptr, *dCodesS = nullptr, *dIdsS = nullptr;
  uint32_t *dLeft = nullptr, *dRight = nullptr, *dLo = nullptr, *dHi = nullptr, *dReadyA = nullptr, *dReadyB = nullptr;
  uint32_t* dRemaining = nullptr;
  void* dTemp = nullptr;
  auto release = [&]() {
    void* ptrs[] = {dPos, dBoxMin, dBoxMax, dNodeMin, dNodeMax, dBounds, dCodes, dIds, dCodesS, dIdsS,
                    dLeft, dRight, dLo, dHi, dReadyA, dReadyB, dRemaining, dTemp};
    for (void* p : ptrs)
      if (p) (void)hipFree(p);
  };

  const uint32_t nNodes = 2 * n - 1, nInternal = n - 1;
  const int B = 256;
  LB_HIP(ctx, hipSetDevice(lpt_internal_device(ctx)));
  LB_HIP(ctx, hipMalloc(&dPos, 9 * (size_t)n * sizeof(float)));
  LB_HIP(ctx, hipMalloc(&dBoxMin, 3 * (size_t)n * sizeof(float)));
  LB_HIP(ctx, hipMalloc(&dBoxMax, 3 * (size_t)n * sizeof(float)));
  LB_HIP(ctx, hipMalloc(&dNodeMin, 3 * (size_t)nNodes * sizeof(float)));
  LB_HIP(ctx, hipMalloc(&dNodeMax, 3 * (size_t)nNodes * sizeof(float)));
  LB_HIP(ctx, hipMalloc(&dBounds, 6 * sizeof(uint32_t)));
  LB_HIP(ctx, hipMalloc(&dCodes, n * sizeof(uint32_t)));
  LB_HIP(ctx, hipMalloc(&dIds, n * sizeof(uint32_t)));
  LB_HIP(ctx, hipMalloc(&dCodesS, n * sizeof(uint32_t)));
  LB_HIP(ctx, hipMalloc(&dIdsS, n * sizeof(uint32_t)));
  LB_HIP(ctx, hipMalloc(&dLeft, (size_t)std::max(1u, nInternal) * sizeof(uint32_t)));
  LB_HIP(ctx, hipMalloc(&dRight, (size_t)std::max(1u, nInternal) * sizeof(uint32_t)));
  LB_HIP(ctx, hipMalloc(&dLo, (size_t)std::max(1u, nInternal) * sizeof(uint32_t)));
  LB_HIP(ctx, hipMalloc(&dHi, (size_t)std::max(1u, nInternal) * sizeof(uint32_t)));
  LB_HIP(ctx, hipMalloc(&dReadyA, (size_t)nNodes * sizeof(uint32_t)));
  LB_HIP(ctx, hipMalloc(&dReadyB, (size_t)nNodes * sizeof(uint32_t)));
  LB_HIP(ctx, hipMalloc(&dRemaining, sizeof(uint32_t)));

  LB_HIP(ctx, hipMemcpyAsync(dPos, positions, 9 * (size_t)n * sizeof(float), hipMemcpyHostToDevice, stream));
  const uint32_t initBounds[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
  LB_HIP(ctx, hipMemcpyAsync(dBounds, initBounds, sizeof initBounds, hipMemcpyHostToDevice, stream));
  LB_HIP(ctx, hipStreamSynchronize(stream));  // initBounds lives on this stack frame
  hipLaunchKernelGGL(k_tri_boxes, dim3((n + B - 1) / B), dim3(B), 0, stream, dPos, n, dBoxMin, dBoxMax, dBounds);
  hipLaunchKernelGGL(k_morton, dim3((n + B - 1) / B), dim3(B), 0, stream, dBoxMin, dBoxMax, n, dBounds, dCodes, dIds);
  LB_HIP(ctx, hipGetLastError());

  size_t tempBytes = 0;
  LB_HIP(ctx, rocprim::radix_sort_pairs(nullptr, tempBytes, dCodes, dCodesS, dIds, dIdsS, n, 0, 30, stream));
  LB_HIP(ctx, hipMalloc(&dTemp, std::max<size_t>(tempBytes, 16)));
  LB_HIP(ctx, rocprim::radix_sort_pairs(dTemp, tempBytes, dCodes, dCodesS, dIds, dIdsS, n, 0, 30, stream));

  LB_HIP(ctx, hipMemsetAsync(dReadyA, 0, (size_t)nNodes * sizeof(uint32_t), stream));
  if (nInternal)
    hipLaunchKernelGGL(k_hierarchy, dim3((nInternal + B - 1) / B), dim3(B), 0, stream, dCodesS, (int)n, dLeft, dRight, dLo, dHi);
  hipLaunchKernelGGL(k_leaf_boxes, dim3((n + B - 1) / B), dim3(B), 0, stream, dBoxMin, dBoxMax, dIdsS, n, dNodeMin, dNodeMax, dReadyA);
  LB_HIP(ctx, hipGetLastError());

  // bottom-up sweeps until every internal node has its box (at most tree-depth sweeps)
  uint32_t remaining = nInternal;
  uint32_t* in = dReadyA;
  uint32_t* out = dReadyB;
  for (int sweep = 0; remaining != 0; ++sweep) {
    if (sweep > 4096) {
      release();
      return lpt_internal_fail(ctx, LPT_ERR_INVALID, "LBVH box fitting did not converge");
    }
    hipLaunchKernelGGL(k_copy_flags, dim3((nNodes + B - 1) / B), dim3(B), 0, stream, in, out, nNodes);
    LB_HIP(ctx, hipMemsetAsync(dRemaining, 0, sizeof(uint32_t), stream));
    hipLaunchKernelGGL(k_fit_sweep, dim3((nInternal + B - 1) / B), dim3(B), 0, stream, dLeft, dRight, nInternal, in, out,
                       dNodeMin, dNodeMax, dRemaining);
    LB_HIP(ctx, hipGetLastError());
    LB_HIP(ctx, hipMemcpyAsync(&remaining, dRemaining, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    LB_HIP(ctx, hipStreamSynchronize(stream));
    std::swap(in, out);
  }

  std::vector<uint32_t> left(nInternal), right(nInternal), lo(nInternal), hi(nInternal);
  std::vector<float> nodeMin(3 * (size_t)nNodes), nodeMax(3 * (size_t)nNodes);
  if (nInternal) {
    LB_HIP(ctx, hipMemcpy(left.data(), dLeft, nInternal * sizeof(uint32_t), hipMemcpyDeviceToHost));
    LB_HIP(ctx, hipMemcpy(right.data(), dRight, nInternal * sizeof(uint32_t), hipMemcpyDeviceToHost));
    LB_HIP(ctx, hipMemcpy(lo.data(), dLo, nInternal * sizeof(uint32_t), hipMemcpyDeviceToHost));
    LB_HIP(ctx, hipMemcpy(hi.data(), dHi, nInternal * sizeof(uint32_t), hipMemcpyDeviceToHost));
  }
  LB_HIP(ctx, hipMemcpy(nodeMin.data(), dNodeMin, nodeMin.size() * sizeof(float), hipMemcpyDeviceToHost));
  LB_HIP(ctx, hipMemcpy(nodeMax.data(), dNodeMax, nodeMax.size() * sizeof(float), hipMemcpyDeviceToHost));
  LB_HIP(ctx, hipMemcpy(primOrderOut, dIdsS, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
  release();

  // depth-first write-out; a subtree spanning at most kLeafSize sorted triangles becomes a leaf
  uint32_t count = 0;
  struct Item { uint32_t node; uint32_t parentSlot; int side; };
  std::vector<Item> st;
  st.push_back({nInternal ? 0u : (n - 1), 0xFFFFFFFFu, 0});
  while (!st.empty()) {
    Item it = st.back();
    st.pop_back();
    uint32_t slot = count++;
    lpt_bvh_node& g = nodesOut[slot];
    std::memset(&g, 0, sizeof g);
    put_box(g, &nodeMin[3 * (size_t)it.node], &nodeMax[3 * (size_t)it.node]);
    if (it.parentSlot != 0xFFFFFFFFu) nodesOut[it.parentSlot].indices[it.side] = slot;
    if (it.node >= nInternal) {  // single triangle
      uint32_t j = it.node - nInternal;
      g.isLeaf = 1u;
      g.indices[0] = j;
      g.indices[1] = j + 1;
    } else if (hi[it.node] - lo[it.node] + 1 <= kLeafSize) {
      g.isLeaf = 1u;
      g.indices[0] = lo[it.node];
      g.indices[1] = hi[it.node] + 1;
    } else {
      st.push_back({right[it.node], slot, 1});
      st.push_back({left[it.node], slot, 0});  // left is written next: first child follows its parent
    }
  }
  *nNodesOut = count;
  return LPT_OK;
}
