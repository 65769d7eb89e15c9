// BVH construction on the GPU for one triangle mesh.  Both builders start alike:
//   triangle boxes + mesh bounds -> 30-bit Morton codes of the box centres -> radix sort (rocPRIM).
// LPT_BVH_LBVH: radix-tree hierarchy over the sorted codes (Karras, "Maximizing Parallelism in the
//   Construction of BVHs, Octrees, and k-d Trees", 2012) -> bottom-up box fitting.
// LPT_BVH_PLOC: bottom-up agglomeration of the sorted triangles (Meister & Bittner, "Parallel Locally-
//   Ordered Clustering for Bounding Volume Hierarchy Construction", 2018): every cluster looks kPlocRadius
//   places to either side along the Morton curve for the neighbour with which it forms the smallest box,
//   mutual choices merge, the cluster list is compacted, until one cluster is left.  The tree follows the
//   geometry (surface area of the merged boxes) instead of the bits of the codes: SAH-class quality.
// The tree is then written out on the host in depth-first order as interchange nodes (include/lpt.h:
// lpt_bvh_node); subtrees of at most kLeafSize triangles become leaves where the surface-area heuristic finds that
// cheaper than descending into them.  Stands in for
// lsg::bvh::SplitBVHBuilder at src/PTSceneConverter.cpp:97-101 of the reference.
#include <hip/hip_runtime.h>
#include <string.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "lpt_internal.h"

namespace {

#ifndef LPT_GPU_BVH_LEAF
#define LPT_GPU_BVH_LEAF 4
#endif
constexpr uint32_t kLeafSize = LPT_GPU_BVH_LEAF;
constexpr float kTraversalCost = 1.5f;  // an inner-node visit in units of one triangle test

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

// ---- PLOC ------------------------------------------------------------------------------------

#ifndef LPT_PLOC_RADIUS
#define LPT_PLOC_RADIUS 16
#endif
constexpr int kPlocRadius = LPT_PLOC_RADIUS;

__device__ __forceinline__ float union_area(const float* __restrict__ nodeMin, const float* __restrict__ nodeMax,
                                            uint32_t a, uint32_t b) {
  float d[3];
  for (int k = 0; k < 3; ++k)
    d[k] = fmaxf(nodeMax[3 * (size_t)a + k], nodeMax[3 * (size_t)b + k]) -
           fminf(nodeMin[3 * (size_t)a + k], nodeMin[3 * (size_t)b + k]);
  return d[0] * d[1] + d[1] * d[2] + d[2] * d[0];
}

// nn[i] = the position j in [i-r, i+r], j != i, whose cluster forms the smallest box with cluster i.  Pairs
// are ranked by a key that both members of a pair compute alike — (area, distance |i - j|, parity of the lower
// position, lower position) — a total order, so the best pair of the whole list always chooses each other and
// every round merges something.  The order among equal areas matters when many areas are equal (coincident
// triangles, triangles on one axis line, extents whose products underflow to 0 or overflow to infinity): nearest
// first, and among the two nearest the pair that starts at an even position, makes the pairs (2k, 2k+1) choose
// each other, so such a list halves every round.  (Ranked by position alone, every cluster would choose the lowest
// position of its window, only the pair (0, 1) would be mutual, and the build would take one round per triangle.)
__global__ void k_ploc_nearest(const uint32_t* __restrict__ cid, uint32_t m, const float* __restrict__ nodeMin,
                               const float* __restrict__ nodeMax, uint32_t* __restrict__ nn) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const uint32_t me = cid[i];
  const uint32_t lo = i > (uint32_t)kPlocRadius ? i - kPlocRadius : 0u;
  const uint32_t hi = min(m - 1u, i + (uint32_t)kPlocRadius);
  float best = INFINITY;
  // some neighbour even if every area is NaN: the other member of the pair (2k, 2k+1) where there is one
  uint32_t bestJ = (i ^ 1u) < m ? (i ^ 1u) : i - 1u;
  uint32_t bestLo = min(i, bestJ), bestDist = 1u;
  bool have = false;
  for (uint32_t j = lo; j <= hi; ++j) {
    if (j == i) continue;
    const float a = union_area(nodeMin, nodeMax, me, cid[j]);
    const uint32_t pl = min(i, j), dist = max(i, j) - pl;
    bool better;
    if (!have) {
      better = a == a;
    } else if (a != best) {
      better = a < best;
    } else if (dist != bestDist) {
      better = dist < bestDist;
    } else if ((pl & 1u) != (bestLo & 1u)) {
      better = (pl & 1u) == 0u;
    } else {
      better = pl < bestLo;
    }
    if (better) {
      best = a;
      bestJ = j;
      bestLo = pl;
      bestDist = dist;
      have = true;
    }
  }
  nn[i] = bestJ;
}

// leader[i] = 1 where i and nn[i] chose each other and i is the lower position (it will hold the merged
// cluster); valid[i] = 0 for the higher position of such a pair (it disappears), 1 otherwise.
// `pairUp`: a round without any mutual choice (possible only with non-finite boxes) pairs neighbours instead.
__global__ void k_ploc_flags(uint32_t* __restrict__ nn, uint32_t m, uint32_t* __restrict__ leader,
                             uint32_t* __restrict__ valid, int pairUp) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  uint32_t j = nn[i];
  bool mutual = nn[j] == i;
  if (pairUp) {
    j = (i & 1u) ? i - 1u : i + 1u;
    mutual = j < m;
  }
  leader[i] = (mutual && i < j) ? 1u : 0u;
  valid[i] = (mutual && i > j) ? 0u : 1u;
}
__global__ void k_ploc_pair_up(uint32_t* __restrict__ nn, uint32_t m) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const uint32_t j = (i & 1u) ? i - 1u : i + 1u;
  nn[i] = j < m ? j : i;  // the odd one out keeps itself: never mutual with anyone
}

// Leaders create their inner node (ids count up from nextId in list order: the numbering does not depend on
// scheduling) and everyone still valid moves to its place in the compacted list.
__global__ void k_ploc_merge(const uint32_t* __restrict__ cid, const uint32_t* __restrict__ nn, uint32_t m,
                             const uint32_t* __restrict__ leader, const uint32_t* __restrict__ leaderRank,
                             const uint32_t* __restrict__ valid, const uint32_t* __restrict__ validRank, uint32_t nextId,
                             uint32_t* __restrict__ left, uint32_t* __restrict__ right, float* __restrict__ nodeMin,
                             float* __restrict__ nodeMax, uint32_t* __restrict__ out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m || !valid[i]) return;
  uint32_t id = cid[i];
  if (leader[i]) {
    const uint32_t a = cid[i], b = cid[nn[i]];
    id = nextId + leaderRank[i];
    left[id] = a;
    right[id] = b;
    for (int k = 0; k < 3; ++k) {
      nodeMin[3 * (size_t)id + k] = fminf(nodeMin[3 * (size_t)a + k], nodeMin[3 * (size_t)b + k]);
      nodeMax[3 * (size_t)id + k] = fmaxf(nodeMax[3 * (size_t)a + k], nodeMax[3 * (size_t)b + k]);
    }
  }
  out[validRank[i]] = id;
}

// The four numbers the host needs after a round's scans, in one place: how many pairs merge and how many
// clusters survive are the last rank plus the last flag of each scan.
__global__ void k_ploc_tail(const uint32_t* __restrict__ leaderRank, const uint32_t* __restrict__ leader,
                            const uint32_t* __restrict__ validRank, const uint32_t* __restrict__ valid, uint32_t m,
                            uint32_t* __restrict__ out /* [2]: merges, survivors */) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    out[0] = leaderRank[m - 1u] + leader[m - 1u];
    out[1] = validRank[m - 1u] + valid[m - 1u];
  }
}

__global__ void k_ploc_init(uint32_t* __restrict__ cid, uint32_t n) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) cid[j] = n - 1u + j;  // leaf ids, in Morton order
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
  if (builder != LPT_BVH_LBVH && builder != LPT_BVH_PLOC) return lpt_internal_fail(ctx, LPT_ERR_INVALID, "unknown BVH builder");
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
  uint32_t* dRemaining2 = nullptr;  // the two words a clustering round reports
  void* dTemp = nullptr;
  uint32_t *dCidA = nullptr, *dCidB = nullptr, *dNn = nullptr, *dLeader = nullptr, *dValid = nullptr, *dLeaderRank = nullptr,
           *dValidRank = nullptr;
  void* dScanTemp = nullptr;
  auto release = [&]() {
    void* ptrs[] = {dPos, dBoxMin, dBoxMax, dNodeMin, dNodeMax, dBounds, dCodes, dIds, dCodesS, dIdsS,
                    dLeft, dRight, dLo, dHi, dReadyA, dReadyB, dRemaining, dRemaining2, dTemp,
                    dCidA, dCidB, dNn, dLeader, dValid, dLeaderRank, dValidRank, dScanTemp};
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
  if (nInternal && builder == LPT_BVH_LBVH)
    hipLaunchKernelGGL(k_hierarchy, dim3((nInternal + B - 1) / B), dim3(B), 0, stream, dCodesS, (int)n, dLeft, dRight, dLo, dHi);
  hipLaunchKernelGGL(k_leaf_boxes, dim3((n + B - 1) / B), dim3(B), 0, stream, dBoxMin, dBoxMax, dIdsS, n, dNodeMin, dNodeMax, dReadyA);
  LB_HIP(ctx, hipGetLastError());

  uint32_t rootId = nInternal ? 0u : n - 1u;  // LBVH: internal node 0; a single triangle: its leaf
  if (builder == LPT_BVH_PLOC && nInternal) {
    // agglomeration rounds; internal ids are handed out in creation order, the last one created is the root
    LB_HIP(ctx, hipMalloc(&dCidA, n * sizeof(uint32_t)));
    LB_HIP(ctx, hipMalloc(&dCidB, n * sizeof(uint32_t)));
    LB_HIP(ctx, hipMalloc(&dNn, n * sizeof(uint32_t)));
    LB_HIP(ctx, hipMalloc(&dLeader, n * sizeof(uint32_t)));
    LB_HIP(ctx, hipMalloc(&dValid, n * sizeof(uint32_t)));
    LB_HIP(ctx, hipMalloc(&dLeaderRank, n * sizeof(uint32_t)));
    LB_HIP(ctx, hipMalloc(&dValidRank, n * sizeof(uint32_t)));
    LB_HIP(ctx, hipMalloc(&dRemaining2, 2 * sizeof(uint32_t)));
    size_t scanBytes = 0;
    LB_HIP(ctx, rocprim::exclusive_scan(nullptr, scanBytes, dLeader, dLeaderRank, 0u, n, rocprim::plus<uint32_t>(), stream));
    LB_HIP(ctx, hipMalloc(&dScanTemp, std::max<size_t>(scanBytes, 16)));
    hipLaunchKernelGGL(k_ploc_init, dim3((n + B - 1) / B), dim3(B), 0, stream, dCidA, n);
    uint32_t m = n, nextId = 0;
    uint32_t* cur = dCidA;
    uint32_t* nxt = dCidB;
    for (int round = 0; m > 1; ++round) {
      if (round > 4096) {  // every round removes at least one cluster in 4 * radius: about 64 ln(n) rounds at worst
        release();
        return lpt_internal_fail(ctx, LPT_ERR_INVALID, "PLOC clustering did not converge");
      }
      const dim3 grid((m + B - 1) / B);
      hipLaunchKernelGGL(k_ploc_nearest, grid, dim3(B), 0, stream, cur, m, dNodeMin, dNodeMax, dNn);
      // A round that would merge fewer than one cluster in 4 * radius (possible only when nearly all candidate
      // boxes are non-finite, so that hardly any two clusters choose each other) pairs neighbours instead: the
      // number of rounds stays logarithmic whatever the input.
      uint32_t tail[2] = {0, 0};  // merges, survivors
      for (int pairUp = 0; pairUp < 2; ++pairUp) {
        if (pairUp) hipLaunchKernelGGL(k_ploc_pair_up, grid, dim3(B), 0, stream, dNn, m);
        hipLaunchKernelGGL(k_ploc_flags, grid, dim3(B), 0, stream, dNn, m, dLeader, dValid, 0);
        LB_HIP(ctx, hipGetLastError());
        LB_HIP(ctx, rocprim::exclusive_scan(dScanTemp, scanBytes, dLeader, dLeaderRank, 0u, m, rocprim::plus<uint32_t>(), stream));
        LB_HIP(ctx, rocprim::exclusive_scan(dScanTemp, scanBytes, dValid, dValidRank, 0u, m, rocprim::plus<uint32_t>(), stream));
        hipLaunchKernelGGL(k_ploc_tail, dim3(1), dim3(64), 0, stream, dLeaderRank, dLeader, dValidRank, dValid, m, dRemaining2);
        LB_HIP(ctx, hipGetLastError());
        LB_HIP(ctx, hipMemcpyAsync(tail, dRemaining2, sizeof tail, hipMemcpyDeviceToHost, stream));
        LB_HIP(ctx, hipStreamSynchronize(stream));
        if ((uint64_t)tail[0] * 4u * (uint32_t)kPlocRadius >= m) break;
      }
      const uint32_t merges = tail[0], survivors = tail[1];
      if (merges == 0 || survivors != m - merges || nextId + merges > nInternal) {
        release();
        return lpt_internal_fail(ctx, LPT_ERR_INVALID, "PLOC clustering made no progress");
      }
      hipLaunchKernelGGL(k_ploc_merge, grid, dim3(B), 0, stream, cur, dNn, m, dLeader, dLeaderRank, dValid, dValidRank, nextId,
                         dLeft, dRight, dNodeMin, dNodeMax, nxt);
      LB_HIP(ctx, hipGetLastError());
      nextId += merges;
      m = survivors;
      std::swap(cur, nxt);
    }
    LB_HIP(ctx, hipStreamSynchronize(stream));
    rootId = nextId - 1u;  // == nInternal - 1
  }

  // LBVH: bottom-up sweeps until every internal node has its box (at most tree-depth sweeps)
  uint32_t remaining = builder == LPT_BVH_LBVH ? nInternal : 0u;
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

  std::vector<uint32_t> left(nInternal), right(nInternal), sorted(n);
  std::vector<float> nodeMin(3 * (size_t)nNodes), nodeMax(3 * (size_t)nNodes);
  if (nInternal) {
    LB_HIP(ctx, hipMemcpy(left.data(), dLeft, nInternal * sizeof(uint32_t), hipMemcpyDeviceToHost));
    LB_HIP(ctx, hipMemcpy(right.data(), dRight, nInternal * sizeof(uint32_t), hipMemcpyDeviceToHost));
  }
  LB_HIP(ctx, hipMemcpy(nodeMin.data(), dNodeMin, nodeMin.size() * sizeof(float), hipMemcpyDeviceToHost));
  LB_HIP(ctx, hipMemcpy(nodeMax.data(), dNodeMax, nodeMax.size() * sizeof(float), hipMemcpyDeviceToHost));
  LB_HIP(ctx, hipMemcpy(sorted.data(), dIdsS, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
  release();

  // Triangles under every internal node (children before parents: iterative post-order); also rejects
  // anything that is not a tree over all n leaves.
  std::vector<uint32_t> tris(nInternal, 0u);
  std::vector<uint8_t> asLeaf;
  {
    std::vector<uint32_t> order;
    order.reserve(nInternal);
    std::vector<uint32_t> walk;
    if (rootId < nInternal) walk.push_back(rootId);
    while (!walk.empty()) {
      const uint32_t i = walk.back();
      walk.pop_back();
      order.push_back(i);
      if (order.size() > nInternal) return lpt_internal_fail(ctx, LPT_ERR_INVALID, "GPU BVH build produced a malformed tree");
      for (uint32_t ch : {left[i], right[i]}) {
        if (ch >= nNodes) return lpt_internal_fail(ctx, LPT_ERR_INVALID, "GPU BVH build produced a malformed tree");
        if (ch < nInternal) walk.push_back(ch);
      }
    }
    for (size_t k = order.size(); k-- > 0;) {
      const uint32_t i = order[k];
      tris[i] = (left[i] < nInternal ? tris[left[i]] : 1u) + (right[i] < nInternal ? tris[right[i]] : 1u);
    }
    // Which subtrees become leaves: of those with at most kLeafSize triangles, the ones that are cheaper to test
    // whole than to descend into (surface-area heuristic, costs in units of one triangle test, an inner-node visit
    // at kTraversalCost: the rule of the host builder).  cost[] = cheapest way to handle the subtree.
    auto area = [&](uint32_t node) {
      const float* mn = &nodeMin[3 * (size_t)node];
      const float* mx = &nodeMax[3 * (size_t)node];
      const float d[3] = {mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2]};
      if (!(d[0] >= 0.0f && d[1] >= 0.0f && d[2] >= 0.0f)) return 0.0f;
      return d[0] * d[1] + d[1] * d[2] + d[2] * d[0];
    };
    std::vector<float> cost(nInternal, 0.0f);
    asLeaf.assign(nInternal, 0);
    for (size_t k = order.size(); k-- > 0;) {
      const uint32_t i = order[k];
      const uint32_t l = left[i], r = right[i];
      const float a = area(i);
      const float cl = l < nInternal ? cost[l] : 1.0f, cr = r < nInternal ? cost[r] : 1.0f;
      // a box without extent (or a non-finite one) gives no ratio: keep its triangles together
      const float split = a > 0.0f && a < INFINITY ? kTraversalCost + (area(l) * cl + area(r) * cr) / a : INFINITY;
      const float whole = (float)tris[i];
      if (tris[i] <= kLeafSize && !(split < whole)) {
        asLeaf[i] = 1;
        cost[i] = whole;
      } else {
        cost[i] = split < INFINITY ? split : whole;
      }
    }
    if ((rootId < nInternal ? tris[rootId] : 1u) != n)
      return lpt_internal_fail(ctx, LPT_ERR_INVALID, "GPU BVH build produced a malformed tree");
  }

  // depth-first write-out; a subtree of at most kLeafSize triangles becomes a leaf whose triangles take the
  // next places of the triangle order (for the radix tree that is the Morton order itself)
  uint32_t count = 0, placed = 0;
  struct Item { uint32_t node; uint32_t parentSlot; int side; };
  std::vector<Item> st;
  std::vector<uint32_t> gather;
  st.push_back({rootId, 0xFFFFFFFFu, 0});
  while (!st.empty()) {
    Item it = st.back();
    st.pop_back();
    if (count >= nNodes) return lpt_internal_fail(ctx, LPT_ERR_INVALID, "GPU BVH build produced a malformed tree");
    uint32_t slot = count++;
    lpt_bvh_node& g = nodesOut[slot];
    std::memset(&g, 0, sizeof g);
    put_box(g, &nodeMin[3 * (size_t)it.node], &nodeMax[3 * (size_t)it.node]);
    if (it.parentSlot != 0xFFFFFFFFu) nodesOut[it.parentSlot].indices[it.side] = slot;
    if (it.node >= nInternal || asLeaf[it.node]) {
      g.isLeaf = 1u;
      g.indices[0] = placed;
      gather.assign(1, it.node);
      while (!gather.empty()) {  // the subtree's triangles, left to right
        const uint32_t v = gather.back();
        gather.pop_back();
        if (v >= nInternal) {
          if (placed >= n) return lpt_internal_fail(ctx, LPT_ERR_INVALID, "GPU BVH build produced a malformed tree");
          primOrderOut[placed++] = sorted[v - nInternal];
        } else {
          gather.push_back(right[v]);
          gather.push_back(left[v]);
        }
      }
      g.indices[1] = placed;
    } else {
      st.push_back({right[it.node], slot, 1});
      st.push_back({left[it.node], slot, 0});  // left is written next: first child follows its parent
    }
  }
  *nNodesOut = count;
  return LPT_OK;
}
