// Persistent-thread closest-hit kernel (k_extend).
//
// Every lane walks its ray through exactly the sequence of node visits, box tests and
// triangle tests that intersect_scene (lpt_device.h) performs, so hits are identical.
// What differs is the scheduling across the 64 lanes of a wave: a lane's traversal is
// cut into unit steps of four kinds — FETCH a new ray, one INNER node (two box tests),
// one TRIangle leaf, enter one OBJect — and in each round the wave runs the kind that
// most lanes are waiting for.  Lanes of other kinds simply wait; waiting never changes a
// lane's own sequence.  This keeps the SIMD lanes of each issued instruction well filled
// where a per-lane "if leaf / else inner" loop would run every branch with few lanes.
//
// A round of the INNER kind goes on in a short burst loop for the lanes whose next reference
// is again an inner node, while enough of them remain; a round of the TRI kind finishes the
// leaf.  The burst loops carry only the reference, the stack pointer and the hit, so they are
// much shorter than a full round of the scheduler.
//
// Rays are handed out in chunks: one atomic on a global cursor per chunk and wave, the
// chunk is then consumed lane by lane as rays finish (no idle tail inside a wave).
//
// The scene is read from global memory.  Each step gathers one 64-byte node, 48-byte triangle or
// 64-byte object record per lane at unrelated addresses; the L1 serves such loads lane by lane and
// that, together with vector issue, is what bounds the kernel (not HBM: the hot part of the scene
// sits in L1/L2).  A few of the hottest nodes are kept in LDS (a second, parallel path for 16-byte
// gathers); larger LDS copies of nodes, triangles and object records cost resident workgroups and
// were slower: random 16-byte reads cost the LDS what they cost the L1.
#pragma once
#include "lpt_device.h"

#pragma clang fp contract(off)

namespace lpt {

#ifdef LPT_EXTEND_STATS
// analysis aid: [2k] rounds (or burst iterations) of kind k, [2k+1] lanes that took part (k < 6);
// [12] longest time a wave went on after the queue ran dry, [14] sum of those times, [15] sum of all wave
// times (100 MHz ticks); [16 + k] rays that took 32k .. 32k+31 steps (the counting slows the kernel down)
__device__ unsigned long long g_extendStats[16 + 64];
// rays by the most stack entries they held at one time
__device__ unsigned long long g_extendDepth[64];
#define LPT_XSTAT(k, lanes) (++xsR[k], xsL[k] += (uint32_t)(lanes))
#define LPT_XDEPTH() (xsDepth = max(xsDepth, sp))
#else
#define LPT_XSTAT(k, lanes) ((void)0)
#define LPT_XDEPTH() ((void)0)
#endif

// Rays are handed out in chunks of kFetchChunk.  The queue consists of kCursorRegions regions (its
// part queues, see QueueFill) with one cursor each (a single atomic word would serialise all waves); a
// wave starts in the region of its workgroup and moves on to the next one when it runs dry.
#ifndef LPT_FETCH_CHUNK
#define LPT_FETCH_CHUNK 128
#endif
constexpr uint32_t kFetchChunk = LPT_FETCH_CHUNK;
#ifndef LPT_CURSOR_REGIONS
#define LPT_CURSOR_REGIONS 8
#endif
constexpr uint32_t kCursorRegions = LPT_CURSOR_REGIONS;
// Every cursor has a cache line of its own: atomics on words of one line are carried out one after the other
// (about 100 per microsecond), and the 7 168 waves of a launch issue several hundred thousand of them.
#ifndef LPT_CURSOR_STRIDE
#define LPT_CURSOR_STRIDE 64
#endif
constexpr uint32_t kCursorStride = LPT_CURSOR_STRIDE;               // words between two cursors
constexpr uint32_t kCursorWords = kCursorRegions * kCursorStride;  // one set of cursors

// A path queue is kept as kCursorRegions part queues of `capPart` entries each: part r holds its entries at
// [r * capPart, r * capPart + fill[r]), and fill[r] is word r * kCursorStride of the queue's set of fill counts
// (laid out like a set of cursors, for the same reason: the appending kernels add to them with atomics, one per
// workgroup, several hundred thousand per launch).  A workgroup appends to the part of its own number modulo
// kCursorRegions, which is also the region its waves take rays from first: consecutive workgroups run on
// different XCDs, so a part is mostly written and read through one XCD's L2.
struct QueueFill {
  const uint32_t* fill;  // set of fill counts
  uint32_t capPart;      // entries per part queue
};
__device__ __forceinline__ uint32_t part_fill(const QueueFill& q, uint32_t part) { return q.fill[part * kCursorStride]; }

#ifndef LPT_FETCH_THRESHOLD
#define LPT_FETCH_THRESHOLD 16
#endif
// An inner-node burst goes on while left * DEN >= entered * NUM lanes remain in it.
#ifndef LPT_BURST_INNER_NUM
#define LPT_BURST_INNER_NUM 1
#define LPT_BURST_INNER_DEN 2
#endif
// Triangle bursts always take the first kBurstTriFree triangles of a leaf (the builders here make
// leaves of at most four), then apply the same kind of rule.
#ifndef LPT_BURST_TRI_NUM
#define LPT_BURST_TRI_NUM 1
#define LPT_BURST_TRI_DEN 4
#endif
constexpr int kBurstTriFree = 4;
constexpr int kStFetch = 0, kStInner = 1, kStTri = 2, kStObj = 3, kStDone = 4;

// The first CACHE derived nodes (the array is ordered hottest first) are kept in LDS with a stride of
// kCacheStride words: 80 bytes, so that 16-byte reads at unrelated nodes spread over all banks.
constexpr int kCacheStride = 20;

// Rays a GUARDed launch could not finish within its S stack entries: their queue indices, for extend_deep.
struct DeepList {
  uint32_t* count;  // entries so far (zeroed before the launch)
  uint32_t* index;  // room for every ray of the queue
};

// `stack` is the workgroup's LDS array of S entries per thread, laid out [entry][thread]; `cache` one of
// CACHE * kCacheStride words (unused if CACHE == 0).
// GUARD: for scenes whose trees may want more than S entries for some ray.  The bound computed at upload is over
// all rays a tree admits; what rays really hold is far less (of 395 M rays of a 400 k-triangle scene whose bound is
// 20, 99.97 % never hold more than 12 entries and 83 more than 15).  A lane whose ray is about to need entry S
// drops the ray — its index goes to `deep`, nothing is written for it — and takes the next one; a small launch
// afterwards (extend_deep) searches the listed rays from the start with their stacks in global memory: the same
// visits in the same order, so the same hits.  This lets every scene run the S = 16 instance with its eight waves
// per SIMD and 48 cached nodes.
template <int S, int BLOCK, int CACHE, bool GUARD = false>
__device__ __forceinline__ void extend_persistent(const DScene& sc, const float4* __restrict__ qo,
                                                  const float4* __restrict__ qd, const QueueFill& fill,
                                                  float4* __restrict__ hits, uint32_t* cursor, uint32_t* stack,
                                                  float* cache, DeepList deep = DeepList{nullptr, nullptr}) {
  uint32_t* stk = &stack[threadIdx.x];
  if (CACHE > 0) {
    const uint32_t words = min((uint32_t)CACHE, sc.nNodes) * 16u;
    const float* src = reinterpret_cast<const float*>(sc.nodes);
    for (uint32_t w = threadIdx.x; w < words; w += BLOCK) cache[(w >> 4) * kCacheStride + (w & 15u)] = src[w];
    __syncthreads();
  }
  const uint32_t lane = threadIdx.x & 63u;

  // wave-uniform chunk of ray indices [chunkNext, chunkEnd)
  uint32_t chunkNext = 0, chunkEnd = 0;
  bool exhausted = false;
  uint32_t region = blockIdx.x % kCursorRegions;  // region (part queue) being drained
  uint32_t regionsTried = 0;

  int state = kStFetch;
  // The ray in world space and in the space of the object being traversed, side by side: each set has one
  // writer (the fetch, the object entry) and `inObj` says which one a step reads, so entering and leaving an
  // object moves no ray data around.
  f3 wo{0, 0, 0}, wd{0, 0, 0}, winv{0, 0, 0};
  f3 oo{0, 0, 0}, od{0, 0, 0}, oinv{0, 0, 0};
  Hit hit{kInfinity, 0u, 0u};
  bool inObj = false;
  uint32_t curObj = 0, triI = 0, triEnd = 0;
  uint32_t cur = kRefSentinel, rayIdx = 0;
  int sp = 0;

#ifdef LPT_EXTEND_STATS
  int xsDepth = 0;
  uint32_t xsR[8] = {}, xsL[8] = {}, xsSteps = 0;
  const unsigned long long xsT0 = wall_clock64();
  unsigned long long xsTx = 0;
#define LPT_XSTEP() (++xsSteps)
#else
#define LPT_XSTEP() ((void)0)
#endif
  for (;;) {
    const unsigned long long bF = __ballot(state == kStFetch);
    const unsigned long long bI = __ballot(state == kStInner);
    const unsigned long long bT = __ballot(state == kStTri);
    const unsigned long long bO = __ballot(state == kStObj);
    const int nF = __popcll(bF), nI = __popcll(bI), nT = __popcll(bT), nO = __popcll(bO);
    const int busy = nI + nT + nO;
    if (busy + nF == 0) break;
#ifdef LPT_EXTEND_STATS
    if (exhausted && !xsTx) xsTx = wall_clock64();
#endif

    bool stepped = false;  // this lane advanced and holds a new `cur` (or needs a pop)
    bool needPop = false;
    bool tooDeep = false;  // GUARD: this lane's ray is about to need more entries than the LDS stack has

    // Idle lanes are pure loss, so refill as soon as a fair number waits (or nothing else is left).
    if (nF > 0 && (nF >= LPT_FETCH_THRESHOLD || busy == 0)) {
      LPT_XSTAT(0, nF);
      while (!exhausted && chunkNext >= chunkEnd) {
        const uint32_t lo = region * fill.capPart, size = part_fill(fill, region);
        uint32_t b = size;
        if (size != 0u) {
          if (lane == 0) b = atomicAdd(&cursor[region * kCursorStride], kFetchChunk);
          b = (uint32_t)__builtin_amdgcn_readfirstlane((int)__shfl((int)b, 0));
        }
        if (b < size) {
          chunkNext = lo + b;
          chunkEnd = lo + min(b + kFetchChunk, size);
        } else {
          region = (region + 1u) % kCursorRegions;
          if (++regionsTried >= kCursorRegions) exhausted = true;
        }
      }
      if (exhausted) {
        if (state == kStFetch) state = kStDone;
      } else {
        const uint32_t avail = chunkEnd - chunkNext;
        if (state == kStFetch) {
          const uint32_t rank =
              __builtin_amdgcn_mbcnt_hi((uint32_t)(bF >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bF, 0u));
          if (rank < avail) {
            rayIdx = chunkNext + rank;
            const float4 o = qo[rayIdx], d = qd[rayIdx];
            wo = f3{o.x, o.y, o.z};
            wd = f3{d.x, d.y, d.z};
            winv = f3{1.0f / wd.x, 1.0f / wd.y, 1.0f / wd.z};
            hit = Hit{kInfinity, 0u, 0u};
            inObj = false;
            stk[0] = kRefSentinel;
            sp = 1;
            cur = sc.rootRef;
            stepped = true;
          }
        }
        chunkNext += min((uint32_t)nF, avail);
      }
    } else if (nI >= nT && nI >= nO) {
      LPT_XSTAT(1, nI);
      if (state == kStInner) {
        const f3 co = inObj ? oo : wo, cinv = inObj ? oinv : winv;
        for (;;) {
          float4 q0, q1, q2, q3;
          if (CACHE > 0 && cur < (uint32_t)CACHE) {
            // volatile LDS reads: kept as ds_read_b128 (left to itself the optimiser merges this branch with the
            // global one into flat loads on a generic pointer; measured 0 % / -2 % kernel time against those)
            typedef float v4f __attribute__((ext_vector_type(4)));
            typedef volatile __attribute__((address_space(3))) v4f lds_v4;
            lds_v4* q = (lds_v4*)(cache + cur * kCacheStride);
            const v4f a0 = q[0], a1 = q[1], a2 = q[2], a3 = q[3];
            q0 = make_float4(a0.x, a0.y, a0.z, a0.w);
            q1 = make_float4(a1.x, a1.y, a1.z, a1.w);
            q2 = make_float4(a2.x, a2.y, a2.z, a2.w);
            q3 = make_float4(a3.x, a3.y, a3.z, a3.w);
          } else {
            const float4* q = record_words(sc.nodes, cur);
            q0 = q[0];
            q1 = q[1];
            q2 = q[2];
            q3 = q[3];
          }
          bool h0, h1;
          box_test_pair(co, cinv, q0, q1, q2, hit.t, h0, h1);
          LPT_XSTEP();
          const uint32_t r0 = as_uint(q3.x), r1 = as_uint(q3.y);
          if (h0 && h1) {
            if (GUARD && sp >= S) {  // no room for the other child
              tooDeep = true;
              break;
            }
            stk[(sp++) * BLOCK] = r0;
            LPT_XDEPTH();
          }
          cur = h1 ? r1 : r0;
          if (!(h0 || h1)) cur = stk[(--sp) * BLOCK];
          // a lane leaves the burst when its next reference is no inner node (the marker, leaves and
          // the sentinel are larger); the others go on while they are enough of those that entered
          if (cur >= kRefMarker) break;
          const int left = __popcll(__ballot(1));
          if (LPT_BURST_INNER_DEN * left < LPT_BURST_INNER_NUM * nI) break;
          LPT_XSTAT(4, left);
        }
        stepped = true;  // `cur` already holds the next reference
      }
    } else if (nT >= nO) {
      LPT_XSTAT(2, nT);
      if (state == kStTri) {
        if (triI < triEnd)
          for (int it = 1;; ++it) {
            const float4* tq = record_words(sc.tris, triI);
            const float4 a = tq[0], b = tq[1], c = tq[2];
            const float td = tri_test(oo, od, f3{a.x, a.y, a.z}, f3{b.x, b.y, b.z}, f3{c.x, c.y, c.z});
            if (td > kEps && td < hit.t) {
              hit.t = td;
              hit.obj = curObj;
              hit.prim = as_uint(a.w);
            }
            LPT_XSTEP();
            if (++triI >= triEnd) break;  // leaf done: this lane leaves the burst
            const int left = __popcll(__ballot(1));
            if (it >= kBurstTriFree && LPT_BURST_TRI_DEN * left < LPT_BURST_TRI_NUM * nT) break;
            LPT_XSTAT(5, left);
          }
        if (triI >= triEnd) {
          needPop = true;
          stepped = true;
        }
      }
    } else {
      LPT_XSTAT(3, nO);
      if (state == kStObj) {
        const uint32_t first = leaf_first(cur);
        const uint32_t count = leaf_count(cur);
        if (count == 0u) {
          needPop = true;
        } else if (GUARD && sp + (count > 1u ? 2 : 1) > S) {
          tooDeep = true;
        } else {
          if (count > 1u) stk[(sp++) * BLOCK] = make_leaf_ref(first + 1u, count - 1u);
          stk[(sp++) * BLOCK] = kRefMarker;
          LPT_XDEPTH();
          const float4* oq = record_words(sc.otrav, first);
          const float4 r0 = oq[0], r1 = oq[1], r2 = oq[2], r3 = oq[3];
          const float rows[12] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w};
          oo = xform_point(rows, wo);
          od = xform_dir(rows, wd);
          curObj = first;
          inObj = true;
          cur = as_uint(r3.x);
          // the reciprocal direction only feeds box tests: not needed when the mesh tree is one leaf
          if (!(cur & kRefLeafBit)) oinv = f3{1.0f / od.x, 1.0f / od.y, 1.0f / od.z};
        }
        stepped = true;
      }
    }

    if (GUARD && tooDeep) {  // rare: the ray is left to extend_deep, the lane is free for the next one
      deep.index[atomicAdd(deep.count, 1u)] = rayIdx;
      state = kStFetch;
      stepped = false;
    }

    // Common tail: take the next reference off the stack if needed and decide what the
    // lane waits for next.
    if (stepped) {
      if (needPop) cur = stk[(--sp) * BLOCK];
      if (cur == kRefMarker) {  // only a pop yields the marker
        inObj = false;
        cur = stk[(--sp) * BLOCK];
      }
      if (cur == kRefSentinel) {
#ifdef LPT_EXTEND_STATS
        atomicAdd(&g_extendStats[16 + min(xsSteps >> 5, 63u)], 1ull);
        atomicAdd(&g_extendDepth[min(xsDepth, 63)], 1ull);
        xsSteps = 0;
        xsDepth = 0;
#endif
        hits[rayIdx] = make_float4(hit.t, as_float(hit.obj), as_float(hit.prim), 0.0f);
        state = kStFetch;
      } else if (!(cur & kRefLeafBit)) {
        state = kStInner;
      } else if (inObj) {
        triI = leaf_first(cur);
        triEnd = triI + leaf_count(cur);
        state = kStTri;
      } else {
        state = kStObj;
      }
    }
  }
#ifdef LPT_EXTEND_STATS
  if (lane == 0) {
    for (int k = 0; k < 6; ++k) {
      atomicAdd(&g_extendStats[2 * k], (unsigned long long)xsR[k]);
      atomicAdd(&g_extendStats[2 * k + 1], (unsigned long long)xsL[k]);
    }
    const unsigned long long t1 = wall_clock64();
    atomicMax(&g_extendStats[12], t1 - (xsTx ? xsTx : t1));
    atomicAdd(&g_extendStats[14], t1 - (xsTx ? xsTx : t1));
    atomicAdd(&g_extendStats[15], t1 - xsT0);
  }
#endif
}

// The rays a GUARDed launch listed, one per wave (a ray is a chain of dependent loads; rays side by side in a wave
// would wait for the longest of them), searched by lane 0 with its stack in global memory: `columns` holds
// kDeepEntries words per wave of the grid.  intersect_scene is the walk extend_persistent's steps are cut from.
constexpr int kDeepEntries = 64;  // the deepest stack an uploaded scene may need
__device__ __forceinline__ void extend_deep(const DScene& sc, const float4* __restrict__ qo, const float4* __restrict__ qd,
                                            const DeepList& deep, float4* __restrict__ hits, uint32_t* columns) {
  const uint32_t n = *deep.count;
  const uint32_t wavesPerBlock = blockDim.x >> 6, wave = blockIdx.x * wavesPerBlock + (threadIdx.x >> 6);
  if ((threadIdx.x & 63u) != 0u) return;
  for (uint32_t k = wave; k < n; k += gridDim.x * wavesPerBlock) {
    const uint32_t i = deep.index[k];
    const float4 o = qo[i], d = qd[i];
    const Hit h = intersect_scene(sc, f3{o.x, o.y, o.z}, f3{d.x, d.y, d.z}, columns + (size_t)wave * kDeepEntries, 1);
    hits[i] = make_float4(h.t, as_float(h.obj), as_float(h.prim), 0.0f);
  }
}

}  // namespace lpt
