// Persistent-thread closest-hit kernel (k_extend).
//
// Every lane walks its ray through exactly the sequence of node visits, box tests and
// triangle tests that intersect_scene (lpt_device.h) performs, so hits are identical.
// What differs is the scheduling across the 64 lanes of a wave: a lane's traversal is
// cut into unit steps of four kinds — FETCH a new ray, one INNER node (two box tests),
// one TRIangle test, enter one OBJect — and in each round the wave runs the kind that
// most lanes are waiting for.  Lanes of other kinds simply wait; waiting never changes a
// lane's own sequence.  This keeps the SIMD lanes of each issued instruction well filled
// where a per-lane "if leaf / else inner" loop would run every branch with few lanes.
//
// Rays are handed out in chunks: one atomic on a global cursor per chunk and wave, the
// chunk is then consumed lane by lane as rays finish (no idle tail inside a wave).
#pragma once
#include "lpt_device.h"

#pragma clang fp contract(off)

namespace lpt {

constexpr uint32_t kFetchChunk = 512;  // rays per cursor atomic
constexpr int kStFetch = 0, kStInner = 1, kStTri = 2, kStObj = 3, kStDone = 4;

template <int S, int BLOCK>
__device__ __forceinline__ void extend_persistent(const DScene& sc, const float4* __restrict__ qo,
                                                  const float4* __restrict__ qd, uint32_t n,
                                                  float4* __restrict__ hits, uint32_t* cursor, uint32_t* stack) {
  uint32_t* stk = &stack[threadIdx.x];
  const uint32_t lane = threadIdx.x & 63u;

  // wave-uniform chunk of ray indices [chunkNext, chunkEnd)
  uint32_t chunkNext = 0, chunkEnd = 0;
  bool exhausted = false;

  int state = kStFetch;
  f3 wo{0, 0, 0}, wd{0, 0, 0}, winv{0, 0, 0};  // ray in world space
  f3 co{0, 0, 0}, cd{0, 0, 0}, cinv{0, 0, 0};  // ray in the space being traversed
  Hit hit{kInfinity, 0u, 0u};
  bool inObj = false;
  uint32_t curObj = 0, triBase = 0, triI = 0, triEnd = 0;
  uint32_t cur = kRefSentinel, rayIdx = 0;
  int sp = 0;

  // Pops the next reference; leaving an object's mesh tree restores the world-space ray.
  auto pop = [&]() {
    cur = stk[(--sp) * BLOCK];
    if (cur == kRefMarker) {
      inObj = false;
      co = wo;
      cd = wd;
      cinv = winv;
      cur = stk[(--sp) * BLOCK];
    }
  };
  // Decides what the lane needs next from the reference it now holds.
  auto dispatch = [&]() {
    if (cur == kRefSentinel) {
      hits[rayIdx] = make_float4(hit.t, as_float(hit.obj), as_float(hit.prim), 0.0f);
      state = kStFetch;
    } else if (!(cur & kRefLeafBit)) {
      state = kStInner;
    } else if (inObj) {
      triI = cur & 0x00FFFFFFu;
      triEnd = triI + ((cur >> 24) & 0x7Fu);
      state = kStTri;
    } else {
      state = kStObj;
    }
  };

  for (;;) {
    const unsigned long long bF = __ballot(state == kStFetch);
    const unsigned long long bI = __ballot(state == kStInner);
    const unsigned long long bT = __ballot(state == kStTri);
    const unsigned long long bO = __ballot(state == kStObj);
    const int nF = __popcll(bF), nI = __popcll(bI), nT = __popcll(bT), nO = __popcll(bO);
    const int busy = nI + nT + nO;
    if (busy + nF == 0) break;

    // Idle lanes are pure loss, so refill as soon as a fair number waits (or nothing else is left).
    if (nF > 0 && (nF >= 16 || busy == 0)) {
      if (!exhausted && chunkNext >= chunkEnd) {
        uint32_t b = 0;
        if (lane == 0) b = atomicAdd(cursor, kFetchChunk);
        b = (uint32_t)__builtin_amdgcn_readfirstlane((int)__shfl((int)b, 0));
        if (b >= n) {
          exhausted = true;
        } else {
          chunkNext = b;
          chunkEnd = min(b + kFetchChunk, n);
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
            co = wo;
            cd = wd;
            cinv = winv;
            hit = Hit{kInfinity, 0u, 0u};
            inObj = false;
            stk[0] = kRefSentinel;
            sp = 1;
            cur = sc.tlasRootRef;
            dispatch();
          }
        }
        chunkNext += min((uint32_t)nF, avail);
      }
      continue;
    }

    if (nI >= nT && nI >= nO) {
      if (state == kStInner) {
        const TNode* nodes = inObj ? sc.blas : sc.tlas;
        const float4* q = reinterpret_cast<const float4*>(nodes + cur);
        const float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
        const float c0mn[3] = {q0.x, q0.y, q0.z}, c0mx[3] = {q0.w, q1.x, q1.y};
        const float c1mn[3] = {q1.z, q1.w, q2.x}, c1mx[3] = {q2.y, q2.z, q2.w};
        const bool h0 = box_test(co, cinv, c0mn, c0mx, hit.t);
        const bool h1 = box_test(co, cinv, c1mn, c1mx, hit.t);
        const uint32_t r0 = as_uint(q3.x), r1 = as_uint(q3.y);
        if (h1) {
          if (h0) stk[(sp++) * BLOCK] = r0;
          cur = r1;
        } else if (h0) {
          cur = r0;
        } else {
          pop();
        }
        dispatch();
      }
    } else if (nT >= nO) {
      if (state == kStTri) {
        if (triI < triEnd) {
          const float4* tq = reinterpret_cast<const float4*>(sc.tris + (triBase + triI));
          const float4 a = tq[0], b = tq[1], c = tq[2];
          const float td = tri_test(co, cd, f3{a.x, a.y, a.z}, f3{b.x, b.y, b.z}, f3{c.x, c.y, c.z});
          if (td > kEps && td < hit.t) {
            hit.t = td;
            hit.obj = curObj;
            hit.prim = 3u * (triBase + triI);
          }
          ++triI;
        }
        if (triI >= triEnd) {
          pop();
          dispatch();
        }
      }
    } else {
      if (state == kStObj) {
        const uint32_t first = cur & 0x00FFFFFFu;
        const uint32_t count = (cur >> 24) & 0x7Fu;
        if (count == 0u) {
          pop();
        } else {
          if (count > 1u) stk[(sp++) * BLOCK] = make_leaf_ref(first + 1u, count - 1u);
          stk[(sp++) * BLOCK] = kRefMarker;
          const float4* oq = reinterpret_cast<const float4*>(sc.otrav + first);
          const float4 r0 = oq[0], r1 = oq[1], r2 = oq[2], r3 = oq[3];
          const float rows[12] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w};
          co = xform_point(rows, wo);
          cd = xform_dir(rows, wd);
          cinv = f3{1.0f / cd.x, 1.0f / cd.y, 1.0f / cd.z};
          curObj = first;
          triBase = as_uint(r3.y);
          inObj = true;
          cur = as_uint(r3.x);
        }
        dispatch();
      }
    }
  }
}

}  // namespace lpt
