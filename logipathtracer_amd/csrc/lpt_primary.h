// Closest hits of the camera rays, one 8x8 pixel block per wave, walked as a PACKET (k_primary).
//
// The 64 camera rays of a pixel block run through nearly the same nodes.  Here the wave visits
// the union of its lanes' visits in one depth-first walk with a wave-wide stack; every entry
// carries the mask of the lanes that want that node.  A lane takes part in exactly the nodes its
// own walk (intersect_scene, lpt_device.h) would visit and in the same order: a lane is in the
// mask of a child iff it was in the parent's mask and its own box test of that child passed
// against its own closest hit at that moment, and a depth-first order of the union, restricted to
// one lane's nodes, is that lane's depth-first order (same child order: y before x).  So every
// lane performs the box and triangle tests of its own walk with the same operands — hits are
// identical — while the node, triangle and object records are the same for all lanes of a step:
// they are read with wave-uniform addresses (scalar loads), not gathered per lane.
//
// The stack lives in registers, entry k in lane k (depth <= 64): a push is a write in one lane,
// a pop three v_readlane.
#pragma once
#include "lpt_device.h"

#pragma clang fp contract(off)

namespace lpt {

struct PacketStack {
  uint32_t ref, maskLo, maskHi;  // this lane's entry
};

__device__ __forceinline__ void packet_push(PacketStack& st, int& sp, uint32_t lane, uint32_t ref,
                                            unsigned long long mask) {
  if (lane == (uint32_t)sp) {
    st.ref = ref;
    st.maskLo = (uint32_t)mask;
    st.maskHi = (uint32_t)(mask >> 32);
  }
  ++sp;
}

__device__ __forceinline__ void packet_pop(const PacketStack& st, int& sp, uint32_t& ref, unsigned long long& mask) {
  --sp;
  ref = (uint32_t)__builtin_amdgcn_readlane((int)st.ref, sp);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)st.maskLo, sp);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)st.maskHi, sp);
  mask = ((unsigned long long)hi << 32) | lo;
}

// What the packet kernel does with the camera rays it generates, besides tracing them: queue them for the
// shading kernel (origin | path id, direction | rng.x, throughput | rng.y, as k_generate does).
struct PrimaryQueue {
  float4* o;
  float4* d;
  float4* m;
};

// Path ids are frame-major, slot-minor (64 consecutive slots are one 8x8 pixel block).  A wave's packet holds
// 64 / packetFrames consecutive slots in packetFrames consecutive frames (packetFrames: a power of two <= 64 that
// divides nFrames; lane = frame offset * slots per packet + slot offset): with 1 it is one pixel block of one frame,
// with 64 one pixel in 64 frames, whose camera rays differ by the jitter inside the pixel only and so walk through
// nearly the same nodes and triangles — the packet's walk is the union of its lanes' walks, and that union is what
// it costs.  The packets of the launch are dealt out slot-major: consecutive waves hold the same slots in
// consecutive frames, so the waves resident on the chip at one time look at a small part of the scene (they share
// its records in the caches), and the paths that go on are queued region by region (measured against frame-major:
// +1.3 % on the Cornell box, +1.8 % on a 400 k-triangle terrain scene).
// The number of slots must be a multiple of 64.  The camera ray is generated here and traced; on return
// `valid` says whether the path id is a pixel of the image, `p` is the path at its first segment and `hit`
// what that segment found.  With STORE the ray is also queued at index f*nValid + validBefore[slot] (or the
// path id itself when validBefore is null: every slot a pixel) and hits[same index] = (t, obj, prim, 0).
// `w` is the number of the wave's packet in the launch (wave-uniform; a number beyond the last packet is a packet
// without pixels).  Every thread returns (no early exit), so the caller may synchronise the workgroup afterwards.
template <bool STORE>
__device__ __forceinline__ void primary_packet(const TNode* __restrict__ nodes, const Tri* __restrict__ tris,
                                               const ObjTrav* __restrict__ otrav, uint32_t rootRef,
                                               const Partition& P, const lpt_ubo* __restrict__ ubos,
                                               uint32_t nFrames, uint32_t packetFrames,
                                               const uint32_t* __restrict__ validBefore, uint32_t nValid,
                                               const PrimaryQueue& q, float4* __restrict__ hits, uint32_t w, Path& p,
                                               Hit& hit, bool& valid, uint32_t& pid) {
  const uint32_t lane = threadIdx.x & 63u;
  // packet w of the launch: slots [sb * perPacket, (sb + 1) * perPacket) in frames [fw, fw + packetFrames)
  const uint32_t perPacket = 64u / packetFrames, chunks = nFrames / packetFrames;
  const uint32_t sb = w / chunks, fw = (w - sb * chunks) * packetFrames + lane / perPacket;
  const uint32_t slotOfLane = sb * perPacket + (lane & (perPacket - 1u));
  valid = slotOfLane < P.nSlots;
  pid = valid ? fw * P.nSlots + slotOfLane : 0xFFFFFFFFu;
  uint32_t i = pid;
  p.o = p.d = f3{0.0f, 0.0f, 1.0f};
  p.mask = p.acc = f3{0.0f, 0.0f, 0.0f};
  p.rng = Rng{0u, 0u};
  if (valid) {
    const uint32_t f = fw, slot = slotOfLane;
    int x, y;
    valid = slot_to_pixel(P, slot, x, y);
    if (valid) {
      generate_ray(ubos[f], x, y, (float)P.W, (float)P.H, p);
      if (STORE) {
        if (validBefore) i = f * nValid + validBefore[slot];
        q.o[i] = make_float4(p.o.x, p.o.y, p.o.z, as_float(pid));
        q.d[i] = make_float4(p.d.x, p.d.y, p.d.z, as_float(p.rng.x));
        q.m[i] = make_float4(p.mask.x, p.mask.y, p.mask.z, as_float(p.rng.y));
      }
    }
  }
  unsigned long long mask = __ballot(valid);

  const f3 wo = p.o, wd = p.d;
  const f3 winv{1.0f / wd.x, 1.0f / wd.y, 1.0f / wd.z};
  f3 co = wo, cd = wd, cinv = winv;
  hit = Hit{kInfinity, 0u, 0u};

  PacketStack st{0u, 0u, 0u};
  int sp = 0;
  packet_push(st, sp, lane, kRefSentinel, 0ull);
  uint32_t cur = mask != 0ull ? rootRef : kRefSentinel;  // a wave without pixels has nothing to walk
  uint32_t curObj = 0u;
  bool inObj = false;

  while (cur != kRefSentinel) {
    if (cur == kRefMarker) {  // left an object's mesh tree: back to the world-space ray
      inObj = false;
      co = wo;
      cd = wd;
      cinv = winv;
      packet_pop(st, sp, cur, mask);
      continue;
    }
    const bool in = ((mask >> lane) & 1ull) != 0ull;
    if (!(cur & kRefLeafBit)) {
      const float4* q = record_words(nodes, cur);
      const float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
      bool h0 = false, h1 = false;
      if (in) box_test_pair(co, cinv, q0, q1, q2, hit.t, h0, h1);
      const unsigned long long m0 = __ballot(h0), m1 = __ballot(h1);
      const uint32_t r0 = as_uint(q3.x), r1 = as_uint(q3.y);
      if (m1 != 0ull) {
        if (m0 != 0ull) packet_push(st, sp, lane, r0, m0);
        cur = r1;
        mask = m1;
      } else if (m0 != 0ull) {
        cur = r0;
        mask = m0;
      } else {
        packet_pop(st, sp, cur, mask);
      }
      continue;
    }
    const uint32_t first = leaf_first(cur);
    const uint32_t count = leaf_count(cur);
    if (inObj) {
      for (uint32_t t = first; t < first + count; ++t) {
        const float4* tq = record_words(tris, t);
        const float4 a = tq[0], b = tq[1], c = tq[2];
        if (in) {
          const float td = tri_test(co, cd, f3{a.x, a.y, a.z}, f3{b.x, b.y, b.z}, f3{c.x, c.y, c.z});
          if (td > kEps && td < hit.t) {
            hit.t = td;
            hit.obj = curObj;
            hit.prim = as_uint(a.w);
          }
        }
      }
      packet_pop(st, sp, cur, mask);
    } else if (count == 0u) {
      packet_pop(st, sp, cur, mask);
    } else {
      // every lane that reached this leaf enters each of its objects in turn
      if (count > 1u) packet_push(st, sp, lane, make_leaf_ref(first + 1u, count - 1u), mask);
      packet_push(st, sp, lane, kRefMarker, mask);
      const float4* oq = record_words(otrav, first);
      const float4 r0 = oq[0], r1 = oq[1], r2 = oq[2], r3 = oq[3];
      const float rows[12] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w};
      cur = as_uint(r3.x);
      if (in) {
        co = xform_point(rows, wo);
        cd = xform_dir(rows, wd);
        if (!(cur & kRefLeafBit)) cinv = f3{1.0f / cd.x, 1.0f / cd.y, 1.0f / cd.z};
      }
      curObj = first;
      inObj = true;
    }
  }
  if (STORE && valid) hits[i] = make_float4(hit.t, as_float(hit.obj), as_float(hit.prim), 0.0f);
}

}  // namespace lpt
