// Persistent-thread path tracer (k_trace): whole paths stay in a lane from ray generation
// to the final radiance, so a batch of frames is ONE launch with no ray queues in memory.
//
// A lane's work is cut into unit steps — one INNER node, one TRIangle, one OBJect entry of the
// closest-hit search (exactly the sequence of intersect_scene, lpt_device.h), and the
// "surface" steps: shade a found hit, one order of the Heitz random walk, start a new path.
// Each round the wave runs one kind of step for the lanes that wait for it.  The search
// kinds are picked by majority.  The surface kinds are costly and rare, so their lanes
// wait until a fair number has gathered and then run together in one round: shade, one walk
// order (lanes fresh from shading take their first order in the same round), the end of the
// bounce, and a new path for every lane whose path ended.
//
// Per lane the arithmetic and the random numbers drawn are those of the wavefront kernels
// (generate_ray, shade_begin, heitz_walk_step, shade_end), so the images are identical.
#pragma once
#include "lpt_extend.h"

#pragma clang fp contract(off)

namespace lpt {

#ifndef LPT_SURFACE_THRESHOLD
#define LPT_SURFACE_THRESHOLD 40
#endif
// Queue-fed launches have no costly ray generation in their surface rounds: a lower threshold pays (+1 %).
#ifndef LPT_SURFACE_THRESHOLD_QUEUED
#define LPT_SURFACE_THRESHOLD_QUEUED 32
#endif
// New paths are started when at least this many lanes have none (or the wave has nothing else to do):
// ray generation is costly and would otherwise run for a handful of lanes in every surface round.
#ifndef LPT_NEW_MIN
#define LPT_NEW_MIN 12
#endif
// A surface round repeats the walk order while at least this many lanes are still in a walk.
#ifndef LPT_NEW_MIN_QUEUED
#define LPT_NEW_MIN_QUEUED 2
#endif
#ifndef LPT_WALK_STAY
#define LPT_WALK_STAY 4
#endif
// Upper bound on the rounds of one wave, far above any real batch (a wave of the largest batch,
// 256 Mi paths, runs a few hundred thousand): an exit every wave reaches within seconds even if
// a lane's state were corrupted.  A wave that runs into it raises flags[0], and the next call that
// synchronises with the stream returns LPT_ERR_TRUNCATED.  The limit in force is a launch argument
// (default: this constant; lpt_debug_set_trace_round_limit lowers it for tests).
#ifndef LPT_TRACE_ROUND_LIMIT
#define LPT_TRACE_ROUND_LIMIT (1u << 24)
#endif
// Words of the totals array the kernels add to, one cache line apart (every wave of a launch adds to them when
// it ends, and atomics on one line are carried out one after the other); [2..15] belong to the analysis builds.
constexpr int kTotPaths = 0, kTotTimed = 16, kTotSegments = 32, kTotalsWords = 48;
constexpr int kTsNew = 0, kTsInner = 1, kTsTri = 2, kTsObj = 3, kTsShade = 4, kTsWalk = 5, kTsDone = 6;

template <int S, int BLOCK>
struct TraceShared {
  uint32_t stack[S * BLOCK];
};

// Where the paths of a launch come from when QUEUED: the part queues (QueueFill, lpt_extend.h) of a path queue
// (origin | path-id word, direction | rng.x, throughput | rng.y) written by the shading kernel of bounce
// `bounce` - 1; the paths go on from bounce `bounce`.
struct TraceQueue {
  const float4* o;
  const float4* d;
  const float4* m;
  QueueFill fill;
  uint32_t bounce;
  uint32_t timed;  // this launch is the one whose time the statistics report: its segments go to totals[kTotTimed] too
};

// `cursor` holds one zeroed set of cursors (kCursorWords words); `totals[kTotPaths]` counts paths, `totals[kTotSegments]` ray segments, `totals[kTotTimed]`
// the segments of timed launches
// (QUEUED: only the segments traced here are added).
template <int S, int BLOCK, bool QUEUED>
__device__ __forceinline__ void trace_persistent(const DScene& sc, const Partition& P,
                                                 const lpt_ubo* __restrict__ ubos, uint32_t nFrames,
                                                 uint32_t maxBounces, int bsdfMode, float4* __restrict__ samples,
                                                 uint32_t* cursor, unsigned long long* totals,
                                                 TraceShared<S, BLOCK>& sh, const TraceQueue& src, uint32_t* flags,
                                                 uint32_t roundLimit) {
  uint32_t* stk = &sh.stack[threadIdx.x];
  const uint32_t lane = threadIdx.x & 63u;
  // path ids of the batch (frame-major, slot-minor) cut into equal regions, or the queue's parts
  const uint32_t n = QUEUED ? 0u : nFrames * P.nSlots;

  // wave-uniform chunk of path ids [chunkNext, chunkEnd)
  uint32_t chunkNext = 0, chunkEnd = 0;
  bool exhausted = false;
  const uint32_t regionSize = (n + kCursorRegions - 1u) / kCursorRegions;
  uint32_t region = blockIdx.x % kCursorRegions;
  uint32_t regionsTried = 0;

  int state = kTsNew;
  Path p;  // p.o / p.d double as the world-space ray of the search
  p.o = p.d = p.mask = p.acc = f3{0, 0, 0};
  p.rng = Rng{0u, 0u};
  uint32_t pid = 0, bounce = 0;
  uint32_t nSegments = 0, nPaths = 0;

  // A lane is either searching for a hit or sampling the BSDF at one, never both, so the two
  // sets of per-lane variables share registers (what keeps four waves per SIMD resident):
  //            search                         surface (state kTsWalk)
  //   A, B, C  ray in the object's space      shading frame u, v, ff
  //            (read while inObj; outside an object the search reads p.o, p.d, winv)
  //   D        winv                           baseLinear
  //   E        hit (t, obj, prim)             walk energy
  //   g0..g5   cur, sp, curObj, triI,         walk height, roughness, ior,
  //            triEnd, inObj                  lobe | order | sides
  //   p.o p.d  world-space ray                next origin (hit point), walk direction
  f3 A{0, 0, 0}, B{0, 0, 0}, C{0, 0, 0}, D{0, 0, 0}, E{0, 0, 0};
  uint32_t g0 = 0, g1 = 0, g2 = 0, g3 = 0, g4 = 0, g5 = 0;

#ifdef LPT_TRACE_STATS
  uint32_t stRounds[5] = {0, 0, 0, 0, 0};
  uint32_t stLanes[5] = {0, 0, 0, 0, 0};
  // wave time per kind of round (0 inner, 1 tri, 2 obj, 3 surface incl. new paths), in clock64 ticks
  unsigned long long stTicks[4] = {0, 0, 0, 0};
  long long stT0 = clock64();
  int stKind = 3;
#define LPT_STAT(k, lanes) (++stRounds[k], stLanes[k] += (uint32_t)(lanes), stKind = (k) > 3 ? 3 : (k))
#else
#define LPT_STAT(k, lanes) ((void)0)
#endif
  for (uint32_t round = 0; round < roundLimit; ++round) {
#ifdef LPT_TRACE_STATS
    {
      const long long now = clock64();
      stTicks[stKind] += (unsigned long long)(now - stT0);
      stT0 = now;
    }
#endif
    const unsigned long long bI = __ballot(state == kTsInner);
    const unsigned long long bT = __ballot(state == kTsTri);
    const unsigned long long bO = __ballot(state == kTsObj);
    const unsigned long long bSurf = __ballot(state == kTsNew || state == kTsShade || state == kTsWalk);
    const int nI = __popcll(bI), nT = __popcll(bT), nO = __popcll(bO), nSurf = __popcll(bSurf);
    const int busy = nI + nT + nO;
    if (busy + nSurf == 0) break;

    if (nSurf > 0 && (nSurf >= (QUEUED ? LPT_SURFACE_THRESHOLD_QUEUED : LPT_SURFACE_THRESHOLD) || busy == 0)) {
      // ---- surface round ------------------------------------------------------------
      LPT_STAT(3, nSurf);
      bool finish = false;  // BSDF sampled: weight / lightDir are valid
      f3 weight{0, 0, 0}, lightDir{0, 0, 0};
      f3 sample{0, 0, 0};
      bool ended = false;        // path over: `sample` is its radiance
      bool startSearch = false;  // this lane holds a new ray in p.o / p.d

      // surface view of the shared registers (meaningful for lanes in kTsWalk)
      Surface sf;
      sf.u = A;
      sf.v = B;
      sf.ff = C;
      sf.baseLinear = D;
      sf.posW = p.o;
      sf.viewDir = f3{0, 0, 0};
      sf.roughness = as_float(g1);
      sf.ior = as_float(g2);
      sf.transmission = 0.0f;
      sf.interaction = g3 & 7u;
      sf.outside = (g3 >> 3) & 1u;
      HeitzWalk walk;
      walk.lightDir = p.d;
      walk.energy = E;
      walk.height = as_float(g0);
      walk.outside = (g3 >> 4) & 1u;
      walk.order = (int)(g3 >> 8);

      if (state == kTsShade) {
        const Hit hit{E.x, as_uint(E.y), as_uint(E.z)};
        if (!shade_begin(sc, p, hit, sf, sample)) {
          ended = true;
        } else if (bounce + 1u >= maxBounces) {  // the path ends here whatever direction is sampled
          sample = p.acc;
          ended = true;
        } else if (bsdfMode != LPT_BSDF_HEITZ) {
          weight = basic_bsdf(p.rng, sf, lightDir);
          finish = true;
        } else {
          heitz_walk_begin(walk, sf.viewDir);
          state = kTsWalk;
        }
      }
      // walk orders: all lanes that are in a walk take one; while a fair number is still walking
      // the round goes on with them (most walks end after two orders)
      for (;;) {
        bool walking = state == kTsWalk && !finish && !ended;
        if (walking) {
          const float alpha = sf.roughness * sf.roughness;
          const float iorOut = sf.outside ? 1.0f : sf.ior;
          const float iorIn = sf.outside ? sf.ior : 1.0f;
          if (heitz_walk_step(p.rng, walk, sf.interaction, sf.baseLinear, alpha, iorOut, iorIn)) {
            weight = heitz_walk_result(walk, sf.interaction, sf.baseLinear, lightDir);
            finish = true;
            walking = false;
          }
        }
        if (__popcll(__ballot(walking)) < LPT_WALK_STAY) break;
      }
      if (finish) {
        if (shade_end(p, sf, weight, lightDir, bounce, maxBounces, sample)) {
          ++bounce;
          startSearch = true;
        } else {
          ended = true;
        }
      }
      if (ended) {
        samples[pid] = make_float4(sample.x, sample.y, sample.z, 0.0f);
        state = kTsNew;
      }
      if (state == kTsWalk && !finish) {  // the walk goes on in a later round
        A = sf.u;
        B = sf.v;
        C = sf.ff;
        D = sf.baseLinear;
        E = walk.energy;
        p.o = sf.posW;
        p.d = walk.lightDir;
        g0 = as_uint(walk.height);
        g1 = as_uint(sf.roughness);
        g2 = as_uint(sf.ior);
        g3 = sf.interaction | (sf.outside ? 8u : 0u) | (walk.outside ? 16u : 0u) | ((uint32_t)walk.order << 8);
      }

      // new paths for the lanes without one
      const unsigned long long bN = __ballot(state == kTsNew);
      const int nN = __popcll(bN);
      const bool othersActive = __ballot(state != kTsNew && state != kTsDone) != 0ull;
      // (taking a path off the queue is cheap next to generating a camera ray: no need to wait for many)
      if (nN > 0 && (nN >= (QUEUED ? LPT_NEW_MIN_QUEUED : LPT_NEW_MIN) || !othersActive)) {
        LPT_STAT(4, nN);
        while (!exhausted && chunkNext >= chunkEnd) {
          // region = [lo, lo + size): a part of the queue, or an equal share of the batch's path ids
          const uint32_t lo = QUEUED ? region * src.fill.capPart : min(n, region * regionSize);
          const uint32_t size = QUEUED ? part_fill(src.fill, region) : min(regionSize, n - lo);
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
          if (state == kTsNew) state = kTsDone;
        } else {
          const uint32_t avail = chunkEnd - chunkNext;
          if (state == kTsNew) {
            const uint32_t rank =
                __builtin_amdgcn_mbcnt_hi((uint32_t)(bN >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bN, 0u));
            if (rank < avail) {
              if (QUEUED) {
                const uint32_t i = chunkNext + rank;
                const float4 o = src.o[i], d = src.d[i], m = src.m[i];
                p.o = f3{o.x, o.y, o.z};
                p.d = f3{d.x, d.y, d.z};
                p.mask = f3{m.x, m.y, m.z};
                p.rng = Rng{as_uint(d.w), as_uint(m.w)};
                const uint32_t pidWord = as_uint(o.w);
                pid = pidWord & ~kAccStored;
                p.acc = f3{0.0f, 0.0f, 0.0f};
                if (pidWord & kAccStored) {
                  const float4 g = samples[pid];
                  p.acc = f3{g.x, g.y, g.z};
                }
                bounce = src.bounce;
                ++nPaths;
                startSearch = true;
              } else {
                pid = chunkNext + rank;
                const uint32_t f = pid / P.nSlots, slot = pid - f * P.nSlots;
                int x, y;
                if (slot_to_pixel(P, slot, x, y)) {
                  generate_ray(ubos[f], x, y, (float)P.W, (float)P.H, p);
                  bounce = 0;
                  ++nPaths;
                  startSearch = true;
                }
              }
            }
          }
          chunkNext += min((uint32_t)nN, avail);
        }
      }

      if (startSearch) {  // search view: a fresh closest-hit search along p.o / p.d
        ++nSegments;
        D = f3{1.0f / p.d.x, 1.0f / p.d.y, 1.0f / p.d.z};
        E = f3{kInfinity, 0.0f, 0.0f};
        g5 = 0u;
        stk[0] = kRefSentinel;
        g1 = 1u;
        g0 = sc.rootRef;
        if (g0 == kRefSentinel) state = kTsShade;
        else if (!(g0 & kRefLeafBit)) state = kTsInner;
        else state = kTsObj;
      }
      continue;
    }

    // ---- search rounds: the kind most lanes wait for ------------------------------------
    // the object-space ray has one writer (the object entry) and `inObj` says which ray a step reads, so
    // entering and leaving an object moves no ray data around
    f3 oo = A, od = B, oinv = C;
    const f3 winv = D;
    Hit hit{E.x, as_uint(E.y), as_uint(E.z)};
    uint32_t cur = g0, curObj = g2, triI = g3, triEnd = g4;
    int sp = (int)g1;
    bool inObj = g5 != 0u;
    bool stepped = false;  // this lane advanced and holds a new `cur` (or needs a pop)
    bool needPop = false;

    if (nI >= nT && nI >= nO) {
      LPT_STAT(0, nI);
      if (state == kTsInner) {  // burst over consecutive inner nodes, as in k_extend
        const f3 co = inObj ? oo : p.o, cinv = inObj ? oinv : winv;
        for (;;) {
          const float4* q = record_words(sc.nodes, cur);
          const float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
          bool h0, h1;
          box_test_pair(co, cinv, q0, q1, q2, hit.t, h0, h1);
          const uint32_t r0 = as_uint(q3.x), r1 = as_uint(q3.y);
          if (h0 && h1) stk[(sp++) * BLOCK] = r0;
          cur = h1 ? r1 : r0;
          if (!(h0 || h1)) cur = stk[(--sp) * BLOCK];
          if (cur >= kRefMarker) break;  // no longer an inner node: this lane leaves the burst
          // the lanes still here go on while they are enough of those that entered
          if (LPT_BURST_INNER_DEN * __popcll(__ballot(1)) < LPT_BURST_INNER_NUM * nI) break;
        }
        stepped = true;  // `cur` already holds the next reference
      }
    } else if (nT >= nO) {
      LPT_STAT(1, nT);
      if (state == kTsTri) {
        if (triI < triEnd) {
          // the whole leaf, as in k_extend; the next triangle is loaded while the current one is tested
          const float4* tq = record_words(sc.tris, triI);
          float4 a = tq[0], b = tq[1], c = tq[2];
          for (int it = 1;; ++it) {
            float4 na = a, nb = b, nc = c;
            if (triI + 1u < triEnd) {
              na = tq[3];
              nb = tq[4];
              nc = tq[5];
            }
            const float td = tri_test(oo, od, f3{a.x, a.y, a.z}, f3{b.x, b.y, b.z}, f3{c.x, c.y, c.z});
            if (td > kEps && td < hit.t) {
              hit.t = td;
              hit.obj = curObj;
              hit.prim = as_uint(a.w);
            }
            if (++triI >= triEnd) break;  // leaf done: this lane leaves the burst
            if (it >= kBurstTriFree && LPT_BURST_TRI_DEN * __popcll(__ballot(1)) < LPT_BURST_TRI_NUM * nT) break;
            tq += 3;
            a = na;
            b = nb;
            c = nc;
          }
        }
        if (triI >= triEnd) {
          needPop = true;
          stepped = true;
        } else {  // stays in this leaf: keep the progress
          E = f3{hit.t, as_float(hit.obj), as_float(hit.prim)};
          g3 = triI;
        }
      }
    } else {
      LPT_STAT(2, nO);
      if (state == kTsObj) {
        const uint32_t first = leaf_first(cur);
        const uint32_t count = leaf_count(cur);
        if (count == 0u) {
          needPop = true;
        } else {
          if (count > 1u) stk[(sp++) * BLOCK] = make_leaf_ref(first + 1u, count - 1u);
          stk[(sp++) * BLOCK] = kRefMarker;
          const float4* oq = record_words(sc.otrav, first);
          const float4 r0 = oq[0], r1 = oq[1], r2 = oq[2], r3 = oq[3];
          const float rows[12] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w};
          oo = xform_point(rows, p.o);
          od = xform_dir(rows, p.d);
          curObj = first;
          inObj = true;
          cur = as_uint(r3.x);
          // the reciprocal direction only feeds box tests: not needed when the mesh tree is one leaf
          if (!(cur & kRefLeafBit)) oinv = f3{1.0f / od.x, 1.0f / od.y, 1.0f / od.z};
        }
        stepped = true;
      }
    }

    // Common tail of the search steps: take the next reference off the stack if needed and
    // decide what the lane waits for next.
    if (stepped) {
      if (needPop) cur = stk[(--sp) * BLOCK];
      if (cur == kRefMarker) {  // only a pop yields the marker
        inObj = false;
        cur = stk[(--sp) * BLOCK];
      }
      if (cur == kRefSentinel) {
        state = kTsShade;
      } else if (!(cur & kRefLeafBit)) {
        state = kTsInner;
      } else if (inObj) {
        triI = leaf_first(cur);
        triEnd = triI + leaf_count(cur);
        state = kTsTri;
      } else {
        state = kTsObj;
      }
      A = oo;
      B = od;
      C = oinv;
      E = f3{hit.t, as_float(hit.obj), as_float(hit.prim)};
      g0 = cur;
      g1 = (uint32_t)sp;
      g2 = curObj;
      g3 = triI;
      g4 = triEnd;
      g5 = inObj ? 1u : 0u;
    }
  }

  // Out of rounds with paths unfinished: their samples are missing, and the host must hear of it
  // (flags[0] is host memory the host reads after synchronising with the stream).
  if (__ballot(state != kTsDone) != 0ull && lane == 0u) flags[0] = 1u;

  // per-wave totals, one atomic pair per wave
  for (int off = 32; off > 0; off >>= 1) {
    nSegments += (uint32_t)__shfl_down((int)nSegments, off);
    nPaths += (uint32_t)__shfl_down((int)nPaths, off);
  }
  if (lane == 0 && nPaths) {
    if (!QUEUED) atomicAdd(&totals[kTotPaths], (unsigned long long)nPaths);
    atomicAdd(&totals[kTotSegments], (unsigned long long)nSegments);
    if (src.timed) atomicAdd(&totals[kTotTimed], (unsigned long long)nSegments);
  }
#ifdef LPT_TRACE_STATS
  if (lane == 0)
    for (int k = 0; k < 5; ++k) {
      atomicAdd(&totals[2 + 2 * k], (unsigned long long)stRounds[k]);
      atomicAdd(&totals[3 + 2 * k], (unsigned long long)stLanes[k]);
      if (k < 4) atomicAdd(&totals[12 + k], stTicks[k]);
    }
#endif
}

}  // namespace lpt
