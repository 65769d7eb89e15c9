// HIP kernels and C-ABI of the MI355X device layer (see include/lpt.h).
//
// Four kernel organisations render identical pixels:
//  * wavefront: Structure-of-Arrays path queues in HBM; the first segment of every path in one packet
//    kernel (k_primary_shade), then per bounce one traversal kernel (k_extend, node stack in LDS) and one
//    shading kernel (k_shade) that compacts surviving paths into the next queue with a wave ballot; the
//    thinned-out late bounces in one queue-fed persistent launch (k_trace);
//  * hybrid: the first segment as above, everything after it in one queue-fed k_trace launch;
//  * persistent: one k_trace launch per batch, paths stay in a lane from ray generation to their radiance;
//  * megakernel: one thread per pixel runs the whole path (the organisation of
//    shaders/path_tracing.comp::main).
//
// Compile: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
//          -fhip-fp32-correctly-rounded-divide-sqrt (no fast-math).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <mutex>
#include <new>
#include <queue>
#include <string>
#include <vector>

#include "lpt_device.h"
#include "lpt_extend.h"
#include "lpt_trace.h"
#include "lpt_primary.h"
#include "lpt_internal.h"

#pragma clang fp contract(off)

using namespace lpt;

static_assert(sizeof(lpt_object) == 208, "GPUObjectData layout");
static_assert(sizeof(lpt_vertex) == 48, "GPUVertex layout");
static_assert(sizeof(lpt_bvh_node) == 48, "GPUBVHNode layout");
static_assert(sizeof(lpt_ubo) == 92, "PathTracerUBO layout");
static_assert(offsetof(lpt_object, bvhOffset) == 192 && offsetof(lpt_object, ior) == 168, "GPUObjectData offsets");
static_assert(offsetof(lpt_bvh_node, isLeaf) == 28 && offsetof(lpt_bvh_node, indices) == 32, "GPUBVHNode offsets");
static_assert(offsetof(lpt_ubo, seed) == 80 && offsetof(lpt_ubo, reset) == 88, "UBO offsets");

namespace {

constexpr int kBlock = 256;

// Path queue, Structure of Arrays; slot i of each array belongs to one path.
// The radiance a path has gathered does not travel with it: it lives in samples[path id], which
// is also where the finished path's radiance belongs.  Bit 31 of the path-id word says whether
// that slot already holds this path's radiance (it is set by the first emissive hit); while it
// is clear the gathered radiance is zero and the slot still holds the previous batch's data.
struct Queue {
  float4* o;  // origin.xyz, path id | kAccStored (bits)
  float4* d;  // direction.xyz, rng.x (bits)
  float4* m;  // throughput.rgb, rng.y (bits)
};

// Block-level append: every thread with `active` gets a distinct slot of a queue whose fill
// count lives in *counter.  One atomic per block: a single counter word sustains only some
// tens of atomics per microsecond, so per-wave atomics would bound the kernel.
// Must be called by all threads of the block.
__device__ __forceinline__ uint32_t block_append(bool active, uint32_t* counter, uint32_t* lds /* [kBlock/64 + 1] */) {
  const unsigned long long ballot = __ballot(active);
  const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(ballot >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ballot, 0u));
  const uint32_t wave = threadIdx.x >> 6;
  __syncthreads();  // the previous call's readers are done with lds
  if ((threadIdx.x & 63u) == 0u) lds[wave] = (uint32_t)__popcll(ballot);
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t total = 0;
    for (int w = 0; w < kBlock / 64; ++w) total += lds[w];
    lds[kBlock / 64] = total ? atomicAdd(counter, total) : 0u;
  }
  __syncthreads();
  uint32_t base = lds[kBlock / 64];
  for (uint32_t w = 0; w < wave; ++w) base += lds[w];
  return base + below;
}

__device__ __forceinline__ void store_path(const Queue& q, uint32_t slot, const Path& p, uint32_t pidWord) {
  q.o[slot] = make_float4(p.o.x, p.o.y, p.o.z, as_float(pidWord));
  q.d[slot] = make_float4(p.d.x, p.d.y, p.d.z, as_float(p.rng.x));
  q.m[slot] = make_float4(p.mask.x, p.mask.y, p.mask.z, as_float(p.rng.y));
}

// ---- wavefront kernels ---------------------------------------------------------------

// Camera rays for `nFrames` frames of the owned pixels (path id = frame*nSlots + slot).  The queue
// is filled densely and without atomics: `validBefore[slot]` is the number of slots below `slot`
// that are pixels of the image (null if every slot is one), `nValid` their total, so the ray of
// (frame f, slot) goes to f*nValid + validBefore[slot]; the host sets the queue count.
__global__ __launch_bounds__(kBlock) void k_generate(Partition P, const lpt_ubo* __restrict__ ubos, uint32_t nFrames,
                                                     Queue q, const uint32_t* __restrict__ validBefore,
                                                     uint32_t nValid) {
  const uint32_t total = nFrames * P.nSlots;
  for (uint32_t pid = blockIdx.x * kBlock + threadIdx.x; pid < total; pid += gridDim.x * kBlock) {
    const uint32_t f = pid / P.nSlots, slot = pid - f * P.nSlots;
    int x, y;
    if (slot_to_pixel(P, slot, x, y)) {
      Path p;
      generate_ray(ubos[f], x, y, (float)P.W, (float)P.H, p);
      store_path(q, validBefore ? f * nValid + validBefore[slot] : pid, p, pid);
    }
  }
}

// Closest hit for every queued ray: persistent waves, node stacks in LDS, rays handed out
// through `cursor` (one set of cursors, kCursorWords words zeroed before the launch).  See lpt_extend.h.
// Resident waves per SIMD: eight with the 16-entry stacks (64 registers, no spill; eight workgroups' stacks and 48
// cached nodes fill the LDS exactly), otherwise what registers and LDS allow (seven with 20-entry stacks, fewer with
// deeper ones).  Measured on the Cornell box with 16-entry stacks: 8 waves and 48 cached nodes 35.90 ms per step, 8 waves
// and 24 nodes 36.24, 7 waves and 80 nodes 36.73.  LPT_EXTEND_WAVES_PER_SIMD forces one value for comparison builds.
#ifdef LPT_EXTEND_WAVES_PER_SIMD
#define LPT_EXTEND_MIN_WAVES(S) LPT_EXTEND_WAVES_PER_SIMD
#else
#define LPT_EXTEND_MIN_WAVES(S) ((S) <= 16 ? 8 : 1)
#endif
// GUARD (see extend_persistent): for scenes whose trees may want more than S entries; rays that do go to `deep`.
template <int S, bool GUARD = false>
__attribute__((amdgpu_waves_per_eu(LPT_EXTEND_MIN_WAVES(S), 8)))
__global__ __launch_bounds__(kBlock) void k_extend(DScene sc, const float4* __restrict__ qo,
                                                   const float4* __restrict__ qd, QueueFill fill,
                                                   float4* __restrict__ hits, uint32_t* cursor, DeepList deep) {
// The hottest nodes (the top of the object tree and of the largest meshes) in LDS: as many as fit beside the stacks
// without costing a resident workgroup — 48 beside eight workgroups' 16-entry stacks, 24 beside seven workgroups'
// 20-entry stacks.  Measured: 24 nodes -2 % kernel time on the Cornell box, -6 % on a 70-object scene; sizes that cost
// occupancy lose (20-entry stacks: 48 to 80 nodes +2 %; earlier, 150: +10 %, 300: +25 %).
#ifndef LPT_EXTEND_CACHE_NODES
#define LPT_EXTEND_CACHE_NODES (S <= 16 ? 48 : 24)
#endif
  constexpr int kCacheNodes = LPT_EXTEND_CACHE_NODES;
  __shared__ uint32_t stack[S * kBlock];
  __shared__ __attribute__((aligned(16))) float cache[kCacheNodes > 0 ? kCacheNodes * kCacheStride : 4];
  extend_persistent<S, kBlock, kCacheNodes, GUARD>(sc, qo, qd, fill, hits, cursor, stack, cache, deep);
}

// The rays a guarded k_extend launch listed (see extend_deep).
constexpr int kDeepBlock = 256, kDeepGrid = 256;
__global__ __launch_bounds__(kDeepBlock) void k_extend_deep(DScene sc, const float4* __restrict__ qo,
                                                            const float4* __restrict__ qd, DeepList deep,
                                                            float4* __restrict__ hits, uint32_t* columns) {
  extend_deep(sc, qo, qd, deep, hits, columns);
}

// Camera rays of a batch: generated, queued for the shading kernel and traced, 64 consecutive path ids
// (one pixel block) per wave walked as a packet with wave-uniform record loads.  See lpt_primary.h.
__global__ __launch_bounds__(kBlock) void k_primary(const TNode* __restrict__ nodes, const Tri* __restrict__ tris,
                                                    const ObjTrav* __restrict__ otrav, uint32_t rootRef, Partition P,
                                                    const lpt_ubo* __restrict__ ubos, uint32_t nFrames,
                                                    uint32_t packetFrames, const uint32_t* __restrict__ validBefore,
                                                    uint32_t nValid, PrimaryQueue q, float4* __restrict__ hits) {
  Path p;
  Hit hit;
  bool valid;
  uint32_t pid;
  primary_packet<true>(nodes, tris, otrav, rootRef, P, ubos, nFrames, packetFrames, validBefore, nValid, q, hits,
                       blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6), p, hit, valid, pid);
}

// Bookkeeping of a shaded queue entry: the radiance this bounce emitted joins what the path has gathered
// in samples[path id]; a path that ends without ever having gathered anything leaves zero there.
__device__ __forceinline__ void settle_emitted(float4* __restrict__ samples, const Path& p, uint32_t& pidWord) {
  if (p.acc.x != 0.0f || p.acc.y != 0.0f || p.acc.z != 0.0f) {
    // gathered + mask*emission, path_tracing.comp:269 (adding a zero product would change nothing)
    const uint32_t slot = pidWord & ~kAccStored;
    f3 acc = p.acc;
    if (pidWord & kAccStored) {
      const float4 g = samples[slot];
      acc = f3{g.x, g.y, g.z} + p.acc;
    }
    samples[slot] = make_float4(acc.x, acc.y, acc.z, 0.0f);
    pidWord |= kAccStored;
  }
}
__device__ __forceinline__ void settle_ended(float4* __restrict__ samples, uint32_t pidWord) {
  if (!(pidWord & kAccStored)) samples[pidWord & ~kAccStored] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);  // it never gathered anything
}
__device__ __forceinline__ void settle_entry(float4* __restrict__ samples, const Path& p, uint32_t& pidWord, bool alive) {
  settle_emitted(samples, p, pidWord);
  if (!alive) settle_ended(samples, pidWord);
}

// ---- Heitz walks of a workgroup: the pool ---------------------------------------------------
// The random walks of a group of hits differ widely in length on rough materials (1 to 16 orders), and within
// an order most of the work (the scattering event) is only done by the walks that have not just left the
// microsurface: a loop per lane leaves most of a wave idle, and finishing the walks group by group keeps whole
// waves busy for the last handful (on rough materials that took two thirds of the shading kernel's time).
// Here every lane takes the start of its own walk — height, scattering, height: where most walks on smooth
// materials end.  A walk that is then about to scatter again is set aside: its state goes to a pool in LDS,
// what the rest of the bounce needs from the hit (position, normal, throughput, path id) goes to a queue entry
// that belongs to the path (`cont`), and the thread moves on to the next group of hits.  Whenever kPoolRun walks
// wait, the workgroup advances whole waves of them by LPT_WALK_PASS_STEPS orders; a walk that ends finishes its
// path there.  The pool is a stack: a pass takes the walks on top and puts back those that go on.  What is left
// when the workgroup runs out of hits is run to the end.  The operations and random numbers of a walk are those
// of heitz_bsdf_unified in the same order, and every path owns its random-number state, so the order in which
// walks are advanced changes no result.
// Measured (1080p, 8 bounces, 64 frames per batch) against finishing the walks group by group: rough 400 k-triangle
// terrain scene, shading kernel 9.4 -> 5.7 ms per launch; Cornell box unchanged.  Steps per pass 1 / 2 / 3:
// +1 % / 0 / -0.5 %; kPoolRun 64 ... 256: within 0.5 %; no inline start (every walk through the pool): -12 % on the
// Cornell box; two scattering events inline: -1 % on the terrain.
#ifndef LPT_WALK_PASS_STEPS
#define LPT_WALK_PASS_STEPS 1
#endif
#ifndef LPT_WALK_HEAD_ORDERS  // scattering events a lane runs inline before its walk goes to the pool
#define LPT_WALK_HEAD_ORDERS 1
#endif
#ifndef LPT_WALK_POOL_RUN
#define LPT_WALK_POOL_RUN 128
#endif
constexpr uint32_t kPoolRun = LPT_WALK_POOL_RUN;
static_assert(kPoolRun >= 64u && kPoolRun <= (uint32_t)kBlock && kPoolRun % 64u == 0u, "passes take whole waves");
constexpr uint32_t kPoolSlots = kPoolRun + kBlock;  // fewer than kPoolRun wait when a group of kBlock hits is shaded
// words of a pooled walk: lightDir 0-2, energy 3-5, height 6, order | walk side << 8 | lobe << 9 | ray side << 12 in 7,
// rng 8-9, F0 10-12, alpha 13, ior 14, index of the path's entry in `cont` 15
constexpr int kWalkJobWords = 16;
constexpr uint32_t kPoolSurfaceOutside = 0x1000u;

struct WalkPool {
  float job[kWalkJobWords][kPoolSlots];  // field-major: lanes of a wave touch consecutive words
  uint32_t waveCount[kBlock / 64];
};

struct PooledWalk {
  HeitzWalk w;
  Rng rng;
  f3 F0;
  float alpha, ior;
  uint32_t kind, entry;
  bool surfaceOutside;  // the ray arrived on the side the surface normal points to
};

__device__ __forceinline__ void pool_store(WalkPool& wp, uint32_t j, const PooledWalk& k) {
  wp.job[0][j] = k.w.lightDir.x;
  wp.job[1][j] = k.w.lightDir.y;
  wp.job[2][j] = k.w.lightDir.z;
  wp.job[3][j] = k.w.energy.x;
  wp.job[4][j] = k.w.energy.y;
  wp.job[5][j] = k.w.energy.z;
  wp.job[6][j] = k.w.height;
  wp.job[7][j] = as_float((uint32_t)k.w.order | (k.w.outside ? 0x100u : 0u) | (k.kind << 9) |
                          (k.surfaceOutside ? kPoolSurfaceOutside : 0u));
  wp.job[8][j] = as_float(k.rng.x);
  wp.job[9][j] = as_float(k.rng.y);
  wp.job[10][j] = k.F0.x;
  wp.job[11][j] = k.F0.y;
  wp.job[12][j] = k.F0.z;
  wp.job[13][j] = k.alpha;
  wp.job[14][j] = k.ior;
  wp.job[15][j] = as_float(k.entry);
}

__device__ __forceinline__ void pool_load(const WalkPool& wp, uint32_t j, PooledWalk& k) {
  k.w.lightDir = f3{wp.job[0][j], wp.job[1][j], wp.job[2][j]};
  k.w.energy = f3{wp.job[3][j], wp.job[4][j], wp.job[5][j]};
  k.w.height = wp.job[6][j];
  const uint32_t packed = as_uint(wp.job[7][j]);
  k.w.order = (int)(packed & 0xFFu);
  k.w.outside = (packed & 0x100u) != 0u;
  k.kind = (packed >> 9) & 7u;
  k.surfaceOutside = (packed & kPoolSurfaceOutside) != 0u;
  k.rng = Rng{as_uint(wp.job[8][j]), as_uint(wp.job[9][j])};
  k.F0 = f3{wp.job[10][j], wp.job[11][j], wp.job[12][j]};
  k.alpha = wp.job[13][j];
  k.ior = wp.job[14][j];
  k.entry = as_uint(wp.job[15][j]);
}

// Block-wide: the threads with `flag` put `k` on the pool, which holds `count` walks; returns the new count (the
// same in every thread).  Synchronises the block before (readers of the pool are done) and after.
__device__ __forceinline__ uint32_t pool_push(WalkPool& wp, uint32_t count, bool flag, const PooledWalk& k) {
  const unsigned long long b = __ballot(flag);
  const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
  const uint32_t wave = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63u) == 0u) wp.waveCount[wave] = (uint32_t)__popcll(b);
  __syncthreads();
  uint32_t before = 0, total = 0;
  for (uint32_t w = 0; w < kBlock / 64; ++w) {
    const uint32_t n = wp.waveCount[w];
    if (w < wave) before += n;
    total += n;
  }
  if (flag) pool_store(wp, count + before + below, k);
  __syncthreads();
  return count + total;
}

// The start of a hit's walk, run by the lane that shades the hit.  Returns true if the walk is over (`weight`,
// `lightDir` hold the sample); otherwise `k` is the walk about to scatter (all of it but `entry`).
__device__ __forceinline__ bool walk_head(Rng& rng, const Surface& sf, PooledWalk& k, f3& weight, f3& lightDir) {
  k.kind = sf.interaction;
  k.F0 = sf.baseLinear;
  k.alpha = sf.roughness * sf.roughness;
  k.ior = sf.ior;
  k.surfaceOutside = sf.outside;
  const float iorOut = sf.outside ? 1.0f : sf.ior, iorIn = sf.outside ? sf.ior : 1.0f;
  heitz_walk_begin(k.w, sf.viewDir);
  bool done = heitz_walk_height(rng, k.w, k.kind, k.alpha);
  for (int order = 0; order < LPT_WALK_HEAD_ORDERS && !done; ++order) {
    done = heitz_walk_scatter(rng, k.w, k.kind, k.F0, k.alpha, iorOut, iorIn);
    if (!done) done = heitz_walk_height(rng, k.w, k.kind, k.alpha);
  }
  if (done) weight = heitz_walk_result(k.w, k.kind, k.F0, lightDir);
  k.rng = rng;
  return done;
}

// What a set-aside hit leaves in its queue entry, and how it comes back.
__device__ __forceinline__ void store_continuation(const Queue& cont, uint32_t entry, const Surface& sf, const Path& p,
                                                   uint32_t pidWord) {
  cont.o[entry] = make_float4(sf.posW.x, sf.posW.y, sf.posW.z, as_float(pidWord));
  cont.d[entry] = make_float4(sf.ff.x, sf.ff.y, sf.ff.z, 0.0f);
  cont.m[entry] = make_float4(p.mask.x, p.mask.y, p.mask.z, 0.0f);
}

// Advances the walks on top of the pool — whole waves of them, or with `all` also the wave that is not full — by
// LPT_WALK_PASS_STEPS orders.  A walk that ends finishes its path (Russian roulette and the rest of
// path_tracing.comp:306-323) from what store_continuation left in cont[entry].
// A path that goes on is appended to part queue [lo, ...) of `out` through *countOut, one that ends leaves its
// radiance settled.  Called by all threads of the block; `pooled` (the same in every thread) is updated.
template <int STEPS>
__device__ __forceinline__ void pool_pass(WalkPool& pool, uint32_t& pooled, bool all, const Queue& cont, const Queue& out,
                                          uint32_t lo, uint32_t* countOut, uint32_t* lds, float4* __restrict__ samples,
                                          uint32_t bounce, uint32_t maxBounces) {
  const uint32_t m = min(all ? pooled : (pooled & ~63u), (uint32_t)kBlock), base = pooled - m;
  bool still = false, alive = false;
  PooledWalk k;
  Path p;
  uint32_t pidWord = 0;
  if (threadIdx.x < m) {
    pool_load(pool, base + threadIdx.x, k);
    const float iorOut = k.surfaceOutside ? 1.0f : k.ior, iorIn = k.surfaceOutside ? k.ior : 1.0f;
    bool done = false;
    for (int step = 0; step < STEPS && !done; ++step) {
      done = heitz_walk_scatter(k.rng, k.w, k.kind, k.F0, k.alpha, iorOut, iorIn);
      if (!done) done = heitz_walk_height(k.rng, k.w, k.kind, k.alpha);
    }
    if (done) {
      f3 lightDir, sample;
      const f3 weight = heitz_walk_result(k.w, k.kind, k.F0, lightDir);
      const float4 o = cont.o[k.entry], d = cont.d[k.entry], t = cont.m[k.entry];
      Surface sf;
      sf.posW = f3{o.x, o.y, o.z};
      sf.ff = f3{d.x, d.y, d.z};
      shading_frame(sf.ff, sf.u, sf.v);
      pidWord = as_uint(o.w);
      p.mask = f3{t.x, t.y, t.z};
      p.acc = f3{0.0f, 0.0f, 0.0f};  // what this bounce emitted was settled when the walk was set aside
      p.rng = k.rng;
      alive = shade_end(p, sf, weight, lightDir, bounce, maxBounces, sample);
      if (!alive) settle_ended(samples, pidWord);
    } else {
      still = true;
    }
  }
  pooled = pool_push(pool, base, still, k);  // its first barrier comes after every load above
  const uint32_t dst = lo + block_append(alive, countOut, lds);
  if (alive) store_path(out, dst, p, pidWord);
}

// ---- Heitz walks of one group of hits, finished before the group's threads go on -------------------
// For a kernel whose workgroups shade a single group of kBlock hits (k_primary_shade) there is nothing to wait
// for: every lane takes the start of its own walk, walks still going are parked in LDS as jobs, the lowest
// threads of the block take the jobs up, LPT_PRIMARY_PASS_STEPS orders per pass, and the list of unfinished jobs is
// compacted between passes; the lane that owns the hit keeps the rest of its state in registers and picks up the
// result.  (Measured against setting the hits aside as k_shade does, completed in their entries of the next queue:
// 23.2 against 23.9 ms per launch on a rough terrain scene, 4.47 against 4.57 on the Cornell box.)
#ifndef LPT_PRIMARY_PASS_STEPS
#define LPT_PRIMARY_PASS_STEPS 2
#endif
struct WalkShared {
  float job[kWalkJobWords][kBlock];  // field-major: lanes of a wave touch consecutive words
  uint32_t list[2][kBlock];
  uint32_t waveCount[kBlock / 64];
};

// Block-wide compaction: the threads with `flag` write `value` to consecutive entries of `out`; returns
// how many there are (the same number in every thread).  Synchronises the block.
__device__ __forceinline__ uint32_t walk_compact(bool flag, uint32_t value, uint32_t* out, uint32_t* waveCount) {
  const unsigned long long b = __ballot(flag);
  const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
  const uint32_t wave = threadIdx.x >> 6;
  __syncthreads();  // readers of the previous list and counts are done
  if ((threadIdx.x & 63u) == 0u) waveCount[wave] = (uint32_t)__popcll(b);
  __syncthreads();
  uint32_t before = 0, total = 0;
  for (uint32_t w = 0; w < kBlock / 64; ++w) {
    const uint32_t k = waveCount[w];
    if (w < wave) before += k;
    total += k;
  }
  if (flag) out[before + below] = value;
  __syncthreads();
  return total;
}

__device__ __forceinline__ void walk_store(WalkShared& ws, uint32_t j, const HeitzWalk& w, const Rng& rng) {
  ws.job[0][j] = w.lightDir.x;
  ws.job[1][j] = w.lightDir.y;
  ws.job[2][j] = w.lightDir.z;
  ws.job[3][j] = w.energy.x;
  ws.job[4][j] = w.energy.y;
  ws.job[5][j] = w.energy.z;
  ws.job[6][j] = w.height;
  ws.job[8][j] = as_float(rng.x);
  ws.job[9][j] = as_float(rng.y);
}

// Samples the Heitz BSDF for every thread with `walking` (all threads of the block must call).
__device__ __forceinline__ void walk_block(WalkShared& ws, bool walking, Rng& rng, const Surface& sf, f3& weight,
                                           f3& lightDir) {
  const uint32_t t = threadIdx.x;
  const float alpha = sf.roughness * sf.roughness;
  const float iorOut = sf.outside ? 1.0f : sf.ior;
  const float iorIn = sf.outside ? sf.ior : 1.0f;
  HeitzWalk w;
  heitz_walk_begin(w, sf.viewDir);
  bool parked = false;
  if (walking) {
    const uint32_t kind = sf.interaction;
    bool done = heitz_walk_height(rng, w, kind, alpha);
    if (!done) done = heitz_walk_scatter(rng, w, kind, sf.baseLinear, alpha, iorOut, iorIn);
    if (!done) done = heitz_walk_height(rng, w, kind, alpha);
    if (done) {
      weight = heitz_walk_result(w, kind, sf.baseLinear, lightDir);
    } else {  // about to scatter again
      parked = true;
      walk_store(ws, t, w, rng);
      ws.job[7][t] = as_float((uint32_t)w.order | (w.outside ? 0x100u : 0u) | (kind << 9));
      ws.job[10][t] = sf.baseLinear.x;
      ws.job[11][t] = sf.baseLinear.y;
      ws.job[12][t] = sf.baseLinear.z;
      ws.job[13][t] = alpha;
      ws.job[14][t] = iorOut;
      ws.job[15][t] = iorIn;
    }
  }
  int cur = 0;
  uint32_t n = walk_compact(parked, t, ws.list[0], ws.waveCount);
  while (n > 0u) {  // n is the same in every thread
    bool still = false;
    uint32_t j = 0;
    if (t < n) {
      j = ws.list[cur][t];
      HeitzWalk jw;
      jw.lightDir = f3{ws.job[0][j], ws.job[1][j], ws.job[2][j]};
      jw.energy = f3{ws.job[3][j], ws.job[4][j], ws.job[5][j]};
      jw.height = ws.job[6][j];
      const uint32_t packed = as_uint(ws.job[7][j]);
      jw.order = (int)(packed & 0xFFu);
      jw.outside = (packed & 0x100u) != 0u;
      const uint32_t kind = packed >> 9;
      Rng jr{as_uint(ws.job[8][j]), as_uint(ws.job[9][j])};
      const f3 F0{ws.job[10][j], ws.job[11][j], ws.job[12][j]};
      const float ja = ws.job[13][j], jo = ws.job[14][j], ji = ws.job[15][j];
      bool done = false;
      for (int k = 0; k < LPT_PRIMARY_PASS_STEPS && !done; ++k) {
        done = heitz_walk_scatter(jr, jw, kind, F0, ja, jo, ji);
        if (!done) done = heitz_walk_height(jr, jw, kind, ja);
      }
      if (done) {  // result in place of the walk: weight over the energy, direction over the direction
        f3 ld;
        const f3 wt = heitz_walk_result(jw, kind, F0, ld);
        jw.lightDir = ld;
        jw.energy = wt;
      } else {
        still = true;
        ws.job[7][j] = as_float((uint32_t)jw.order | (jw.outside ? 0x100u : 0u) | (kind << 9));
      }
      walk_store(ws, j, jw, jr);
    }
    n = walk_compact(still, j, ws.list[cur ^ 1], ws.waveCount);
    cur ^= 1;
  }
  if (parked) {
    lightDir = f3{ws.job[0][t], ws.job[1][t], ws.job[2][t]};
    weight = f3{ws.job[3][t], ws.job[4][t], ws.job[5][t]};
    rng = Rng{as_uint(ws.job[8][t]), as_uint(ws.job[9][t])};
  }
}

// The lobe a hit on this object will most likely sample (0 diffuse, 1 conductor, 2 dielectric):
// the largest of the weights of heitz/interaction_type.glsl:12-14.
__device__ __forceinline__ uint32_t likely_lobe(const ObjShade& os) {
  const float wm = os.metallic, wt = (1.0f - os.metallic) * os.transmission;
  const float wd = (1.0f - os.transmission) * (1.0f - os.metallic);
  return (wm >= wt && wm >= wd) ? 1u : (wt >= wd ? 2u : 0u);
}

constexpr int kShadeClasses = 3;

// One bounce of shading for the whole queue.  Rays that missed are finished while the queue is
// scanned; the indices of the hits collect in LDS, one list per likely lobe, and a list is
// shaded when it holds a full block, so the expensive part runs with full waves whose lanes
// mostly take the same branches.  The order in which paths are shaded does not matter: every
// path owns its random-number state.  Paths that go on are appended to `out`.
// Five waves per SIMD (96 VGPRs): the kernel waits on memory two thirds of its wave cycles, so one wave more
// than the compiler's own choice (4) pays on rough materials, where the walks are long (measured: 3 waves
// -8 % Cornell / -14 % on a 70-object terrain scene, 4 as compiled, 5 +0 % / +2 %, 6 -3 % / +3 %).
#ifndef LPT_SHADE_WAVES_PER_SIMD
#define LPT_SHADE_WAVES_PER_SIMD 5
#endif
// The scan sorts hits by the likely lobe of the object hit: for the first LPT_SHADE_LOBE_TABLE objects that comes
// out of a table in LDS (one byte each) instead of a gather into the object records that the scan would have to
// wait for (measured: -0.3 % step time on an 11-object and on a 70-object scene; 0 switches the table off).
#ifndef LPT_SHADE_LOBE_TABLE
#define LPT_SHADE_LOBE_TABLE 1024
#endif
__attribute__((amdgpu_waves_per_eu(LPT_SHADE_WAVES_PER_SIMD, LPT_SHADE_WAVES_PER_SIMD)))
__global__ __launch_bounds__(kBlock) void k_shade(DScene sc, Queue in, const float4* __restrict__ hits,
                                                  QueueFill fillIn, Queue out, uint32_t* fillOut,
                                                  float4* __restrict__ samples, uint32_t bounce, uint32_t maxBounces,
                                                  int bsdfMode) {
  __shared__ uint32_t lds[kBlock / 64 + 1];
  __shared__ uint32_t waveCount[kShadeClasses][kBlock / 64];
  __shared__ uint32_t pending[kShadeClasses][2 * kBlock];  // fewer than kBlock entries each between scans
  __shared__ WalkPool pool;
#if LPT_SHADE_LOBE_TABLE > 0
  // the likely lobe of the first LPT_SHADE_LOBE_TABLE objects, so that the scan below sorts a hit without a
  // dependent gather into the object records
  __shared__ uint8_t lobeOf[LPT_SHADE_LOBE_TABLE];
  for (uint32_t k = threadIdx.x; k < min(sc.nObjects, (uint32_t)LPT_SHADE_LOBE_TABLE); k += kBlock)
    lobeOf[k] = (uint8_t)likely_lobe(sc.oshade[k]);
  __syncthreads();
#endif
  // This workgroup's part of the queue (its own number modulo the parts; the grid is a multiple of them): it
  // shades that part's entries together with the other workgroups of the same part, and what goes on is
  // appended to the same part of `out`, which therefore never holds more than the input part did.
  const uint32_t part = blockIdx.x % kCursorRegions, rank = blockIdx.x / kCursorRegions;
  const uint32_t blocksOfPart = gridDim.x / kCursorRegions;
  const uint32_t lo = part * fillIn.capPart, n = lo + part_fill(fillIn, part);  // entries [lo, n)
  uint32_t* countOut = fillOut + part * kCursorStride;
  const uint32_t wave = threadIdx.x >> 6;
  uint32_t nPending[kShadeClasses] = {0u, 0u, 0u};  // same values in every thread of the block
  uint32_t pooled = 0;                              // Heitz walks waiting in the pool (same in every thread)

  // Shades the first min(count, kBlock) entries of list c (rays that hit something): the hit, the BSDF sample,
  // Russian roulette; called by all threads.  A Heitz walk that is still going after its first step(s) goes to the
  // pool, and its path is finished by pool_pass.
  auto shade_list = [&](int c, uint32_t count) {
    bool began = false, alive = false, parked = false;
    Path p;
    Surface sf;
    PooledWalk k;
    uint32_t pidWord = 0;
    f3 sample{0.0f, 0.0f, 0.0f}, weight{0.0f, 0.0f, 0.0f}, lightDir{0.0f, 0.0f, 0.0f};
    if (threadIdx.x < count) {
      const uint32_t i = pending[c][threadIdx.x];
      const float4 o = in.o[i], d = in.d[i], m = in.m[i], h = hits[i];
      p.o = f3{o.x, o.y, o.z};
      p.d = f3{d.x, d.y, d.z};
      p.mask = f3{m.x, m.y, m.z};
      p.acc = f3{0.0f, 0.0f, 0.0f};  // shade_begin leaves this bounce's emitted radiance here (0 + mask*emission)
      p.rng = Rng{as_uint(d.w), as_uint(m.w)};
      pidWord = as_uint(o.w);
      k.entry = i;
      began = shade_begin(sc, p, Hit{h.x, as_uint(h.y), as_uint(h.z)}, sf, sample);
    }
    // at the last bounce the sampled direction has no use: every path ends there whatever it is
    // (path_tracing.comp:306-323 draws from the path's own random numbers, which end with the path)
    const bool last = bounce + 1u >= maxBounces;
    if (bsdfMode == LPT_BSDF_HEITZ) {
      if (began && !last && !walk_head(p.rng, sf, k, weight, lightDir)) {
        parked = true;  // about to scatter again: the rest of this path's bounce waits for the walk
        settle_emitted(samples, p, pidWord);
        store_continuation(in, k.entry, sf, p, pidWord);
      }
      pooled = pool_push(pool, pooled, parked, k);
    } else if (began && !last) {
      weight = basic_bsdf(p.rng, sf, lightDir);
    }
    if (began && !parked) {
      alive = !last && shade_end(p, sf, weight, lightDir, bounce, maxBounces, sample);
      settle_entry(samples, p, pidWord, alive);
    }
    const uint32_t carry = threadIdx.x + kBlock < count ? pending[c][threadIdx.x + kBlock] : 0u;
    const uint32_t dst = lo + block_append(alive, countOut, lds);  // synchronises the block
    if (alive) store_path(out, dst, p, pidWord);
    if (threadIdx.x + kBlock < count) pending[c][threadIdx.x] = carry;
  };

  // One loop with one place each for the three things the workgroup does, most urgent first: advance the pool
  // when enough walks wait (and, once everything else is done, until it is empty); shade a list that holds a full
  // group (at the end: what is left of it); look at the next kBlock entries of the part.
  uint32_t base = lo + rank * kBlock;
  for (;;) {
    const bool scanned = base >= n;
    int c = -1;
    for (int k = kShadeClasses - 1; k >= 0; --k)
      if (nPending[k] >= (scanned ? 1u : (uint32_t)kBlock)) c = k;
    if (pooled >= kPoolRun || (scanned && c < 0 && pooled > 0u)) {
      pool_pass<LPT_WALK_PASS_STEPS>(pool, pooled, scanned && c < 0 && pooled < 64u, in, out, lo, countOut, lds, samples, bounce, maxBounces);
      continue;
    }
    if (c >= 0) {
      const uint32_t count = c == 0 ? nPending[0] : (c == 1 ? nPending[1] : nPending[2]);
      shade_list(c, count);
      const uint32_t took = min(count, (uint32_t)kBlock);
      for (int k = 0; k < kShadeClasses; ++k) nPending[k] -= k == c ? took : 0u;
      continue;
    }
    if (scanned) break;
    const uint32_t i = base + threadIdx.x;
    base += blocksOfPart * kBlock;
    uint32_t cls = kShadeClasses;  // none
    if (i < n) {
      const float4 h = hits[i];
      if (h.x == kInfinity) {  // path_tracing.comp:220-223 — assignment: radiance gathered so far is dropped
        const float4 m = in.m[i];
        const f3 s = f3{m.x, m.y, m.z} * 0.2f;
        samples[as_uint(in.o[i].w) & ~kAccStored] = make_float4(s.x, s.y, s.z, 0.0f);
      } else {
#if LPT_SHADE_LOBE_TABLE > 0
        const uint32_t obj = as_uint(h.y);
        cls = obj < (uint32_t)LPT_SHADE_LOBE_TABLE ? (uint32_t)lobeOf[obj] : likely_lobe(sc.oshade[obj]);
#else
        cls = likely_lobe(sc.oshade[as_uint(h.y)]);
#endif
      }
    }
    // ranks within the block, per class
    uint32_t below[kShadeClasses];
    for (int k = 0; k < kShadeClasses; ++k) {
      const unsigned long long b = __ballot(cls == (uint32_t)k);
      below[k] = __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
      if ((threadIdx.x & 63u) == 0u) waveCount[k][wave] = (uint32_t)__popcll(b);
    }
    __syncthreads();
    for (int k = 0; k < kShadeClasses; ++k) {
      uint32_t before = 0, total = 0;
      for (uint32_t w = 0; w < kBlock / 64; ++w) {
        const uint32_t t = waveCount[k][w];
        if (w < wave) before += t;
        total += t;
      }
      if (cls == (uint32_t)k) pending[k][nPending[k] + before + below[k]] = i;
      nPending[k] += total;
    }
    __syncthreads();  // the lists are complete; waveCount may be rewritten by the next scan
  }
}

// The first segment of every path of a batch in one kernel: camera ray, packet walk (as k_primary) and
// the shading of what it hit (as k_shade does for bounce 0, without the detour through a ray queue and a
// hit array).  The pixels of a block mostly hit the same object, so the lanes shade in step without the
// sorting k_shade does.  Paths that go on are appended to `out`.
// (Letting a workgroup take several groups of paths in turn, so that Heitz walks can wait across groups as in
// k_shade, was built and measured: the loop around the packet walk costs registers — 77 spilled at 6 waves per
// SIMD — and the kernel is mostly the packet walk: -1.7 ms of 23 on a rough terrain scene, +0.2 ms of 4.5 on the
// Cornell box.  Not kept.)
// Six waves per SIMD (80 VGPRs): the packet walk is a chain of dependent scalar loads and wants the waves
// (measured: 4 waves -1.2 %, 5 as the compiler's own choice, 6 +0.7 % Cornell / +1.4 % rough terrain, 8 -0.5 %).
#ifndef LPT_PRIMARY_WAVES_PER_SIMD
#define LPT_PRIMARY_WAVES_PER_SIMD 6
#endif
__attribute__((amdgpu_waves_per_eu(LPT_PRIMARY_WAVES_PER_SIMD, LPT_PRIMARY_WAVES_PER_SIMD)))
__global__ __launch_bounds__(kBlock) void k_primary_shade(const TNode* __restrict__ nodes, const Tri* __restrict__ tris,
                                                          const ObjTrav* __restrict__ otrav, DScene sc, Partition P,
                                                          const lpt_ubo* __restrict__ ubos, uint32_t nFrames, Queue out,
                                                          uint32_t* fillOut, uint32_t capPart, float4* __restrict__ samples,
                                                          uint32_t maxBounces, int bsdfMode, uint32_t packetFrames) {
  __shared__ uint32_t lds[kBlock / 64 + 1];
  __shared__ WalkShared walk;
  Path p;
  Hit hit;
  bool valid;
  uint32_t pid;
  primary_packet<false>(nodes, tris, otrav, sc.rootRef, P, ubos, nFrames, packetFrames, nullptr, 0u,
                        PrimaryQueue{nullptr, nullptr, nullptr}, nullptr, blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6), p, hit,
                        valid, pid);
  // with a bounce limit of one every path ends at its first hit whatever direction is sampled there
  const bool last = maxBounces <= 1u;
  bool alive = false, began = false;
  uint32_t pidWord = pid;
  Surface sf;
  f3 s{0.0f, 0.0f, 0.0f}, weight{0.0f, 0.0f, 0.0f}, lightDir{0.0f, 0.0f, 0.0f};
  if (valid) began = shade_begin(sc, p, hit, sf, s);  // a miss ends the path with s = 0.2 * throughput
  if (bsdfMode == LPT_BSDF_HEITZ) {
    walk_block(walk, began && !last, p.rng, sf, weight, lightDir);
  } else if (began && !last) {
    weight = basic_bsdf(p.rng, sf, lightDir);
  }
  if (began && !last) alive = shade_end(p, sf, weight, lightDir, 0u, maxBounces, s);
  if (valid) {
    if (hit.t == kInfinity) {
      samples[pid] = make_float4(s.x, s.y, s.z, 0.0f);
    } else if (p.acc.x != 0.0f || p.acc.y != 0.0f || p.acc.z != 0.0f) {
      samples[pid] = make_float4(p.acc.x, p.acc.y, p.acc.z, 0.0f);  // emitted by the first hit
      pidWord |= kAccStored;
    } else if (!alive) {
      samples[pid] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
  }
  // appended to the part queue of this workgroup's number: at most one workgroup's paths per kCursorRegions
  // workgroups land there, which is what the parts are sized for
  const uint32_t part = blockIdx.x % kCursorRegions;
  const uint32_t dst = part * capPart + block_append(alive, fillOut + part * kCursorStride, lds);
  if (alive) store_path(out, dst, p, pidWord);
}

// Adds the finished frames to the accumulation image in frame order
// (shaders/path_tracing.comp:347-351: reset stores (rgb,1), otherwise adds (rgb,1)).
__global__ __launch_bounds__(kBlock) void k_accumulate(Partition P, const lpt_ubo* __restrict__ ubos, uint32_t nFrames,
                                                       const float4* __restrict__ samples, float4* __restrict__ accum) {
  for (uint32_t slot = blockIdx.x * kBlock + threadIdx.x; slot < P.nSlots; slot += gridDim.x * kBlock) {
    int x, y;
    if (!slot_to_pixel(P, slot, x, y)) continue;
    float4 a = accum[slot];
    for (uint32_t f = 0; f < nFrames; ++f) {
      float4 s = samples[(size_t)f * P.nSlots + slot];
      if (ubos[f].reset) {
        a = make_float4(s.x, s.y, s.z, 1.0f);
      } else {
        a = make_float4(a.x + s.x, a.y + s.y, a.z + s.z, a.w + 1.0f);
      }
    }
    accum[slot] = a;
  }
}

// totals[kTotPaths] += paths started, totals[kTotSegments] += ray segments traced (queue fills per bounce), totals[kTotTimed] += those
// of the bounces from `timedFrom` on (the ones whose extend launches are timed).  `fills`: one set of fill
// counts per bounce.
// `report` (may be null): words in host memory that receive the batch's queue fills, for the host to choose the
// next batch's hand-over bounce by: [b] = fill of bounce b for b <= min(nBounces, kReportBounces - 1) (the queue of
// bounce nBounces is the one handed to the persistent launch, if any), [kReportBounces] = nBounces,
// [kReportBounces + 1] = paths started, [kReportBounces + 2] = `generation` (non-zero).  The host reads the
// report without waiting for the stream, so it is written like a sequence lock: the generation word is cleared
// first and set last, with system-wide fences in between, and the host takes a report only if it reads the
// generation it expects both before and after the other words.
constexpr uint32_t kReportBounces = 64, kReportGeneration = kReportBounces + 2, kReportWords = kReportBounces + 3;
__global__ void k_tally(const uint32_t* __restrict__ fills, uint32_t nBounces, uint32_t timedFrom,
                        unsigned long long* totals, uint32_t haveNext, uint32_t* report, uint32_t generation) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    unsigned long long seg = 0, timed = 0, first = 0;
    if (report) {
      __hip_atomic_store(&report[kReportGeneration], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __threadfence_system();
    }
    for (uint32_t b = 0; b < nBounces + haveNext; ++b) {
      unsigned long long n = 0;
      for (uint32_t r = 0; r < kCursorRegions; ++r) n += fills[b * kCursorWords + r * kCursorStride];
      if (report && b < kReportBounces) report[b] = (uint32_t)n;
      if (b >= nBounces) break;  // the queue handed on: reported, but its segments are counted by the launch that takes it
      seg += n;
      if (b >= timedFrom) timed += n;
      if (b == 0) first = n;
    }
    if (report) {
      report[kReportBounces + 1] = (uint32_t)first;
      report[kReportBounces] = nBounces;
      __threadfence_system();
      __hip_atomic_store(&report[kReportGeneration], generation, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    totals[kTotPaths] += first;
    totals[kTotSegments] += seg;
    totals[kTotTimed] += timed;
  }
}

// ---- persistent path tracer ----------------------------------------------------------

// Traces every path of a batch of frames from ray generation to its radiance in `samples`
// (lpt_trace.h).  `cursor` = one zeroed set of cursors (kCursorWords words).
#ifndef LPT_TRACE_WAVES_PER_SIMD
#define LPT_TRACE_WAVES_PER_SIMD 4
#endif
template <int S, bool QUEUED>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(LPT_TRACE_WAVES_PER_SIMD, LPT_TRACE_WAVES_PER_SIMD)))
void k_trace(DScene sc, Partition P, const lpt_ubo* __restrict__ ubos,
                                                  uint32_t nFrames, uint32_t maxBounces, int bsdfMode,
                                                  float4* __restrict__ samples, uint32_t* cursor,
                                                  unsigned long long* totals, TraceQueue src, uint32_t* flags,
                                                  uint32_t roundLimit) {
  __shared__ TraceShared<S, kBlock> sh;
  trace_persistent<S, kBlock, QUEUED>(sc, P, ubos, nFrames, maxBounces, bsdfMode, samples, cursor, totals, sh, src, flags,
                                      roundLimit);
}

// ---- megakernel --------------------------------------------------------------------

template <int S>
__global__ __launch_bounds__(kBlock) void k_megakernel(DScene sc, Partition P, lpt_ubo ubo, uint32_t maxBounces,
                                                       int bsdfMode, float4* __restrict__ accum,
                                                       unsigned long long* totals) {
  __shared__ uint32_t stack[S * kBlock];
  uint32_t slot = blockIdx.x * kBlock + threadIdx.x;
  uint32_t segments = 0, started = 0;
  int x, y;
  if (slot < P.nSlots && slot_to_pixel(P, slot, x, y)) {
    started = 1;
    Path p;
    generate_ray(ubo, x, y, (float)P.W, (float)P.H, p);
    f3 s{0.0f, 0.0f, 0.0f};
    for (uint32_t bounce = 0; bounce < maxBounces; ++bounce) {
      ++segments;
      Hit h = intersect_scene(sc, p.o, p.d, &stack[threadIdx.x], kBlock);
      if (!shade_hit<false>(sc, p, h, bounce, maxBounces, bsdfMode, s)) break;
    }
    if (ubo.reset) {
      accum[slot] = make_float4(s.x, s.y, s.z, 1.0f);
    } else {
      float4 a = accum[slot];
      accum[slot] = make_float4(a.x + s.x, a.y + s.y, a.z + s.z, a.w + 1.0f);
    }
  }
  // per-wave reduction of the counters, one atomic pair per wave
  for (int off = 32; off > 0; off >>= 1) {
    segments += (uint32_t)__shfl_down((int)segments, off);
    started += (uint32_t)__shfl_down((int)started, off);
  }
  if ((threadIdx.x & 63) == 0 && started) {
    atomicAdd(&totals[kTotPaths], (unsigned long long)started);
    atomicAdd(&totals[kTotSegments], (unsigned long long)segments);
  }
}

// ---- image kernels -------------------------------------------------------------------

// Compact owned-pixel buffer -> scan-line image (pixels of other ranks become 0).
__global__ void k_unpermute(Partition P, const float4* __restrict__ accum, float4* __restrict__ image) {
  int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= P.W || y >= P.H) return;
  int owner;
  uint32_t slot;
  pixel_to_slot(P, x, y, owner, slot);
  image[(size_t)y * P.W + x] = owner == P.rank ? accum[slot] : make_float4(0.f, 0.f, 0.f, 0.f);
}

// Concatenated compact buffers of all ranks -> scan-line image.
__global__ void k_assemble(Partition P, const float4* __restrict__ gathered, float4* __restrict__ image) {
  int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= P.W || y >= P.H) return;
  int owner;
  uint32_t slot;
  pixel_to_slot(P, x, y, owner, slot);
  image[(size_t)y * P.W + x] = gathered[(size_t)owner * P.nSlots + slot];
}

// shaders/tex_to_quad.frag:24-33: display row y shows accumulation row H-1-y.
__global__ void k_resolve(Partition P, const float4* __restrict__ accum, float invSampleCount,
                          float4* __restrict__ image) {
  int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= P.W || y >= P.H) return;
  int owner;
  uint32_t slot;
  pixel_to_slot(P, x, P.H - 1 - y, owner, slot);
  float4 out = make_float4(0.f, 0.f, 0.f, 1.f);
  if (owner == P.rank) {
    float4 a = accum[slot];
    const float exposure = 1.5f, invGamma = 1.0f / 2.2f;
    out.x = pow_fixed(1.0f - exp_fixed(-(a.x * invSampleCount) * exposure), invGamma);
    out.y = pow_fixed(1.0f - exp_fixed(-(a.y * invSampleCount) * exposure), invGamma);
    out.z = pow_fixed(1.0f - exp_fixed(-(a.z * invSampleCount) * exposure), invGamma);
  }
  image[(size_t)y * P.W + x] = out;
}

// baseLinear = SRGBToLinear(baseColorFactor.rgb) with the device's own pow, so host code
// never has to reproduce it.
__global__ void k_finish_objects(ObjShade* os, const float* __restrict__ baseColor, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  os[i].baseLinear[0] = srgb_to_linear(baseColor[4 * i + 0]);
  os[i].baseLinear[1] = srgb_to_linear(baseColor[4 * i + 1]);
  os[i].baseLinear[2] = srgb_to_linear(baseColor[4 * i + 2]);
}

// ---- test hooks ----------------------------------------------------------------------

__global__ void k_debug_math(int op, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                             uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float x = a[i], y = b ? b[i] : 0.0f, r = 0.0f, s, c;
  switch (op) {
    case 0: sincos_fixed(x, s, c); r = s; break;
    case 1: sincos_fixed(x, s, c); r = c; break;
    case 2: r = tan_fixed(x); break;
    case 3: r = log_fixed(x); break;
    case 4: r = exp_fixed(x); break;
    case 5: r = pow_fixed(x, y); break;
    case 6: r = x / y; break;
    case 7: r = __builtin_sqrtf(x); break;
    case 8: r = srgb_to_linear(x); break;
    default: break;
  }
  out[i] = r;
}

template <int S>
__global__ __launch_bounds__(kBlock) void k_debug_intersect(DScene sc, const float* __restrict__ o,
                                                            const float* __restrict__ d, uint32_t n,
                                                            float* __restrict__ t, uint32_t* __restrict__ obj,
                                                            uint32_t* __restrict__ prim) {
  __shared__ uint32_t stack[S * kBlock];
  uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  Hit h = intersect_scene(sc, f3{o[3 * i], o[3 * i + 1], o[3 * i + 2]}, f3{d[3 * i], d[3 * i + 1], d[3 * i + 2]},
                          &stack[threadIdx.x], kBlock);
  t[i] = h.t;
  obj[i] = h.obj;
  prim[i] = h.prim;
}

// ---- scratch of the entry point that allocates nothing (lpt_render_frame_static) ---------------
// One array per GPU, held from the moment the library is loaded on it until the process ends.  Layout:
// [0, 16 KiB) the ray cursors of the persistent launch and the frame's uniform block, then one RGBA32F
// sample per owned pixel slot.  The name below is a host-side stand-in: host code uses the device address
// lpt_create obtains with hipGetSymbolAddress (lpt_ctx::dStatic), never the name.
constexpr uint32_t kStaticMaxSlots = 2304u * 1024u;  // 2 359 296 pixel slots: up to 1920x1080 (36 MiB of samples)
constexpr size_t kStaticHeaderBytes = 16384;
constexpr size_t kStaticUboOffset = kCursorWords * sizeof(uint32_t);
static_assert(kStaticUboOffset + sizeof(lpt_ubo) <= kStaticHeaderBytes, "cursors and uniform block fit the header");
constexpr size_t kStaticBytes = kStaticHeaderBytes + (size_t)kStaticMaxSlots * sizeof(float4);
static __device__ __align__(256) unsigned char scratch[kStaticBytes];

__global__ void k_store_ubo(lpt_ubo value, lpt_ubo* dst) {
  if (blockIdx.x == 0 && threadIdx.x == 0) *dst = value;
}

// Every context on a GPU shares that GPU's scratch array, and every context has a stream of its own: the calls
// that use the array are put in order across contexts and streams.  Per GPU: an event recorded behind the last
// call's work and who recorded it; a call from another context or stream makes its stream wait for that event
// first (hipStreamWaitEvent: no allocation).  The events are created by lpt_create and live as long as the process.
constexpr int kMaxStaticDevices = 64;
struct StaticUse {
  hipEvent_t done = nullptr;  // recorded after the last lpt_render_frame_static on this GPU
  const void* ctx = nullptr;  // who recorded it
  hipStream_t stream = nullptr;
  bool recorded = false;
};
StaticUse g_staticUse[kMaxStaticDevices];
std::mutex g_staticMutex;

thread_local std::string g_create_error;

}  // namespace

// ---------------------------------------------------------------------------------------
// Context
// ---------------------------------------------------------------------------------------

struct lpt_ctx {
  int device = 0;
  hipStream_t ownStream = nullptr;
  hipStream_t stream = nullptr;
  int numCUs = 256;
  std::string err;

  // scene (device)
  TNode* dNodes = nullptr;  // object tree + mesh trees, hottest first
  Tri* dTris = nullptr;
  ObjTrav* dObjTrav = nullptr;
  ObjShade* dObjShade = nullptr;
  lpt_vertex* dVerts = nullptr;
  DTexture* dTextures = nullptr;
  uint8_t* dTexPixels = nullptr;  // all texture images, 16-byte aligned each
  uint32_t nTextures = 0;
  uint32_t rootRef = kRefSentinel;
  uint32_t nNodes = 0, nTris = 0, nObjects = 0;
  int stackNeed = 0;
  bool sceneLoaded = false;

  // image
  Partition part{};
  bool sized = false;
  float4* dAccum = nullptr;
  float4* dImage = nullptr;  // scan-line staging for read-back / resolve
  uint32_t* dValidBefore = nullptr;  // per slot: pixels of the image among the lower slots (null: all slots are)
  uint32_t nValid = 0;               // slots that are pixels of the image

  // options
  int maxBounces = 10;  // MAX_TRACE_DEPTH, shaders/path_tracing.comp:20
  int bsdfMode = LPT_BSDF_HEITZ;
  int pipeline = LPT_PIPELINE_AUTO;
  int activePipeline = LPT_PIPELINE_WAVEFRONT;  // what AUTO resolved to for the current call
  int framesInFlight = 0;

  // wavefront pool
  size_t poolPaths = 0;
  uint32_t capPart = 0;  // entries per part queue: the queue arrays hold kCursorRegions * capPart entries
  int poolArrays = 0;  // float4 arrays per path in dPool: 8 (wavefront) or 1 (persistent: samples only)
  int poolBounces = 0;
  float4* dPool = nullptr;  // 2 queues x 3 arrays + hits + samples
  Queue queues[2]{};
  float4* dHits = nullptr;
  float4* dSamples = nullptr;
  uint32_t* dCounts = nullptr;   // [poolBounces] sets of queue fill counts, then [poolBounces] sets of ray cursors
  lpt_ubo* dUbos = nullptr;
  // Pinned staging of the frames' uniform blocks (+ one word behind them), two buffers used in turn: a call may
  // be enqueued while the previous one still runs; evStaging[k] says when the copy out of buffer k is done.
  lpt_ubo* hUbos[2] = {nullptr, nullptr};
  hipEvent_t evStaging[2] = {nullptr, nullptr};
  bool stagingBusy[2] = {false, false};
  int stagingNext = 0;
  int ubosCap = 0;
  unsigned long long* dTotals = nullptr;  // kTotalsWords words: paths, segments, segments of timed launches (lpt_trace.h)
  // Words the kernels raise when they give up: pinned host memory the device writes through its mapped
  // address, so the host reads them after a stream synchronisation without a copy.
  // [0]: a persistent launch reached its round limit with paths still in flight.
  // [4 ..]: the queue fills the last finished wavefront batch reported (k_tally), read without waiting.
  volatile uint32_t* hFlags = nullptr;
  uint32_t* dFlags = nullptr;
  uint32_t traceRoundLimit = LPT_TRACE_ROUND_LIMIT;
  // Reports of queue fills are only taken from batches of the current scene, image size and options: every
  // change of those starts a new generation, and a report carries the generation of the batch that wrote it.
  uint32_t reportGeneration = 1;
  // Device time per Mi paths of the last finished lpt_render_frames call of this generation (0: none yet): what
  // LPT_PIPELINE_AUTO predicts a batch's time by.
  double msPerMiPath = 0.0;
  int autoChoice = -1;  // what LPT_PIPELINE_AUTO chose last in this generation (-1: nothing yet)
  uint32_t timedGeneration = 0;  // the generation the call recorded in evStart / evStop belonged to (0: not to be used)
  bool callAllocated = false;    // the current lpt_render_frames call had to allocate device or pinned memory
  unsigned char* dStatic = nullptr;  // this GPU's copy of `scratch` (device address)
  uint32_t* dDeepIndex = nullptr;    // guarded k_extend: queue indices of the rays left to k_extend_deep
  size_t deepEntries = 0;
  uint32_t* dDeepColumns = nullptr;  // k_extend_deep: a stack in memory per wave of its grid

  // timing of the last lpt_render_frames call
  hipEvent_t evStart = nullptr, evStop = nullptr;
  std::vector<hipEvent_t> evPool;
  size_t evUsed = 0;
  uint64_t lastFrames = 0;
  uint32_t timedLo = 0, timedHi = 0;  // bounces whose segments the timed launches of the last batch traced
  bool haveTiming = false;
};

namespace {

// Queue-fill reports of earlier batches no longer describe what the next batch will do.
void next_report_generation(lpt_ctx* c) {
  if (++c->reportGeneration == 0u) c->reportGeneration = 1u;  // 0 means "being written"
  c->msPerMiPath = 0.0;
  c->autoChoice = -1;
}

int fail(lpt_ctx* ctx, int code, const std::string& msg) {
  if (ctx) ctx->err = msg;
  return code;
}

#define LPT_HIP(ctx, call)                                                                    \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess)                                                                     \
      return fail(ctx, LPT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));       \
  } while (0)

// Every entry point that launches, copies or synchronises makes the context's GPU current first:
// contexts on different GPUs may be used in turn from one thread.
#define LPT_USE_DEVICE(ctx) LPT_HIP(ctx, hipSetDevice((ctx)->device))

// After a stream synchronisation: did a kernel of the finished work give up?  The flag is consumed.
int check_flags(lpt_ctx* c) {
  if (c->hFlags && c->hFlags[0]) {
    c->hFlags[0] = 0u;
    return fail(c, LPT_ERR_TRUNCATED,
                "a persistent launch reached its round limit with paths still in flight: the last render is incomplete");
  }
  return 0;
}

template <typename T>
void dfree(T*& p) {
  if (p) (void)hipFree(p);
  p = nullptr;
}

void free_scene(lpt_ctx* c) {
  dfree(c->dNodes);
  dfree(c->dTris);
  dfree(c->dObjTrav);
  dfree(c->dObjShade);
  dfree(c->dVerts);
  c->sceneLoaded = false;
}

void free_textures(lpt_ctx* c) {
  dfree(c->dTextures);
  dfree(c->dTexPixels);
  c->nTextures = 0;
}

void free_pool(lpt_ctx* c) {
  dfree(c->dPool);
  dfree(c->dCounts);
  dfree(c->dDeepIndex);
  dfree(c->dDeepColumns);
  c->poolPaths = 0;
  c->poolBounces = 0;
  c->deepEntries = 0;
}

DScene device_scene(const lpt_ctx* c) {
  return DScene{c->dNodes,      c->dTris,   c->dObjTrav, c->dObjShade, c->dVerts,
                c->dTextures,   c->rootRef, c->nNodes,   c->nTris,     c->nObjects};
}

// Validates the binary tree rooted at nodes[base] (child indices are offset by `base`) and
// returns the number of stack entries a traversal of it can hold at once; leafNeed(first,count)
// gives what a leaf adds (and validates its range; negative = error code).  Iterative
// post-order; rejects bad indices, cycles and leaves too large for a traversal reference.
// `owner` records for every node the `tag` of the first tree that reached it: the derived arrays
// hold one copy of a node (and of a leaf's triangles), so a node may only be shared by trees with
// the same tag (mesh trees: the object's bvhOffset and verticesOffset packed into one word).
constexpr uint64_t kNoOwner = ~0ull;
template <typename LeafNeed>
int tree_need(const lpt_bvh_node* nodes, uint32_t nNodes, uint32_t base, std::vector<int>& need, LeafNeed leafNeed,
              std::string& err, std::vector<uint64_t>& owner, uint64_t tag) {
  if (base >= nNodes) {
    err = "BVH root index out of range";
    return LPT_ERR_INVALID;
  }
  auto claim = [&](uint32_t i) -> bool {
    if (owner[i] == kNoOwner) owner[i] = tag;
    if (owner[i] == tag) return true;
    err = "mesh BVH nodes are shared by objects that differ in bvhOffset or verticesOffset";
    return false;
  };
  if (!claim(base)) return LPT_ERR_UNSUPPORTED;
  auto leaf = [&](const lpt_bvh_node& n) -> int {
    uint32_t first = n.indices[0], last = n.indices[1];
    uint32_t count = last > first ? last - first : 0u;
    if (count > kMaxLeafCount) {
      err = "BVH leaf holds more than 63 primitives";
      return LPT_ERR_UNSUPPORTED;
    }
    return leafNeed(first, count);
  };
  if (nodes[base].isLeaf) return leaf(nodes[base]);
  // need[] doubles as the visit state: -2 unvisited, -1 on the stack, >=0 done
  std::vector<uint32_t> st;
  st.push_back(base);
  while (!st.empty()) {
    uint32_t i = st.back();
    const lpt_bvh_node& n = nodes[i];
    uint64_t c64[2] = {(uint64_t)base + n.indices[0], (uint64_t)base + n.indices[1]};
    if (c64[0] >= nNodes || c64[1] >= nNodes) {
      err = "BVH child index out of range";
      return LPT_ERR_INVALID;
    }
    uint32_t c[2] = {(uint32_t)c64[0], (uint32_t)c64[1]};
    if (!claim(c[0]) || !claim(c[1])) return LPT_ERR_UNSUPPORTED;
    if (need[i] == -2) {
      need[i] = -1;
      bool pushed = false;
      for (int k = 0; k < 2; ++k) {
        if (!nodes[c[k]].isLeaf) {
          if (need[c[k]] == -1) {
            err = "BVH contains a cycle";
            return LPT_ERR_INVALID;
          }
          if (need[c[k]] == -2) {
            st.push_back(c[k]);
            pushed = true;
          }
        }
      }
      if (pushed) continue;
    }
    int cn[2];
    for (int k = 0; k < 2; ++k) {
      if (nodes[c[k]].isLeaf) {
        cn[k] = leaf(nodes[c[k]]);
        if (cn[k] < 0) return cn[k];
      } else {
        cn[k] = need[c[k]];
      }
    }
    // child 1 is walked first while child 0 may wait on the stack
    need[i] = std::max(1 + cn[1], cn[0]);
    st.pop_back();
  }
  return need[base];
}

// Surface area of the world-space box of an object-space box.
float world_area(const float* mn, const float* mx, const float* m /* column-major 4x4, or null */) {
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int k = 0; k < 8; ++k) {
    float p[3] = {(k & 1) ? mx[0] : mn[0], (k & 2) ? mx[1] : mn[1], (k & 4) ? mx[2] : mn[2]};
    float q[3] = {p[0], p[1], p[2]};
    if (m)
      for (int r = 0; r < 3; ++r) q[r] = m[r] * p[0] + m[4 + r] * p[1] + m[8 + r] * p[2] + m[12 + r];
    for (int r = 0; r < 3; ++r) {
      lo[r] = std::min(lo[r], q[r]);
      hi[r] = std::max(hi[r], q[r]);
    }
  }
  float d[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
  if (!(d[0] >= 0.0f && d[1] >= 0.0f && d[2] >= 0.0f)) return 0.0f;
  return 2.0f * (d[0] * d[1] + d[1] * d[2] + d[2] * d[0]);
}

int ensure_events(lpt_ctx* c, size_t n) {
  while (c->evPool.size() < n) {
    hipEvent_t e;
    LPT_HIP(c, hipEventCreate(&e));
    c->evPool.push_back(e);
  }
  return 0;
}

int auto_frames_in_flight(const lpt_ctx* c) {
  if (c->framesInFlight > 0) return c->framesInFlight;
  // Every traversal launch ends with a latency-bound tail (its longest rays, about half a
  // millisecond whatever the number of rays), so large batches pay off: aim for 256 Mi paths in
  // flight (about 34 GB of queues at 128 B per path, 17 GB for the hybrid organisation's 64 B;
  // 1080p: 128 frames).  Measured on MI355X, Cornell box 1080p, wavefront kernels: 32 frames in
  // flight 7.0, 64: 7.9, 128: 8.4 Gsamples/s.
  const double target = 256.0 * 1024 * 1024;
  int f = (int)(target / std::max<uint32_t>(c->part.nSlots, 1u));
  return std::min(2048, std::max(1, f));
}

// Camera rays are traced as packets unless LPT_PRIMARY_PACKETS=0 (comparison runs).
bool primary_packets() {
  static const bool on = !(std::getenv("LPT_PRIMARY_PACKETS") && std::atoi(std::getenv("LPT_PRIMARY_PACKETS")) == 0);
  return on;
}

// Unless LPT_PRIMARY_FUSED=0 (comparison runs), the packet kernel also shades what the camera rays hit.
bool primary_fused() {
  static const bool on = !(std::getenv("LPT_PRIMARY_FUSED") && std::atoi(std::getenv("LPT_PRIMARY_FUSED")) == 0);
  return on;
}

constexpr size_t kCountSets = 3;  // per bounce: queue fill counts, ray cursors, deep-ray count

int ensure_pool(lpt_ctx* c, int frames) {
  size_t paths = (size_t)frames * c->part.nSlots;
  // float4 arrays of one entry per path: wavefront 2 queues x 3 + hits + samples = 8; hybrid with the fused
  // first segment the queue it writes + samples = 4; persistent samples only
  const bool oneQueue = c->activePipeline == LPT_PIPELINE_HYBRID && primary_packets() && primary_fused();
  int arrays = c->activePipeline == LPT_PIPELINE_PERSISTENT ? 1 : (oneQueue ? 4 : 8);
  // A pool laid out for more arrays serves an organisation that wants fewer (its arrays are among them): a context
  // whose batches alternate between organisations keeps the larger pool instead of allocating anew every time.
  if (paths <= c->poolPaths && arrays < c->poolArrays) arrays = c->poolArrays;
  if (paths > c->poolPaths || arrays != c->poolArrays) {
    c->callAllocated = true;
    dfree(c->dPool);
    c->poolPaths = 0;
    // A queue array holds kCursorRegions part queues.  A part receives the paths of every kCursorRegions-th
    // workgroup of the kernel that starts the paths (256 each), later kernels keep every path in its part:
    // ceil(workgroups / parts) * 256 entries hold whatever happens.  One array of `paths` entries for the samples.
    const size_t blocks = (paths + kBlock - 1) / kBlock;
    const size_t capPart = ((blocks + kCursorRegions - 1) / kCursorRegions) * kBlock;
    const size_t entries = capPart * kCursorRegions;
    if (entries >= (1ull << 32)) return fail(c, LPT_ERR_INVALID, "too many paths in flight for 32-bit queue indices");
    if (hipMalloc(&c->dPool, ((size_t)(arrays - 1) * entries + paths) * sizeof(float4)) != hipSuccess) {
      (void)hipGetLastError();
      c->dPool = nullptr;
      return fail(c, LPT_ERR_NOMEM, "not enough device memory for " + std::to_string(frames) + " frames in flight");
    }
    float4* p = c->dPool;
    c->queues[0] = c->queues[1] = Queue{nullptr, nullptr, nullptr};
    c->dHits = nullptr;
    for (int q = arrays == 8 ? 0 : 1; q < 2 && arrays > 1; ++q) {
      c->queues[q].o = p; p += entries;
      c->queues[q].d = p; p += entries;
      c->queues[q].m = p; p += entries;
    }
    if (arrays == 8) {
      c->dHits = p; p += entries;
    }
    c->dSamples = p;
    c->capPart = (uint32_t)capPart;
    LPT_HIP(c, hipMemsetAsync(c->dSamples, 0, paths * sizeof(float4), c->stream));
    c->poolPaths = paths;
    c->poolArrays = arrays;
  }
  if (c->maxBounces + 1 > c->poolBounces) {
    c->callAllocated = true;
    dfree(c->dCounts);
    c->poolBounces = 0;  // stays 0 if the allocation fails: the next call tries again instead of using a null array
    LPT_HIP(c, hipMalloc(&c->dCounts, kCountSets * kCursorWords * (size_t)(c->maxBounces + 1) * sizeof(uint32_t)));
    c->poolBounces = c->maxBounces + 1;
  }
  if (frames > c->ubosCap) {
    c->callAllocated = true;
    LPT_HIP(c, hipStreamSynchronize(c->stream));  // copies out of the old staging buffers are done
    dfree(c->dUbos);
    c->ubosCap = 0;
    for (int k = 0; k < 2; ++k) {
      if (c->hUbos[k]) (void)hipHostFree(c->hUbos[k]);
      c->hUbos[k] = nullptr;
      c->stagingBusy[k] = false;
    }
    LPT_HIP(c, hipMalloc(&c->dUbos, (size_t)frames * sizeof(lpt_ubo)));
    for (int k = 0; k < 2; ++k)  // the frames' blocks and one set of fill counts behind them
      LPT_HIP(c, hipHostMalloc(&c->hUbos[k], (size_t)frames * sizeof(lpt_ubo) + kCursorWords * sizeof(uint32_t)));
    c->ubosCap = frames;
  }
  return 0;
}

// The uniform blocks of a batch to c->dUbos and, with `firstQueueFill`, the fill counts of the first queue to
// the first set of c->dCounts (that queue is filled densely in path-id order, if it is written at all: its
// parts are full one after the other; the counts also tell k_tally how many paths were started), through the
// pinned staging buffer whose turn it is.  Waits only if that buffer's previous copy (two batches ago) is still
// pending.
int stage_frames(lpt_ctx* c, const lpt_ubo* ubos, int frames, bool firstQueueFill) {
  const int k = c->stagingNext;
  if (c->stagingBusy[k]) LPT_HIP(c, hipEventSynchronize(c->evStaging[k]));
  lpt_ubo* h = c->hUbos[k];
  std::memcpy(h, ubos, (size_t)frames * sizeof(lpt_ubo));
  LPT_HIP(c, hipMemcpyAsync(c->dUbos, h, (size_t)frames * sizeof(lpt_ubo), hipMemcpyHostToDevice, c->stream));
  if (firstQueueFill) {
    uint32_t* fills = reinterpret_cast<uint32_t*>(h + frames);
    std::memset(fills, 0, kCursorWords * sizeof(uint32_t));
    size_t left = (size_t)frames * c->nValid;
    for (uint32_t r = 0; r < kCursorRegions; ++r) {
      const size_t take = std::min<size_t>(left, c->capPart);
      fills[r * kCursorStride] = (uint32_t)take;
      left -= take;
    }
    LPT_HIP(c, hipMemcpyAsync(c->dCounts, fills, kCursorWords * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
  }
  LPT_HIP(c, hipEventRecord(c->evStaging[k], c->stream));
  c->stagingBusy[k] = true;
  c->stagingNext = k ^ 1;
  return 0;
}

int grid_for(const lpt_ctx* c, size_t items, int blocksPerCU) {
  size_t need = (items + kBlock - 1) / kBlock;
  size_t cap = (size_t)c->numCUs * blocksPerCU;
  return (int)std::max<size_t>(1, std::min(need, cap));
}

// `deepCount`: GUARD only — a zeroed word for the number of rays the launch leaves to k_extend_deep.
template <int S, bool GUARD>
int launch_extend(lpt_ctx* c, const Queue& q, const uint32_t* fill, uint32_t* cursor, size_t maxItems, uint32_t* deepCount) {
  // persistent grid: exactly as many workgroups as are resident at once (registers and LDS decide)
  static const int resident = [] {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_extend<S, GUARD>, kBlock, 0) != hipSuccess || n < 1) n = 4;
    return n;
  }();
  int perCU = resident;
  static const int capPerCU = std::getenv("LPT_EXTEND_BLOCKS_PER_CU") ? std::atoi(std::getenv("LPT_EXTEND_BLOCKS_PER_CU")) : 0;
  if (capPerCU > 0) perCU = std::min(perCU, capPerCU);
  size_t need = (maxItems + kBlock - 1) / kBlock;
  int grid = (int)std::max<size_t>(1, std::min<size_t>(need, (size_t)c->numCUs * perCU));
  DeepList deep{nullptr, nullptr};
  if (GUARD) {
    // room for every ray of the queue in the list, and the stacks in memory of k_extend_deep's waves
    const size_t entries = (size_t)c->capPart * kCursorRegions;
    if (entries > c->deepEntries) {
      dfree(c->dDeepIndex);
      c->deepEntries = 0;
      c->callAllocated = true;
      LPT_HIP(c, hipMalloc(&c->dDeepIndex, entries * sizeof(uint32_t)));
      c->deepEntries = entries;
    }
    if (!c->dDeepColumns) {
      c->callAllocated = true;
      LPT_HIP(c, hipMalloc(&c->dDeepColumns, (size_t)kDeepGrid * (kDeepBlock / 64) * kDeepEntries * sizeof(uint32_t)));
    }
    deep = DeepList{deepCount, c->dDeepIndex};
  }
  hipLaunchKernelGGL((k_extend<S, GUARD>), dim3(grid), dim3(kBlock), 0, c->stream, device_scene(c), q.o, q.d,
                     QueueFill{fill, c->capPart}, c->dHits, cursor, deep);
  if (GUARD)
    hipLaunchKernelGGL(k_extend_deep, dim3(kDeepGrid), dim3(kDeepBlock), 0, c->stream, device_scene(c), q.o, q.d, deep,
                       c->dHits, c->dDeepColumns);
  return 0;
}

// Which k_extend instance serves a scene whose trees may need `stackNeed` entries: up to 20 the instance whose LDS
// stacks hold that bound (16 entries: eight waves per SIMD; 20: seven), beyond that the guarded 16-entry instance
// instead of 24 / 32 / 48 / 64 entries at six / four / three / two waves.  Measured on a 400 k-triangle scene, k_extend
// per launch: bound 20 (host SAH trees) 30.1 ms exact, 30.5 guarded (the follow-up launch for the few deep rays costs
// more than the eighth wave brings); bound 22 (GPU LBVH trees) 36.6 ms with 24-entry stacks, 35.5 guarded.
// LPT_EXTEND_EXACT_STACK=1 / 0 forces the exact instances / the guarded one wherever the bound exceeds 16.
bool extend_guarded(const lpt_ctx* c) {
  static const int forced = std::getenv("LPT_EXTEND_EXACT_STACK") ? std::atoi(std::getenv("LPT_EXTEND_EXACT_STACK")) : -1;
  if (c->stackNeed <= 16) return false;
  if (forced >= 0) return forced == 0;
  return c->stackNeed > 20;
}

// `fill`: the queue's set of fill counts, `cursor`: a zeroed set of cursors, `deepCount`: a zeroed word
int dispatch_extend(lpt_ctx* c, const Queue& q, const uint32_t* fill, uint32_t* cursor, size_t maxItems, uint32_t* deepCount) {
  if (c->stackNeed <= 16) return launch_extend<16, false>(c, q, fill, cursor, maxItems, nullptr);
  if (extend_guarded(c)) return launch_extend<16, true>(c, q, fill, cursor, maxItems, deepCount);
  if (c->stackNeed <= 20) return launch_extend<20, false>(c, q, fill, cursor, maxItems, nullptr);
  if (c->stackNeed <= 24) return launch_extend<24, false>(c, q, fill, cursor, maxItems, nullptr);
  if (c->stackNeed <= 32) return launch_extend<32, false>(c, q, fill, cursor, maxItems, nullptr);
  if (c->stackNeed <= 48) return launch_extend<48, false>(c, q, fill, cursor, maxItems, nullptr);
  return launch_extend<64, false>(c, q, fill, cursor, maxItems, nullptr);
}

// c->dCounts holds kCountSets sets of kCursorWords words per bounce, zeroed before every batch.
// The set of fill counts of bounce b's queue / the set of ray cursors for bounce b's launch.
uint32_t* fills_of(const lpt_ctx* c, uint32_t b) { return c->dCounts + (size_t)b * kCursorWords; }
uint32_t* cursors_of(const lpt_ctx* c, uint32_t b) { return c->dCounts + ((size_t)c->poolBounces + b) * kCursorWords; }
// The word that counts the rays bounce b's guarded k_extend launch leaves to k_extend_deep.
uint32_t* deep_count_of(const lpt_ctx* c, uint32_t b) { return c->dCounts + (2 * (size_t)c->poolBounces + b) * kCursorWords; }

// Frames per packet of the camera-ray kernels: the largest power of two <= 4 that divides the batch's frames
// (LPT_PRIMARY_PACKET_FRAMES sets another limit up to 64; 1 = one 8x8 pixel block of one frame per packet).
// Measured, 1080p, 64 frames per batch, whole step: 1 / 4 / 16 / 64 frames per packet — Cornell box 35.82 / 35.72 / 35.86 /
// 35.96 ms, 400 k-triangle terrain 140.6 / 138.9 / 138.6 / 138.5 ms.  A pixel block's rays already share most of their
// walk (the kernel is bound by vector issue, not by the size of the union), so narrower packets gain little.
uint32_t packet_frames(int frames) {
  static const int limit = std::getenv("LPT_PRIMARY_PACKET_FRAMES") ? std::atoi(std::getenv("LPT_PRIMARY_PACKET_FRAMES")) : 4;
  uint32_t f = 1;
  while (2 * f <= (uint32_t)std::min(limit, 64) && frames % (int)(2 * f) == 0) f *= 2;
  return f;
}

// Camera rays of `frames` frames into queue `q` (as k_generate leaves them) and their closest hits into c->dHits.
void launch_primary(lpt_ctx* c, const Queue& q, int frames) {
  const size_t paths = (size_t)frames * c->part.nSlots;
  if (paths == 0 || c->nValid == 0) return;
  const DScene sc = device_scene(c);
  hipLaunchKernelGGL(k_primary, dim3((unsigned)((paths + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream, sc.nodes, sc.tris,
                     sc.otrav, sc.rootRef, c->part, c->dUbos, (uint32_t)frames, packet_frames(frames), c->dValidBefore, c->nValid,
                     PrimaryQueue{q.o, q.d, q.m}, c->dHits);
}

// The whole first segment of `frames` frames: paths that go on land in queue `out`, fill counts of bounce 1.
void launch_primary_shade(lpt_ctx* c, const Queue& out, int frames) {
  const size_t paths = (size_t)frames * c->part.nSlots;
  if (paths == 0 || c->nValid == 0) return;
  const DScene sc = device_scene(c);
  hipLaunchKernelGGL(k_primary_shade, dim3((unsigned)((paths + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream, sc.nodes,
                     sc.tris, sc.otrav, sc, c->part, c->dUbos, (uint32_t)frames, out, fills_of(c, 1), c->capPart, c->dSamples,
                     (uint32_t)c->maxBounces, c->bsdfMode, packet_frames(frames));
}

// Shading of bounce b: queue `in` with c->dHits -> queue `out`; fill counts of bounces b and b + 1.  The grid is
// a multiple of the number of part queues (every workgroup shades one part).
void launch_shade(lpt_ctx* c, const Queue& in, const Queue& out, uint32_t b, uint32_t nB, size_t paths) {
  static const int resident = [] {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_shade, kBlock, 0) != hipSuccess || n < 1) n = 4;
    return n;
  }();
  static const int capPerCU = std::getenv("LPT_SHADE_BLOCKS_PER_CU") ? std::atoi(std::getenv("LPT_SHADE_BLOCKS_PER_CU")) : 0;
  const int perCU = capPerCU > 0 ? std::min(resident, capPerCU) : resident;
  const int grid = (grid_for(c, paths, perCU) + (int)kCursorRegions - 1) / (int)kCursorRegions * (int)kCursorRegions;
  hipLaunchKernelGGL(k_shade, dim3(grid), dim3(kBlock), 0, c->stream, device_scene(c), in, c->dHits,
                     QueueFill{fills_of(c, b), c->capPart}, out, fills_of(c, b + 1), c->dSamples, b, nB, c->bsdfMode);
}

// `ubos` / `samples`: null = the context's pool (c->dUbos, c->dSamples)
template <bool QUEUED>
void dispatch_trace(lpt_ctx* c, int frames, uint32_t* cursor, const TraceQueue& src, const lpt_ubo* ubos = nullptr,
                    float4* samples = nullptr);

int render_batch_wavefront(lpt_ctx* c, const lpt_ubo* ubos, int frames) {
  int rc = ensure_pool(c, frames);
  if (rc) return rc;
  const size_t paths = (size_t)frames * c->part.nSlots;
  const uint32_t nB = (uint32_t)c->maxBounces;
  LPT_HIP(c, hipMemsetAsync(c->dCounts, 0, kCountSets * kCursorWords * (size_t)c->poolBounces * sizeof(uint32_t), c->stream));
  rc = stage_frames(c, ubos, frames, true);
  if (rc) return rc;
  if (c->nValid && !primary_packets())  // the packet kernel generates the camera rays itself
    hipLaunchKernelGGL(k_generate, dim3(grid_for(c, paths, 8)), dim3(kBlock), 0, c->stream, c->part, c->dUbos,
                       (uint32_t)frames, c->queues[0], c->dValidBefore, c->nValid);
  rc = ensure_events(c, c->evUsed + 2 * (size_t)nB);
  if (rc) return rc;
  // From bounce `tail` on one persistent launch finishes the paths: once a queue has become short (Russian
  // roulette thins the paths out from bounce 3; in an open scene most rays leave) a launch pair per bounce costs
  // more in start-up and tails than it does work.  Where that happens depends on the scene, so the choice follows
  // the queue fills the last finished batch reported: the first bounce whose queue held fewer than kShortQueue
  // rays (scaled to this batch's size); without a report, or when even the last reported queue was long, no
  // hand-over at all.  Measured, 1080p: Cornell box, 64 frames, 3.3 M rays in bounce 7: a launch pair per bounce
  // for every bounce +2.2 % over handing over at bounce 4; a scene whose queues drop to 66 k rays at bounce 5:
  // handing over there +1 % over no hand-over.  LPT_WAVEFRONT_TAIL=b fixes the bounce.
  static const int tailFixed = std::getenv("LPT_WAVEFRONT_TAIL") ? std::atoi(std::getenv("LPT_WAVEFRONT_TAIL")) : 0;
  uint32_t tail = nB;
  if (tailFixed >= 1) {
    tail = std::min<uint32_t>((uint32_t)tailFixed, nB);
  } else {
    // A consistent copy of the last report of this generation, if there is one: the batch that wrote it has run to
    // its end (the generation word is set last), and no later batch has begun to overwrite it (it is cleared first).
    const volatile uint32_t* rep = c->hFlags + 4;
    uint32_t copy[kReportWords];
    bool have = rep[kReportGeneration] == c->reportGeneration;
    if (have) {
      std::atomic_thread_fence(std::memory_order_acquire);
      for (uint32_t k = 0; k < kReportWords; ++k) copy[k] = rep[k];
      std::atomic_thread_fence(std::memory_order_acquire);
      have = rep[kReportGeneration] == c->reportGeneration;
    }
    const uint32_t repBounces = have ? copy[kReportBounces] : 0u, repPaths = have ? copy[kReportBounces + 1] : 0u;
    if (repBounces >= 1u && repPaths > 0u) {
      constexpr double kShortQueue = 1024.0 * 1024.0;
      const double scale = (double)paths / (double)repPaths;
      const uint32_t known = std::min(std::min(repBounces, kReportBounces - 1u), nB - 1u);  // fills [0, known] are reported
      for (uint32_t b = 1; b <= known; ++b)
        if ((double)copy[b] * scale < kShortQueue) {
          tail = b;
          break;
        }
      // every reported queue was long: if the report ended at a hand-over, try one bounce further
      if (tail == nB && repBounces < nB) tail = std::min(nB, known + 1u);
    }
  }
  c->timedLo = primary_packets() ? 1u : 0u;
  c->timedHi = tail;
  for (uint32_t b = 0; b < tail; ++b) {
    const Queue& in = c->queues[b & 1];
    const Queue& out = c->queues[(b + 1) & 1];
    if (b == 0 && primary_packets() && primary_fused()) {
      launch_primary_shade(c, out, frames);
      continue;
    }
    if (b == 0 && primary_packets()) {
      launch_primary(c, in, frames);
    } else {  // the k_extend launches are timed one by one (lpt_stats.extend_ms)
      LPT_HIP(c, hipEventRecord(c->evPool[c->evUsed++], c->stream));
      dispatch_extend(c, in, fills_of(c, b), cursors_of(c, b), paths, deep_count_of(c, b));
      LPT_HIP(c, hipEventRecord(c->evPool[c->evUsed++], c->stream));
    }
    launch_shade(c, in, out, b, nB, paths);
  }
  if (tail < nB) {
    const Queue& q = c->queues[tail & 1];
    dispatch_trace<true>(c, frames, cursors_of(c, tail),
                         TraceQueue{q.o, q.d, q.m, QueueFill{fills_of(c, tail), c->capPart}, tail, 0u});
  }
  hipLaunchKernelGGL(k_accumulate, dim3(grid_for(c, c->part.nSlots, 8)), dim3(kBlock), 0, c->stream, c->part, c->dUbos,
                     (uint32_t)frames, c->dSamples, c->dAccum);
  // segments of the bounces run here; the persistent launch has added those it traced
  hipLaunchKernelGGL(k_tally, dim3(1), dim3(64), 0, c->stream, c->dCounts, tail, primary_packets() ? 1u : 0u, c->dTotals,
                     tail < nB ? 1u : 0u, c->dFlags + 4, c->reportGeneration);
  LPT_HIP(c, hipGetLastError());
  return 0;
}

// QUEUED: the paths come from `src` (at most frames * slots entries), `cursor` = one zeroed set of cursors.
template <int S, bool QUEUED>
void launch_trace(lpt_ctx* c, int frames, uint32_t* cursor, const TraceQueue& src, const lpt_ubo* ubos, float4* samples) {
  // persistent grid: exactly as many workgroups as are resident at once (registers and LDS decide)
  static const int resident = [] {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_trace<S, QUEUED>, kBlock, 0) != hipSuccess || n < 1) n = 4;
    return n;
  }();
  int perCU = resident;
  static const int capPerCU = std::getenv("LPT_TRACE_BLOCKS_PER_CU") ? std::atoi(std::getenv("LPT_TRACE_BLOCKS_PER_CU")) : 0;
  if (capPerCU > 0) perCU = std::min(perCU, capPerCU);
  const size_t paths = (size_t)frames * c->part.nSlots;
  const size_t need = (paths + kBlock - 1) / kBlock;
  const int grid = (int)std::max<size_t>(1, std::min<size_t>(need, (size_t)c->numCUs * perCU));
  hipLaunchKernelGGL((k_trace<S, QUEUED>), dim3(grid), dim3(kBlock), 0, c->stream, device_scene(c), c->part,
                     ubos ? ubos : c->dUbos, (uint32_t)frames, (uint32_t)c->maxBounces, c->bsdfMode,
                     samples ? samples : c->dSamples, cursor, c->dTotals, src, c->dFlags, c->traceRoundLimit);
}

template <bool QUEUED>
void dispatch_trace(lpt_ctx* c, int frames, uint32_t* cursor, const TraceQueue& src, const lpt_ubo* ubos, float4* samples) {
  if (c->stackNeed <= 16) launch_trace<16, QUEUED>(c, frames, cursor, src, ubos, samples);
  else if (c->stackNeed <= 20) launch_trace<20, QUEUED>(c, frames, cursor, src, ubos, samples);
  else if (c->stackNeed <= 24) launch_trace<24, QUEUED>(c, frames, cursor, src, ubos, samples);
  else if (c->stackNeed <= 32) launch_trace<32, QUEUED>(c, frames, cursor, src, ubos, samples);
  else if (c->stackNeed <= 48) launch_trace<48, QUEUED>(c, frames, cursor, src, ubos, samples);
  else launch_trace<64, QUEUED>(c, frames, cursor, src, ubos, samples);
}

// A batch of frames as one persistent launch plus the frame-ordered accumulation.
int render_batch_persistent(lpt_ctx* c, const lpt_ubo* ubos, int frames) {
  int rc = ensure_pool(c, frames);
  if (rc) return rc;
  rc = stage_frames(c, ubos, frames, false);
  if (rc) return rc;
  LPT_HIP(c, hipMemsetAsync(c->dCounts, 0, kCursorWords * sizeof(uint32_t), c->stream));
  rc = ensure_events(c, c->evUsed + 2);
  if (rc) return rc;
  c->timedLo = 0u;
  c->timedHi = (uint32_t)c->maxBounces;
  LPT_HIP(c, hipEventRecord(c->evPool[c->evUsed++], c->stream));
  dispatch_trace<false>(c, frames, c->dCounts, TraceQueue{nullptr, nullptr, nullptr, QueueFill{nullptr, 0u}, 0u, 1u});
  LPT_HIP(c, hipEventRecord(c->evPool[c->evUsed++], c->stream));
  hipLaunchKernelGGL(k_accumulate, dim3(grid_for(c, c->part.nSlots, 8)), dim3(kBlock), 0, c->stream, c->part, c->dUbos,
                     (uint32_t)frames, c->dSamples, c->dAccum);
  LPT_HIP(c, hipGetLastError());
  return 0;
}

// A batch of frames in two parts: the camera rays run through the wavefront kernels (generated into a
// queue, traced as packets, their hits shaded at full waves), the paths that go on from there are
// finished by one persistent launch that takes them from the queue.
int render_batch_hybrid(lpt_ctx* c, const lpt_ubo* ubos, int frames) {
  int rc = ensure_pool(c, frames);
  if (rc) return rc;
  const size_t paths = (size_t)frames * c->part.nSlots;
  const uint32_t nB = (uint32_t)c->maxBounces;
  // (poolBounces >= 2: the sets of bounce 0 and bounce 1 exist)
  LPT_HIP(c, hipMemsetAsync(c->dCounts, 0, kCountSets * kCursorWords * (size_t)c->poolBounces * sizeof(uint32_t), c->stream));
  rc = stage_frames(c, ubos, frames, true);
  if (rc) return rc;
  rc = ensure_events(c, c->evUsed + 2);
  if (rc) return rc;
  const Queue& q0 = c->queues[0];
  const Queue& q1 = c->queues[1];
  if (c->nValid) {
    if (primary_packets() && primary_fused()) {
      launch_primary_shade(c, q1, frames);
    } else if (primary_packets()) {
      launch_primary(c, q0, frames);
      launch_shade(c, q0, q1, 0, nB, paths);
    } else {
      hipLaunchKernelGGL(k_generate, dim3(grid_for(c, paths, 8)), dim3(kBlock), 0, c->stream, c->part, c->dUbos,
                         (uint32_t)frames, q0, c->dValidBefore, c->nValid);
      dispatch_extend(c, q0, fills_of(c, 0), cursors_of(c, 0), paths, deep_count_of(c, 0));
      launch_shade(c, q0, q1, 0, nB, paths);
    }
  }
  c->timedLo = 1u;
  c->timedHi = nB;
  LPT_HIP(c, hipEventRecord(c->evPool[c->evUsed++], c->stream));
  if (nB > 1 && c->nValid)
    dispatch_trace<true>(c, frames, cursors_of(c, 1),
                         TraceQueue{q1.o, q1.d, q1.m, QueueFill{fills_of(c, 1), c->capPart}, 1u, 1u});
  LPT_HIP(c, hipEventRecord(c->evPool[c->evUsed++], c->stream));
  hipLaunchKernelGGL(k_accumulate, dim3(grid_for(c, c->part.nSlots, 8)), dim3(kBlock), 0, c->stream, c->part, c->dUbos,
                     (uint32_t)frames, c->dSamples, c->dAccum);
  // camera paths and their segments; the persistent launch has added the segments it traced
  hipLaunchKernelGGL(k_tally, dim3(1), dim3(64), 0, c->stream, c->dCounts, 1u, 1u, c->dTotals, 0u, (uint32_t*)nullptr, 0u);
  LPT_HIP(c, hipGetLastError());
  return 0;
}

template <int S>
void launch_mega(lpt_ctx* c, const lpt_ubo& ubo) {
  int grid = (int)((c->part.nSlots + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(k_megakernel<S>, dim3(grid), dim3(kBlock), 0, c->stream, device_scene(c), c->part, ubo,
                     (uint32_t)c->maxBounces, c->bsdfMode, c->dAccum, c->dTotals);
}

int render_frame_mega(lpt_ctx* c, const lpt_ubo& ubo) {
  if (c->stackNeed <= 16) launch_mega<16>(c, ubo);
  else if (c->stackNeed <= 32) launch_mega<32>(c, ubo);
  else launch_mega<64>(c, ubo);
  LPT_HIP(c, hipGetLastError());
  return 0;
}

int set_partition(lpt_ctx* c, int W, int H, int tileW, int tileH, int rank, int nRanks) {
  Partition P{};
  P.W = W;
  P.H = H;
  P.tileW = tileW;
  P.tileH = tileH;
  P.tilesX = (W + tileW - 1) / tileW;
  P.tilesY = (H + tileH - 1) / tileH;
  P.rank = rank;
  P.nRanks = nRanks;
  P.nLocalTiles = (P.tilesX * P.tilesY + nRanks - 1) / nRanks;
  size_t slots = (size_t)P.nLocalTiles * tileW * tileH;
  if (slots >= (1ull << 31)) return fail(c, LPT_ERR_INVALID, "image too large for one context");
  P.nSlots = (uint32_t)slots;
  P.dense = (W % tileW == 0 && H % tileH == 0 && (P.tilesX * P.tilesY) % nRanks == 0) ? 1 : 0;
  LPT_USE_DEVICE(c);
  // nothing renders into a half-built image: the context counts as unsized until every buffer exists
  c->sized = false;
  c->part.nSlots = 0;
  c->nValid = 0;
  dfree(c->dAccum);
  dfree(c->dImage);
  dfree(c->dValidBefore);
  free_pool(c);
  c->nValid = P.nSlots;
  if (!P.dense) {
    std::vector<uint32_t> before(P.nSlots);
    uint32_t nValid = 0;
    for (uint32_t s = 0; s < P.nSlots; ++s) {
      int x, y;
      before[s] = nValid;
      if (slot_to_pixel(P, s, x, y)) ++nValid;
    }
    LPT_HIP(c, hipMalloc(&c->dValidBefore, slots * sizeof(uint32_t)));
    LPT_HIP(c, hipMemcpy(c->dValidBefore, before.data(), slots * sizeof(uint32_t), hipMemcpyHostToDevice));
    c->nValid = nValid;
  }
  LPT_HIP(c, hipMalloc(&c->dAccum, slots * sizeof(float4)));
  LPT_HIP(c, hipMalloc(&c->dImage, (size_t)W * H * sizeof(float4)));
  LPT_HIP(c, hipMemsetAsync(c->dAccum, 0, slots * sizeof(float4), c->stream));
  c->part = P;
  c->sized = true;
  next_report_generation(c);
  return 0;
}

int ready(lpt_ctx* c) {
  if (!c) return LPT_ERR_INVALID;
  if (!c->sceneLoaded) return fail(c, LPT_ERR_INVALID, "no scene uploaded");
  if (!c->sized) return fail(c, LPT_ERR_INVALID, "lpt_resize has not been called");
  return 0;
}

}  // namespace

int lpt_internal_device(const lpt_ctx* ctx) { return ctx->device; }
hipStream_t lpt_internal_stream(const lpt_ctx* ctx) { return ctx->stream; }
int lpt_internal_fail(lpt_ctx* ctx, int code, const char* msg) { return fail(ctx, code, msg); }

// ---------------------------------------------------------------------------------------
// C-ABI
// ---------------------------------------------------------------------------------------

extern "C" {

int lpt_create(int device, lpt_ctx** out) {
  if (!out) return LPT_ERR_INVALID;
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    g_create_error = std::string("no HIP device available: ") + hipGetErrorString(e);
    return LPT_ERR_HIP;
  }
  if (device < 0 || device >= n) {
    g_create_error = "device index out of range";
    return LPT_ERR_INVALID;
  }
  lpt_ctx* c = new (std::nothrow) lpt_ctx();
  if (!c) return LPT_ERR_NOMEM;
  c->device = device;
  auto bail = [&](const char* what, hipError_t err) {
    g_create_error = std::string(what) + ": " + hipGetErrorString(err);
    lpt_destroy(c);
    return LPT_ERR_HIP;
  };
  if ((e = hipSetDevice(device)) != hipSuccess) return bail("hipSetDevice", e);
  hipDeviceProp_t prop;
  if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) return bail("hipGetDeviceProperties", e);
  c->numCUs = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if ((e = hipStreamCreateWithFlags(&c->ownStream, hipStreamNonBlocking)) != hipSuccess)
    return bail("hipStreamCreate", e);
  c->stream = c->ownStream;
  if ((e = hipEventCreate(&c->evStart)) != hipSuccess) return bail("hipEventCreate", e);
  if ((e = hipEventCreate(&c->evStop)) != hipSuccess) return bail("hipEventCreate", e);
  for (int k = 0; k < 2; ++k)
    if ((e = hipEventCreateWithFlags(&c->evStaging[k], hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
  if ((e = hipMalloc(&c->dTotals, kTotalsWords * sizeof(unsigned long long))) != hipSuccess) return bail("hipMalloc", e);
  if ((e = hipMemset(c->dTotals, 0, kTotalsWords * sizeof(unsigned long long))) != hipSuccess) return bail("hipMemset", e);
  // this GPU's copy of the static scratch array (the device is current here)
  if ((e = hipGetSymbolAddress(reinterpret_cast<void**>(&c->dStatic), HIP_SYMBOL(scratch))) != hipSuccess)
    return bail("hipGetSymbolAddress(scratch)", e);
  if (device < kMaxStaticDevices) {  // the event that orders this GPU's users of the array (created once per GPU)
    bool have;
    {
      std::lock_guard<std::mutex> lock(g_staticMutex);
      have = g_staticUse[device].done != nullptr;
    }
    if (!have) {
      hipEvent_t ev = nullptr;
      if ((e = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
      std::lock_guard<std::mutex> lock(g_staticMutex);
      if (!g_staticUse[device].done) g_staticUse[device].done = ev;
      else (void)hipEventDestroy(ev);  // another thread's lpt_create was first
    }
  }
  void* flags = nullptr;
  if ((e = hipHostMalloc(&flags, (4 + kReportWords) * sizeof(uint32_t), hipHostMallocMapped | hipHostMallocCoherent)) != hipSuccess)
    return bail("hipHostMalloc", e);
  c->hFlags = static_cast<volatile uint32_t*>(flags);
  for (uint32_t k = 0; k < 4 + kReportWords; ++k) c->hFlags[k] = 0u;
  if ((e = hipHostGetDevicePointer(reinterpret_cast<void**>(&c->dFlags), flags, 0)) != hipSuccess)
    return bail("hipHostGetDevicePointer", e);
  *out = c;
  return LPT_OK;
}

void lpt_destroy(lpt_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->device < kMaxStaticDevices) {  // its work on the static scratch is done; the record must not outlive it
    std::lock_guard<std::mutex> lock(g_staticMutex);
    if (g_staticUse[c->device].ctx == c) g_staticUse[c->device] = StaticUse{g_staticUse[c->device].done, nullptr, nullptr, false};
  }
  free_scene(c);
  free_textures(c);
  free_pool(c);
  dfree(c->dAccum);
  dfree(c->dImage);
  dfree(c->dValidBefore);
  dfree(c->dUbos);
  dfree(c->dTotals);
  for (int k = 0; k < 2; ++k) {
    if (c->hUbos[k]) (void)hipHostFree(c->hUbos[k]);
    if (c->evStaging[k]) (void)hipEventDestroy(c->evStaging[k]);
  }
  if (c->hFlags) (void)hipHostFree(const_cast<uint32_t*>(c->hFlags));
  for (hipEvent_t e : c->evPool) (void)hipEventDestroy(e);
  if (c->evStart) (void)hipEventDestroy(c->evStart);
  if (c->evStop) (void)hipEventDestroy(c->evStop);
  if (c->ownStream) (void)hipStreamDestroy(c->ownStream);
  delete c;
}

const char* lpt_last_error(const lpt_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int lpt_set_stream(lpt_ctx* c, void* hip_stream) {
  if (!c) return LPT_ERR_INVALID;
  LPT_USE_DEVICE(c);
  LPT_HIP(c, hipStreamSynchronize(c->stream));
  c->stream = hip_stream ? (hipStream_t)hip_stream : c->ownStream;
  return check_flags(c);
}

int lpt_upload_textures(lpt_ctx* c, const lpt_texture* textures, uint32_t n) {
  if (!c) return LPT_ERR_INVALID;
  if (n > LPT_MAX_TEXTURES) return fail(c, LPT_ERR_UNSUPPORTED, "more than 512 textures");
  if (n && !textures) return fail(c, LPT_ERR_INVALID, "null texture array");
  std::vector<size_t> offset(n);
  size_t total = 0;
  for (uint32_t i = 0; i < n; ++i) {
    const lpt_texture& t = textures[i];
    if (!t.pixels || t.width == 0 || t.height == 0 || t.width > 32768u || t.height > 32768u)
      return fail(c, LPT_ERR_INVALID, "texture without pixels or with a size outside 1..32768");
    if (t.mag_filter > 1u || t.address_u > 3u || t.address_v > 3u)
      return fail(c, LPT_ERR_UNSUPPORTED, "texture filter or address mode not supported");
    offset[i] = total;
    total += ((size_t)t.width * t.height * 4 + 15) & ~(size_t)15;
  }
  LPT_HIP(c, hipSetDevice(c->device));
  LPT_HIP(c, hipStreamSynchronize(c->stream));
  free_scene(c);  // its texture indices were checked against the old set
  free_textures(c);
  if (n == 0) return LPT_OK;
  LPT_HIP(c, hipMalloc(&c->dTexPixels, total));
  LPT_HIP(c, hipMalloc(&c->dTextures, n * sizeof(DTexture)));
  std::vector<DTexture> desc(n);
  for (uint32_t i = 0; i < n; ++i) {
    const lpt_texture& t = textures[i];
    LPT_HIP(c, hipMemcpy(c->dTexPixels + offset[i], t.pixels, (size_t)t.width * t.height * 4, hipMemcpyHostToDevice));
    desc[i] = DTexture{c->dTexPixels + offset[i], t.width, t.height, t.mag_filter, t.address_u, t.address_v, 0u};
  }
  LPT_HIP(c, hipMemcpy(c->dTextures, desc.data(), n * sizeof(DTexture), hipMemcpyHostToDevice));
  c->nTextures = n;
  return LPT_OK;
}

int lpt_upload_scene(lpt_ctx* c, const lpt_object* objects, uint32_t nObj, const lpt_bvh_node* objBvh,
                     uint32_t nObjBvh, const lpt_vertex* verts, uint32_t nVerts, const lpt_bvh_node* meshBvh,
                     uint32_t nMeshBvh) {
  if (!c) return LPT_ERR_INVALID;
  if (!objects || !objBvh || !verts || !meshBvh || nObj == 0 || nObjBvh == 0 || nMeshBvh == 0)
    return fail(c, LPT_ERR_INVALID, "empty scene buffer");
  if (nVerts % 3u != 0u) return fail(c, LPT_ERR_INVALID, "vertex count is not a multiple of 3");
  for (uint32_t i = 0; i < nObj; ++i) {
    const lpt_object& o = objects[i];
    const uint32_t texIds[5] = {o.colorTexture, o.emissionTexture, o.metallicRoughnessTexture, o.transmissionTexture,
                                o.normalTexture};
    for (uint32_t id : texIds)
      if (id != 0xFFFFFFFFu && id >= c->nTextures)
        return fail(c, LPT_ERR_INVALID, "object references a texture that was not uploaded (lpt_upload_textures first)");
    if (o.verticesOffset % 3u != 0u || o.verticesOffset > nVerts)
      return fail(c, LPT_ERR_INVALID, "object verticesOffset is not a triangle boundary inside the vertex buffer");
    if (o.bvhOffset >= nMeshBvh) return fail(c, LPT_ERR_INVALID, "object bvhOffset out of range");
  }
  LPT_HIP(c, hipSetDevice(c->device));
  LPT_HIP(c, hipStreamSynchronize(c->stream));

  std::string err;
  const uint32_t nTrisIn = nVerts / 3u;

  // 1. validate both levels and size the traversal stack
  std::vector<int> needB(nMeshBvh, -2);
  std::vector<uint64_t> ownerB(nMeshBvh, kNoOwner);
  std::vector<int> objNeed(nObj, 0);
  for (uint32_t i = 0; i < nObj; ++i) {
    const uint32_t triBase = objects[i].verticesOffset / 3u;
    auto triLeaf = [&](uint32_t first, uint32_t count) -> int {
      if ((uint64_t)triBase + first + count > nTrisIn) {
        err = "mesh BVH leaf references triangles outside the vertex buffer";
        return LPT_ERR_INVALID;
      }
      return 0;
    };
    // Instances (objects with the same bvhOffset and verticesOffset) share a tree; anything else that
    // reaches the same nodes is refused: child indices are relative to bvhOffset and triangles to
    // verticesOffset (shaders/path_tracing.comp:149,160), so such objects would need their own copies.
    const uint64_t tag = ((uint64_t)objects[i].bvhOffset << 32) | objects[i].verticesOffset;
    int need = tree_need(meshBvh, nMeshBvh, objects[i].bvhOffset, needB, triLeaf, err, ownerB, tag);
    if (need < 0) return fail(c, need, err);
    objNeed[i] = need;
  }
  std::vector<int> needT(nObjBvh, -2);
  auto objLeaf = [&](uint32_t first, uint32_t count) -> int {
    if ((uint64_t)first + count > nObj) {
      err = "object BVH leaf references objects outside the object buffer";
      return LPT_ERR_INVALID;
    }
    int worst = 0;
    for (uint32_t k = first; k < first + count; ++k) worst = std::max(worst, objNeed[k]);
    return (count > 1u ? 2 : 1) + worst;  // remaining-range entry + end-of-object marker
  };
  std::vector<uint64_t> ownerT(nObjBvh, kNoOwner);
  int needScene = tree_need(objBvh, nObjBvh, 0, needT, objLeaf, err, ownerT, 0ull);
  if (needScene < 0) return fail(c, needScene, err);
  int stackNeed = needScene + 1;  // bottom-of-stack sentinel
  if (stackNeed > 64) return fail(c, LPT_ERR_STACK, "BVH needs a traversal stack deeper than 64 entries");

  // 2. order nodes and leaf triangles hottest first: best-first expansion from the scene
  //    root by world-space surface area (the chance that a random ray meets the box)
  struct Item {
    float prio;
    uint32_t seq;   // insertion order breaks ties deterministically
    int kind;       // 0 object-tree inner node, 1 mesh-tree inner node, 2 mesh-tree leaf
    uint32_t node;  // index into objBvh (kind 0) or meshBvh (kinds 1, 2)
    uint32_t obj;
    bool operator<(const Item& o) const { return prio != o.prio ? prio < o.prio : seq > o.seq; }
  };
  std::priority_queue<Item> heap;
  uint32_t seq = 0;
  std::vector<int64_t> newT(nObjBvh, -1), newB(nMeshBvh, -1), newTri(nMeshBvh, -1);
  struct Placed { int kind; uint32_t node; uint32_t obj; };
  std::vector<Placed> order;
  std::vector<Tri> tris;
  auto enterObject = [&](uint32_t oi, float prio) {
    uint32_t root = objects[oi].bvhOffset;
    heap.push(Item{prio, seq++, meshBvh[root].isLeaf ? 2 : 1, root, oi});
  };
  auto expandObjLeaf = [&](const lpt_bvh_node& n) {
    float prio = world_area(n.minCorner, n.maxCorner, nullptr);
    uint32_t first = n.indices[0], last = std::max(n.indices[0], n.indices[1]);
    for (uint32_t k = first; k < last; ++k) enterObject(k, prio);
  };
  if (objBvh[0].isLeaf) expandObjLeaf(objBvh[0]);
  else heap.push(Item{INFINITY, seq++, 0, 0u, 0u});
  while (!heap.empty()) {
    Item it = heap.top();
    heap.pop();
    if (it.kind == 0) {
      if (newT[it.node] >= 0) continue;
      newT[it.node] = (int64_t)order.size();
      order.push_back(Placed{0, it.node, 0u});
      for (int k = 0; k < 2; ++k) {
        uint32_t ci = objBvh[it.node].indices[k];
        const lpt_bvh_node& ch = objBvh[ci];
        if (ch.isLeaf) expandObjLeaf(ch);
        else heap.push(Item{world_area(ch.minCorner, ch.maxCorner, nullptr), seq++, 0, ci, 0u});
      }
    } else if (it.kind == 1) {
      if (newB[it.node] >= 0) continue;
      newB[it.node] = (int64_t)order.size();
      order.push_back(Placed{1, it.node, it.obj});
      const uint32_t base = objects[it.obj].bvhOffset;
      for (int k = 0; k < 2; ++k) {
        uint32_t ci = base + meshBvh[it.node].indices[k];
        const lpt_bvh_node& ch = meshBvh[ci];
        heap.push(Item{world_area(ch.minCorner, ch.maxCorner, objects[it.obj].worldMatrix), seq++, ch.isLeaf ? 2 : 1, ci, it.obj});
      }
    } else {
      if (newTri[it.node] >= 0) continue;
      newTri[it.node] = (int64_t)tris.size();
      const lpt_bvh_node& lf = meshBvh[it.node];
      const uint32_t triBase = objects[it.obj].verticesOffset / 3u;
      for (uint32_t t = lf.indices[0]; t < std::max(lf.indices[0], lf.indices[1]); ++t) {
        const lpt_vertex* v = verts + 3 * (size_t)(triBase + t);
        Tri tr;
        for (int k = 0; k < 3; ++k) {
          tr.v0[k] = v[0].position[k];
          tr.e1[k] = v[1].position[k] - v[0].position[k];
          tr.e2[k] = v[2].position[k] - v[0].position[k];
        }
        tr.prim = 3u * (triBase + t);
        tr.pad1 = tr.pad2 = 0.0f;
        tris.push_back(tr);
      }
    }
  }
  if (tris.size() > kMaxLeafFirst || order.size() >= kRefMarker)
    return fail(c, LPT_ERR_UNSUPPORTED, "scene too large for 25-bit triangle references");

  // 3. emit derived nodes and object records with references into the new order
  auto meshLeafRef = [&](uint32_t leafIdx) {
    const lpt_bvh_node& lf = meshBvh[leafIdx];
    uint32_t count = lf.indices[1] > lf.indices[0] ? lf.indices[1] - lf.indices[0] : 0u;
    return make_leaf_ref((uint32_t)newTri[leafIdx], count);
  };
  auto objLeafRef = [&](const lpt_bvh_node& lf) {
    uint32_t count = lf.indices[1] > lf.indices[0] ? lf.indices[1] - lf.indices[0] : 0u;
    return make_leaf_ref(lf.indices[0], count);
  };
  std::vector<TNode> nodes(std::max<size_t>(1, order.size()));
  std::memset(nodes.data(), 0, nodes.size() * sizeof(TNode));
  for (size_t i = 0; i < order.size(); ++i) {
    const Placed& pl = order[i];
    TNode t{};
    const lpt_bvh_node* src = pl.kind == 0 ? objBvh : meshBvh;
    const uint32_t base = pl.kind == 0 ? 0u : objects[pl.obj].bvhOffset;
    uint32_t refs[2];
    const lpt_bvh_node* ch[2];
    for (int k = 0; k < 2; ++k) {
      uint32_t ci = base + src[pl.node].indices[k];
      ch[k] = &src[ci];
      if (pl.kind == 0) refs[k] = ch[k]->isLeaf ? objLeafRef(*ch[k]) : (uint32_t)newT[ci];
      else refs[k] = ch[k]->isLeaf ? meshLeafRef(ci) : (uint32_t)newB[ci];
    }
    for (int k = 0; k < 2; ++k) {
      t.mnx[k] = ch[k]->minCorner[0]; t.mny[k] = ch[k]->minCorner[1]; t.mnz[k] = ch[k]->minCorner[2];
      t.mxx[k] = ch[k]->maxCorner[0]; t.mxy[k] = ch[k]->maxCorner[1]; t.mxz[k] = ch[k]->maxCorner[2];
    }
    t.ref0 = refs[0];
    t.ref1 = refs[1];
    nodes[i] = t;
  }
  const uint32_t sceneRoot = objBvh[0].isLeaf ? objLeafRef(objBvh[0]) : (uint32_t)newT[0];

  std::vector<ObjTrav> otrav(nObj);
  std::vector<ObjShade> oshade(nObj);
  std::vector<float> baseColor(4 * (size_t)nObj);
  for (uint32_t i = 0; i < nObj; ++i) {
    const lpt_object& o = objects[i];
    ObjTrav& t = otrav[i];
    ObjShade& sh = oshade[i];
    std::memset(&t, 0, sizeof t);
    std::memset(&sh, 0, sizeof sh);
    const float* mi = o.worldMatrixInverse;
    const float* mw = o.worldMatrix;
    for (int r = 0; r < 3; ++r) {
      for (int k = 0; k < 4; ++k) t.inv[4 * r + k] = sh.inv[4 * r + k] = mi[4 * k + r];
      for (int k = 0; k < 3; ++k) sh.nrm[3 * r + k] = mw[4 * k + r];
    }
    // an object never reached from the object tree keeps an empty root
    t.rootRef = make_leaf_ref(0u, 0u);
    if (meshBvh[o.bvhOffset].isLeaf) {
      if (newTri[o.bvhOffset] >= 0) t.rootRef = meshLeafRef(o.bvhOffset);
    } else if (newB[o.bvhOffset] >= 0) {
      t.rootRef = (uint32_t)newB[o.bvhOffset];
    }
    for (int k = 0; k < 3; ++k) sh.emission[k] = o.emissionFactor[k];
    float rf = o.roughnessFactor;
    sh.roughness = (rf != rf) ? 0.001f : (rf < 0.001f ? 0.001f : rf);  // max(roughness, 0.001)
    sh.metallic = o.metallicFactor;
    sh.transmission = o.transmissionFactor;
    sh.ior = o.ior;
    std::memcpy(&baseColor[4 * (size_t)i], o.baseColorFactor, 16);
    std::memcpy(sh.baseFactor, o.baseColorFactor, 16);
    auto id16 = [](uint32_t id) { return id == 0xFFFFFFFFu ? kNoTexture16 : id; };  // ids < LPT_MAX_TEXTURES
    sh.texColorEmission = id16(o.colorTexture) | (id16(o.emissionTexture) << 16);
    sh.texMetalRoughTransmission = id16(o.metallicRoughnessTexture) | (id16(o.transmissionTexture) << 16);
    sh.texNormal = id16(o.normalTexture);
    sh.textured = (o.colorTexture & o.emissionTexture & o.metallicRoughnessTexture & o.transmissionTexture &
                   o.normalTexture) != 0xFFFFFFFFu;
  }

  free_scene(c);
  float* dBase = nullptr;
  struct Cleanup {  // the staging buffer goes on every path out of this function
    float*& p;
    ~Cleanup() { (void)hipFree(p); }
  } cleanup{dBase};
  LPT_HIP(c, hipMalloc(&c->dNodes, nodes.size() * sizeof(TNode)));
  LPT_HIP(c, hipMalloc(&c->dTris, std::max<size_t>(1, tris.size()) * sizeof(Tri)));
  LPT_HIP(c, hipMalloc(&c->dObjTrav, otrav.size() * sizeof(ObjTrav)));
  LPT_HIP(c, hipMalloc(&c->dObjShade, oshade.size() * sizeof(ObjShade)));
  LPT_HIP(c, hipMalloc(&c->dVerts, std::max<size_t>(1, (size_t)nVerts) * sizeof(lpt_vertex)));
  LPT_HIP(c, hipMalloc(&dBase, baseColor.size() * sizeof(float)));
  LPT_HIP(c, hipMemcpy(c->dNodes, nodes.data(), nodes.size() * sizeof(TNode), hipMemcpyHostToDevice));
  if (!tris.empty()) LPT_HIP(c, hipMemcpy(c->dTris, tris.data(), tris.size() * sizeof(Tri), hipMemcpyHostToDevice));
  LPT_HIP(c, hipMemcpy(c->dObjTrav, otrav.data(), otrav.size() * sizeof(ObjTrav), hipMemcpyHostToDevice));
  LPT_HIP(c, hipMemcpy(c->dObjShade, oshade.data(), oshade.size() * sizeof(ObjShade), hipMemcpyHostToDevice));
  if (nVerts) LPT_HIP(c, hipMemcpy(c->dVerts, verts, (size_t)nVerts * sizeof(lpt_vertex), hipMemcpyHostToDevice));
  LPT_HIP(c, hipMemcpy(dBase, baseColor.data(), baseColor.size() * sizeof(float), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_finish_objects, dim3((nObj + 63) / 64), dim3(64), 0, c->stream, c->dObjShade, dBase, nObj);
  LPT_HIP(c, hipGetLastError());
  LPT_HIP(c, hipStreamSynchronize(c->stream));
  c->rootRef = sceneRoot;
  c->nNodes = (uint32_t)order.size();
  c->nTris = (uint32_t)tris.size();
  c->nObjects = nObj;
  c->stackNeed = stackNeed;
  c->sceneLoaded = true;
  next_report_generation(c);
  return LPT_OK;
}

int lpt_resize(lpt_ctx* c, int width, int height) {
  if (!c) return LPT_ERR_INVALID;
  if (width <= 0 || height <= 0 || width > 32768 || height > 32768) return fail(c, LPT_ERR_INVALID, "bad image size");
  LPT_USE_DEVICE(c);
  LPT_HIP(c, hipStreamSynchronize(c->stream));
  return set_partition(c, width, height, (width + 7) & ~7, (height + 7) & ~7, 0, 1);
}

int lpt_set_partition(lpt_ctx* c, int tile_w, int tile_h, int rank, int n_ranks) {
  if (!c) return LPT_ERR_INVALID;
  if (!c->sized) return fail(c, LPT_ERR_INVALID, "lpt_resize has not been called");
  if (tile_w <= 0 || tile_h <= 0 || (tile_w & 7) || (tile_h & 7)) return fail(c, LPT_ERR_INVALID, "tile size must be a positive multiple of 8");
  if (n_ranks <= 0 || rank < 0 || rank >= n_ranks) return fail(c, LPT_ERR_INVALID, "bad rank / n_ranks");
  LPT_USE_DEVICE(c);
  LPT_HIP(c, hipStreamSynchronize(c->stream));
  return set_partition(c, c->part.W, c->part.H, tile_w, tile_h, rank, n_ranks);
}

int lpt_set_options(lpt_ctx* c, int max_bounces, int bsdf_mode, int pipeline, int frames_in_flight) {
  if (!c) return LPT_ERR_INVALID;
  if (max_bounces < 1 || max_bounces > 4096) return fail(c, LPT_ERR_INVALID, "max_bounces out of range");
  if (bsdf_mode != LPT_BSDF_HEITZ && bsdf_mode != LPT_BSDF_BASIC) return fail(c, LPT_ERR_INVALID, "bad bsdf_mode");
  if (pipeline != LPT_PIPELINE_WAVEFRONT && pipeline != LPT_PIPELINE_MEGAKERNEL && pipeline != LPT_PIPELINE_PERSISTENT &&
      pipeline != LPT_PIPELINE_AUTO && pipeline != LPT_PIPELINE_HYBRID)
    return fail(c, LPT_ERR_INVALID, "bad pipeline");
  if (frames_in_flight < 0 || frames_in_flight > 4096) return fail(c, LPT_ERR_INVALID, "bad frames_in_flight");
  if (c->maxBounces != max_bounces || c->bsdfMode != bsdf_mode || c->pipeline != pipeline ||
      c->framesInFlight != frames_in_flight)
    next_report_generation(c);
  c->maxBounces = max_bounces;
  c->bsdfMode = bsdf_mode;
  c->pipeline = pipeline;
  c->framesInFlight = frames_in_flight;
  return LPT_OK;
}

// LPT_PIPELINE_AUTO takes the wavefront organisation when Mi paths x (ms per Mi path)^kAutoCostExponent reaches this.
constexpr double kAutoCostExponent = 0.64, kAutoWavefrontScore = 8.5, kAutoHysteresis = 0.12;

int lpt_render_frames(lpt_ctx* c, const lpt_ubo* ubos, int n_frames) {
  int rc = ready(c);
  if (rc) return rc;
  if (!ubos || n_frames <= 0) return fail(c, LPT_ERR_INVALID, "no frames");
  LPT_USE_DEVICE(c);
  // Enqueues behind whatever the stream still holds: the queues, counters and images are only touched in stream
  // order, the pinned staging alternates between two buffers (stage_frames), and the timing events are simply
  // recorded again — statistics are those of the latest call.  (RendererCore::drawFrame waits for its single
  // in-flight fence here, src/RendererCore.cpp:376; this layer allows the host to run one call ahead of the GPU.)
  rc = check_flags(c);  // an earlier render was cut short and nobody has asked since
  if (rc) return rc;
  // what the last call cost per path, if it has finished (the events are recorded anew below)
  if (c->haveTiming && c->timedGeneration == c->reportGeneration && c->lastFrames > 0 && c->part.nSlots > 0 &&
      hipEventQuery(c->evStop) == hipSuccess) {
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, c->evStart, c->evStop) == hipSuccess && ms > 0.0f)
      c->msPerMiPath = (double)ms / ((double)c->lastFrames * c->part.nSlots / (1024.0 * 1024.0));
  }
  (void)hipGetLastError();  // hipEventQuery reports "not ready" as an error
  c->evUsed = 0;
  c->haveTiming = false;
  c->callAllocated = false;
  LPT_HIP(c, hipMemsetAsync(c->dTotals, 0, kTotalsWords * sizeof(unsigned long long), c->stream));
  LPT_HIP(c, hipEventRecord(c->evStart, c->stream));
  if (c->pipeline == LPT_PIPELINE_MEGAKERNEL) {
    c->activePipeline = LPT_PIPELINE_MEGAKERNEL;
    for (int f = 0; f < n_frames; ++f) {
      rc = render_frame_mega(c, ubos[f]);
      if (rc) return rc;
    }
  } else {
    // as many frames per batch as wanted and as device memory allows
    int fif = std::min(auto_frames_in_flight(c), n_frames);
    // path ids of a batch must leave bit 31 free (kAccStored)
    fif = std::max(1, std::min<int>(fif, (int)(0x7FFFFFFFu / std::max<uint32_t>(c->part.nSlots, 1u))));
    // AUTO.  The first segment of every path runs in the packet kernel whenever there are queues.  After that:
    //  - HYBRID (one persistent launch for everything beyond the first segment) has no per-bounce launches
    //    and tails: best for small batches (Cornell box 1080p, one frame: 3.4x the wavefront kernels);
    //  - WAVEFRONT (one extend and one shade launch per bounce while the queues are long, the rest in one
    //    persistent launch) keeps seven waves per SIMD in the search and full waves in shading: ahead once a
    //    batch amortises its launches.  Cornell box 1080p, G segments/s by frames per batch (wavefront / hybrid):
    //    2: 4.36 / 4.84, 4: 6.03 / 6.64, 6: 6.95 / 7.22, 8: 7.45 / 7.50, 10: 7.84 / 7.69, 12: 8.18 / 7.86, 16: 8.58 / 8.06,
    //    24: 9.09 / 8.25: they cross at nine frames, 18 Mi paths, batches of 6 to 7 ms.  On a scene whose paths cost four
    //    times as much (400 k-triangle terrain, rough materials; 1 / 2 / 4 / 8 frames: 1.57 / 1.90, 2.25 / 2.58, 2.85 / 2.78,
    //    3.29 / 2.87) they cross at three to four frames, 7 Mi paths, batches of 10 ms: what a batch must amortise — the
    //    launches and their tails, which grow with the cost of the scene's longest rays — lies between a number of
    //    paths and a time.  The switch interpolates between the two scenes: wavefront when
    //    Mi paths x (ms per Mi path)^0.64 >= 8.5 (Cornell box 15.8 x 0.37^0.64 = 8.4 at the crossing, terrain 6.7 x 1.49^0.64 =
    //    8.7), the cost per path being what the last finished call of the same scene, size and options took (calls that
    //    allocated memory aside); by the path count while there is no such call yet;
    //  - PERSISTENT (no queues, 16 instead of 64 / 128 bytes per path in flight) when the queues do not fit.
    c->activePipeline = c->pipeline;
    if (c->pipeline == LPT_PIPELINE_AUTO) {
      const double miPaths = (double)fif * c->part.nSlots / (1024.0 * 1024.0);
      bool largeBatch = miPaths >= 18.0;
      if (c->msPerMiPath > 0.0) {
        // near the crossing the two are equally fast and the measured cost wobbles: the last choice of this scene, size
        // and options stands until the score has moved clearly past the threshold
        const double score = miPaths * std::pow(c->msPerMiPath, kAutoCostExponent);
        const double threshold = kAutoWavefrontScore * (c->autoChoice == LPT_PIPELINE_WAVEFRONT ? 1.0 - kAutoHysteresis
                                                        : c->autoChoice == LPT_PIPELINE_HYBRID  ? 1.0 + kAutoHysteresis
                                                                                                : 1.0);
        largeBatch = score >= threshold;
      }
      c->activePipeline = largeBatch ? LPT_PIPELINE_WAVEFRONT : LPT_PIPELINE_HYBRID;
      c->autoChoice = c->activePipeline;
    }
    rc = ensure_pool(c, fif);
    if (rc == LPT_ERR_NOMEM && c->pipeline == LPT_PIPELINE_AUTO) {
      c->activePipeline = LPT_PIPELINE_PERSISTENT;
      rc = ensure_pool(c, fif);
    }
    while (rc == LPT_ERR_NOMEM && fif > 1) {
      fif = (fif + 1) / 2;
      rc = ensure_pool(c, fif);
    }
    if (rc) return rc;
    for (int f = 0; f < n_frames; f += fif) {
      int nb = std::min(fif, n_frames - f);
      rc = c->activePipeline == LPT_PIPELINE_PERSISTENT ? render_batch_persistent(c, ubos + f, nb)
           : c->activePipeline == LPT_PIPELINE_HYBRID   ? render_batch_hybrid(c, ubos + f, nb)
                                                        : render_batch_wavefront(c, ubos + f, nb);
      if (rc) return rc;
    }
  }
  LPT_HIP(c, hipEventRecord(c->evStop, c->stream));
  c->lastFrames = (uint64_t)n_frames;
  c->haveTiming = true;
  c->timedGeneration = c->callAllocated ? 0u : c->reportGeneration;  // a call that allocated says little about the next
  return LPT_OK;
}

int lpt_render_frame(lpt_ctx* c, const lpt_ubo* ubo) { return lpt_render_frames(c, ubo, 1); }

int lpt_render_frame_static(lpt_ctx* c, const lpt_ubo* ubo) {
  int rc = ready(c);
  if (rc) return rc;
  if (!ubo) return fail(c, LPT_ERR_INVALID, "no frame");
  if (c->part.nSlots > kStaticMaxSlots)
    return fail(c, LPT_ERR_UNSUPPORTED,
                "lpt_render_frame_static: the context owns " + std::to_string(c->part.nSlots) + " pixel slots, the static "
                "scratch holds " + std::to_string(kStaticMaxSlots) + " (1920x1080); use lpt_render_frame");
  if (!c->dStatic) return fail(c, LPT_ERR_HIP, "static scratch address unknown");
  if (c->device >= kMaxStaticDevices) return fail(c, LPT_ERR_UNSUPPORTED, "lpt_render_frame_static: device index beyond 63");
  LPT_USE_DEVICE(c);
  // Under stream capture nothing may be recorded that the host reads back later (timing events), and the order
  // against other users of the array is the caller's to keep between graph launches.
  hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(c->stream, &capture) != hipSuccess) {
    (void)hipGetLastError();
    capture = hipStreamCaptureStatusNone;
  }
  const bool capturing = capture != hipStreamCaptureStatusNone;
  // everything below is stream-ordered work on memory that exists already: the static array (cursors,
  // uniform block, samples), the context's counters and the accumulation image
  uint32_t* cursor = reinterpret_cast<uint32_t*>(c->dStatic);
  lpt_ubo* dUbo = reinterpret_cast<lpt_ubo*>(c->dStatic + kStaticUboOffset);
  float4* samples = reinterpret_cast<float4*>(c->dStatic + kStaticHeaderBytes);
  c->evUsed = 0;
  c->haveTiming = false;
  c->activePipeline = LPT_PIPELINE_PERSISTENT;
  c->timedLo = 0u;
  c->timedHi = (uint32_t)c->maxBounces;
  std::lock_guard<std::mutex> lock(g_staticMutex);  // one call at a time enqueues work on this GPU's array
  StaticUse& use = g_staticUse[c->device];
  if (!capturing && use.recorded && (use.ctx != c || use.stream != c->stream))
    LPT_HIP(c, hipStreamWaitEvent(c->stream, use.done, 0));  // behind the last call of another context or stream
  LPT_HIP(c, hipMemsetAsync(c->dTotals, 0, kTotalsWords * sizeof(unsigned long long), c->stream));
  LPT_HIP(c, hipMemsetAsync(cursor, 0, kCursorWords * sizeof(uint32_t), c->stream));  // the last call's cursors
  if (!capturing) LPT_HIP(c, hipEventRecord(c->evStart, c->stream));
  hipLaunchKernelGGL(k_store_ubo, dim3(1), dim3(64), 0, c->stream, *ubo, dUbo);
  if (c->nValid)
    dispatch_trace<false>(c, 1, cursor, TraceQueue{nullptr, nullptr, nullptr, QueueFill{nullptr, 0u}, 0u, 1u}, dUbo, samples);
  hipLaunchKernelGGL(k_accumulate, dim3(grid_for(c, c->part.nSlots, 8)), dim3(kBlock), 0, c->stream, c->part, dUbo, 1u,
                     samples, c->dAccum);
  LPT_HIP(c, hipGetLastError());
  if (!capturing) {
    LPT_HIP(c, hipEventRecord(c->evStop, c->stream));
    LPT_HIP(c, hipEventRecord(use.done, c->stream));
    use.ctx = c;
    use.stream = c->stream;
    use.recorded = true;
    c->lastFrames = 1;
    c->haveTiming = true;
    c->timedGeneration = c->reportGeneration;
  }
  return LPT_OK;
}

int lpt_synchronize(lpt_ctx* c) {
  if (!c) return LPT_ERR_INVALID;
  LPT_USE_DEVICE(c);
  LPT_HIP(c, hipStreamSynchronize(c->stream));
  return check_flags(c);
}

int lpt_get_stats(lpt_ctx* c, lpt_stats* out) {
  if (!c || !out) return LPT_ERR_INVALID;
  std::memset(out, 0, sizeof *out);
  if (!c->haveTiming) return fail(c, LPT_ERR_INVALID, "nothing rendered yet");
  LPT_USE_DEVICE(c);
  LPT_HIP(c, hipStreamSynchronize(c->stream));
  {
    const int rc = check_flags(c);
    if (rc) return rc;
  }
  unsigned long long totals[kTotalsWords] = {};
  LPT_HIP(c, hipMemcpy(totals, c->dTotals, sizeof totals, hipMemcpyDeviceToHost));
#ifdef LPT_EXTEND_STATS
  {
    unsigned long long t[16 + 64];
    LPT_HIP(c, hipMemcpyFromSymbol(t, HIP_SYMBOL(lpt::g_extendStats), sizeof t));
    const char* names[6] = {"fetch", "inner", "tri", "obj", "inner+", "tri+"};
    for (int k = 0; k < 6; ++k)
      std::fprintf(stderr, "extend rounds %-7s %12llu  lanes/round %.1f\n", names[k], t[2 * k],
                   t[2 * k] ? (double)t[2 * k + 1] / (double)t[2 * k] : 0.0);
    std::fprintf(stderr, "extend waves after the queue ran dry: %.2f %% of all wave time, longest %.3f ms\n",
                 t[15] ? 100.0 * (double)t[14] / (double)t[15] : 0.0, (double)t[12] * 1e-5);
    std::fprintf(stderr, "extend rays by steps (bins of 32):");
    for (int k = 0; k < 64 && t[16 + k]; ++k) std::fprintf(stderr, " %llu", t[16 + k]);
    std::fprintf(stderr, "\n");
    unsigned long long d[64];
    LPT_HIP(c, hipMemcpyFromSymbol(d, HIP_SYMBOL(lpt::g_extendDepth), sizeof d));
    std::fprintf(stderr, "extend rays by most stack entries held:");
    for (int k = 0; k < 64; ++k) std::fprintf(stderr, " %llu", d[k]);
    std::fprintf(stderr, "\n");
  }
#endif
#ifdef LPT_TRACE_STATS
  {
    unsigned long long t[16];
    LPT_HIP(c, hipMemcpy(t, c->dTotals, sizeof t, hipMemcpyDeviceToHost));
    const char* names[5] = {"inner", "tri", "obj", "surface", "new"};
    for (int k = 0; k < 5; ++k)
      std::fprintf(stderr, "trace rounds %-8s %12llu  lanes/round %.1f\n", names[k], t[2 + 2 * k],
                   t[2 + 2 * k] ? (double)t[3 + 2 * k] / (double)t[2 + 2 * k] : 0.0);
    const double all = (double)(t[12] + t[13] + t[14] + t[15]);
    for (int k = 0; k < 4; ++k)
      std::fprintf(stderr, "trace wave time %-8s %5.1f %%\n", names[k], all > 0 ? 100.0 * (double)t[12 + k] / all : 0.0);
  }
#endif
  out->paths = totals[kTotPaths];
  out->segments = totals[kTotSegments];
  out->kernel_segments = c->activePipeline == LPT_PIPELINE_MEGAKERNEL ? totals[kTotSegments] : totals[kTotTimed];
  out->frames = c->lastFrames;
  float ms = 0.0f;
  LPT_HIP(c, hipEventElapsedTime(&ms, c->evStart, c->evStop));
  out->device_ms = ms;
  double ext = 0.0;
  for (size_t i = 0; i + 1 < c->evUsed; i += 2) {
    LPT_HIP(c, hipEventElapsedTime(&ms, c->evPool[i], c->evPool[i + 1]));
    ext += ms;
  }
  out->extend_ms = ext;
  out->extend_launches = (uint32_t)(c->evUsed / 2);
  out->pipeline = (uint32_t)c->activePipeline;
  out->timed_bounce_lo = c->activePipeline == LPT_PIPELINE_MEGAKERNEL ? 0u : c->timedLo;
  out->timed_bounce_hi = c->activePipeline == LPT_PIPELINE_MEGAKERNEL ? (uint32_t)c->maxBounces : c->timedHi;
  out->stack_need = (uint32_t)c->stackNeed;
  // what the search kernel that ran keeps in LDS: k_extend 16 entries per lane, guarded if the scene's bound is higher
  // (rays that outgrow them are searched again with a stack in memory); the other kernels the scene's bound
  out->stack_entries = c->stackNeed <= 16 || (c->activePipeline == LPT_PIPELINE_WAVEFRONT && extend_guarded(c)) ? 16u
                       : c->stackNeed <= 20 ? 20u : c->stackNeed <= 24 ? 24u : c->stackNeed <= 32 ? 32u
                       : c->stackNeed <= 48 ? 48u : 64u;
  return LPT_OK;
}

int lpt_read_accum(lpt_ctx* c, float* dst) {
  if (!c || !dst) return LPT_ERR_INVALID;
  if (!c->sized) return fail(c, LPT_ERR_INVALID, "lpt_resize has not been called");
  const Partition& P = c->part;
  LPT_USE_DEVICE(c);
  dim3 blk(16, 16), grd((P.W + 15) / 16, (P.H + 15) / 16);
  hipLaunchKernelGGL(k_unpermute, grd, blk, 0, c->stream, P, c->dAccum, c->dImage);
  LPT_HIP(c, hipGetLastError());
  LPT_HIP(c, hipMemcpyAsync(dst, c->dImage, (size_t)P.W * P.H * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
  LPT_HIP(c, hipStreamSynchronize(c->stream));
  return check_flags(c);
}

int lpt_owned_slots(lpt_ctx* c, uint64_t* n) {
  if (!c || !n) return LPT_ERR_INVALID;
  if (!c->sized) return fail(c, LPT_ERR_INVALID, "lpt_resize has not been called");
  *n = c->part.nSlots;
  return LPT_OK;
}

int lpt_owned_accum_device(lpt_ctx* c, void** p) {
  if (!c || !p) return LPT_ERR_INVALID;
  if (!c->sized) return fail(c, LPT_ERR_INVALID, "lpt_resize has not been called");
  *p = c->dAccum;
  return LPT_OK;
}

int lpt_assemble_gathered(lpt_ctx* c, const void* gathered, int n_ranks, void* image) {
  if (!c || !gathered || !image) return LPT_ERR_INVALID;
  if (!c->sized) return fail(c, LPT_ERR_INVALID, "lpt_resize has not been called");
  if (n_ranks != c->part.nRanks) return fail(c, LPT_ERR_INVALID, "n_ranks differs from the partition");
  const Partition& P = c->part;
  LPT_USE_DEVICE(c);
  dim3 blk(16, 16), grd((P.W + 15) / 16, (P.H + 15) / 16);
  hipLaunchKernelGGL(k_assemble, grd, blk, 0, c->stream, P, (const float4*)gathered, (float4*)image);
  LPT_HIP(c, hipGetLastError());
  return LPT_OK;
}

int lpt_resolve(lpt_ctx* c, float inv_sample_count, float* dst) {
  if (!c || !dst) return LPT_ERR_INVALID;
  if (!c->sized) return fail(c, LPT_ERR_INVALID, "lpt_resize has not been called");
  const Partition& P = c->part;
  LPT_USE_DEVICE(c);
  dim3 blk(16, 16), grd((P.W + 15) / 16, (P.H + 15) / 16);
  hipLaunchKernelGGL(k_resolve, grd, blk, 0, c->stream, P, c->dAccum, inv_sample_count, c->dImage);
  LPT_HIP(c, hipGetLastError());
  LPT_HIP(c, hipMemcpyAsync(dst, c->dImage, (size_t)P.W * P.H * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
  LPT_HIP(c, hipStreamSynchronize(c->stream));
  return check_flags(c);
}

int lpt_debug_set_trace_round_limit(lpt_ctx* c, uint32_t rounds) {
  if (!c) return LPT_ERR_INVALID;
  c->traceRoundLimit = rounds ? rounds : (uint32_t)LPT_TRACE_ROUND_LIMIT;
  return LPT_OK;
}

int lpt_debug_math(lpt_ctx* c, int op, const float* a, const float* b, float* out, uint32_t n) {
  if (!c || !a || !out || n == 0) return LPT_ERR_INVALID;
  LPT_USE_DEVICE(c);
  float *da = nullptr, *db = nullptr, *dout = nullptr;
  struct Cleanup {  // the buffers go on every path out of this function
    float *&a, *&b, *&o;
    ~Cleanup() {
      (void)hipFree(a);
      (void)hipFree(b);
      (void)hipFree(o);
    }
  } cleanup{da, db, dout};
  LPT_HIP(c, hipMalloc(&da, n * sizeof(float)));
  LPT_HIP(c, hipMalloc(&dout, n * sizeof(float)));
  LPT_HIP(c, hipMemcpy(da, a, n * sizeof(float), hipMemcpyHostToDevice));
  if (b) {
    LPT_HIP(c, hipMalloc(&db, n * sizeof(float)));
    LPT_HIP(c, hipMemcpy(db, b, n * sizeof(float), hipMemcpyHostToDevice));
  }
  hipLaunchKernelGGL(k_debug_math, dim3((n + 255) / 256), dim3(256), 0, c->stream, op, da, db, dout, n);
  LPT_HIP(c, hipGetLastError());
  LPT_HIP(c, hipStreamSynchronize(c->stream));
  LPT_HIP(c, hipMemcpy(out, dout, n * sizeof(float), hipMemcpyDeviceToHost));
  return LPT_OK;
}

int lpt_debug_intersect(lpt_ctx* c, const float* o, const float* d, uint32_t n, float* t, uint32_t* obj,
                        uint32_t* prim) {
  if (!c || !o || !d || !t || !obj || !prim || n == 0) return LPT_ERR_INVALID;
  if (!c->sceneLoaded) return fail(c, LPT_ERR_INVALID, "no scene uploaded");
  LPT_USE_DEVICE(c);
  float *dO = nullptr, *dD = nullptr, *dT = nullptr;
  uint32_t *dObj = nullptr, *dPrim = nullptr;
  struct Cleanup {  // the buffers go on every path out of this function
    float *&o, *&d, *&t;
    uint32_t *&obj, *&prim;
    ~Cleanup() {
      (void)hipFree(o);
      (void)hipFree(d);
      (void)hipFree(t);
      (void)hipFree(obj);
      (void)hipFree(prim);
    }
  } cleanup{dO, dD, dT, dObj, dPrim};
  LPT_HIP(c, hipMalloc(&dO, 3 * (size_t)n * sizeof(float)));
  LPT_HIP(c, hipMalloc(&dD, 3 * (size_t)n * sizeof(float)));
  LPT_HIP(c, hipMalloc(&dT, n * sizeof(float)));
  LPT_HIP(c, hipMalloc(&dObj, n * sizeof(uint32_t)));
  LPT_HIP(c, hipMalloc(&dPrim, n * sizeof(uint32_t)));
  LPT_HIP(c, hipMemcpy(dO, o, 3 * (size_t)n * sizeof(float), hipMemcpyHostToDevice));
  LPT_HIP(c, hipMemcpy(dD, d, 3 * (size_t)n * sizeof(float), hipMemcpyHostToDevice));
  dim3 grd((n + kBlock - 1) / kBlock), blk(kBlock);
  DScene sc = device_scene(c);
  if (c->stackNeed <= 16) hipLaunchKernelGGL(k_debug_intersect<16>, grd, blk, 0, c->stream, sc, dO, dD, n, dT, dObj, dPrim);
  else if (c->stackNeed <= 32) hipLaunchKernelGGL(k_debug_intersect<32>, grd, blk, 0, c->stream, sc, dO, dD, n, dT, dObj, dPrim);
  else hipLaunchKernelGGL(k_debug_intersect<64>, grd, blk, 0, c->stream, sc, dO, dD, n, dT, dObj, dPrim);
  LPT_HIP(c, hipGetLastError());
  LPT_HIP(c, hipStreamSynchronize(c->stream));
  LPT_HIP(c, hipMemcpy(t, dT, n * sizeof(float), hipMemcpyDeviceToHost));
  LPT_HIP(c, hipMemcpy(obj, dObj, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
  LPT_HIP(c, hipMemcpy(prim, dPrim, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
  return LPT_OK;
}

}  // extern "C"
