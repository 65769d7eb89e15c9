/* Analysis aid (not product, not oracle): how many traversal steps would a closest-hit search need on the
 * interchange trees under different visiting orders?  Rays come from a simple diffuse/mirror/glass walk.
 *
 *   SH  the order path_tracing.comp:130-209 fixes (second child first, cull against the hit so far)
 *   NF  nearer child first, cull against hit*(1+rel)+abs
 *   BD  the fixed order again, but started with the bound hit*(1+rel)+abs instead of infinity
 *
 * Counted per segment: inner-node visits (one visit = both children box-tested), triangle tests, objects
 * entered; for NF also how often a second primitive lies within the (rel, abs) window of the nearest one.
 *
 * gcc -O2 -fopenmp -ffp-contract=off -shared -fPIC -o /tmp/liborder_study.so tools/order_study.c -lm
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

typedef struct { float worldMatrix[16], worldMatrixInverse[16], baseColorFactor[4], emissionFactor[3];
  float metallicFactor, roughnessFactor, transmissionFactor, ior; uint32_t tex[5], bvhOffset, verticesOffset; uint8_t pad[8]; } object_t;
typedef struct { float position[3], pad0, normal[3], pad1, uv[2], pad2[2]; } vertex_t;
typedef struct { float mn[3], pad0, mx[3]; uint32_t isLeaf, idx[2], pad1[2]; } node_t;
typedef struct { const object_t* objects; uint32_t nObjects; const node_t* tlas; uint32_t nTlas;
  const vertex_t* vertices; uint32_t nVertices; const node_t* blas; uint32_t nBlas; } scene_t;

typedef struct { float x, y, z; } v3;
static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 scale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline float dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline v3 cross(v3 a, v3 b) { return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static inline v3 normalize(v3 a) { return scale(a, 1.0f / sqrtf(dot(a, a))); }
static inline v3 mpoint(const float* m, v3 p) {
  return V(((m[0] * p.x + m[4] * p.y) + m[8] * p.z) + m[12], ((m[1] * p.x + m[5] * p.y) + m[9] * p.z) + m[13],
           ((m[2] * p.x + m[6] * p.y) + m[10] * p.z) + m[14]);
}
static inline v3 mvec(const float* m, v3 p) {
  return V((m[0] * p.x + m[4] * p.y) + m[8] * p.z, (m[1] * p.x + m[5] * p.y) + m[9] * p.z, (m[2] * p.x + m[6] * p.y) + m[10] * p.z);
}
typedef struct { v3 o, d, inv; } ray_t;
static inline ray_t mkray(v3 o, v3 d) { ray_t r; r.o = o; r.d = d; r.inv = V(1.0f / d.x, 1.0f / d.y, 1.0f / d.z); return r; }

/* slab test; returns 1 when the box passes against `dist`, *t0out = entry distance (<=0 when inside) */
static inline int box(const ray_t* r, const node_t* n, float dist, float* t0out) {
  float nx = (n->mn[0] - r->o.x) * r->inv.x, fx = (n->mx[0] - r->o.x) * r->inv.x;
  float ny = (n->mn[1] - r->o.y) * r->inv.y, fy = (n->mx[1] - r->o.y) * r->inv.y;
  float nz = (n->mn[2] - r->o.z) * r->inv.z, fz = (n->mx[2] - r->o.z) * r->inv.z;
  float t0 = fmaxf(fmaxf(fminf(nx, fx), fminf(ny, fy)), fminf(nz, fz));
  float t1 = fminf(fminf(fmaxf(nx, fx), fmaxf(ny, fy)), fmaxf(nz, fz));
  *t0out = t0;
  if (t0 > t1) return 0;
  if (t0 > 0.0f) return t0 < dist;
  return t1 > 0.0f;
}
static inline float tri(const ray_t* r, v3 v0, v3 v1, v3 v2) {
  v3 e1 = sub(v1, v0), e2 = sub(v2, v0), p = cross(r->d, e2);
  float det = 1.0f / dot(e1, p);
  v3 tv = sub(r->o, v0);
  float u = dot(tv, p) * det;
  if (u < 0.0f || u > 1.0f) return 3.4e38f;
  v3 q = cross(tv, e1);
  float v = dot(r->d, q) * det;
  if (v < 0.0f || u + v > 1.0f) return 3.4e38f;
  return dot(e2, q) * det;
}
static inline v3 vp(const vertex_t* v) { return V(v->position[0], v->position[1], v->position[2]); }

typedef struct { uint64_t inner, tris, objs; } cnt_t;
typedef struct { float t; uint32_t obj, prim; } hit_t;

#define STK 256
/* fixed order, optional initial bound */
static hit_t search_fixed(const scene_t* s, const ray_t* ray, float bound, cnt_t* c) {
  hit_t h = {bound, 0xffffffffu, 0};
  int st[STK], sp = 0; st[sp++] = -1; int idx = 0;
  while (idx > -1) {
    const node_t* n = &s->tlas[idx];
    if (n->isLeaf) {
      for (uint32_t oi = n->idx[0]; oi < n->idx[1]; oi++) {
        const object_t* ob = &s->objects[oi];
        c->objs++;
        ray_t ro = mkray(mpoint(ob->worldMatrixInverse, ray->o), mvec(ob->worldMatrixInverse, ray->d));
        int st2[STK], sp2 = 0; st2[sp2++] = -1; int j = (int)ob->bvhOffset;
        while (j > -1) {
          const node_t* m = &s->blas[j];
          if (m->isLeaf) {
            for (uint32_t i = m->idx[0]; i < m->idx[1]; i++) {
              uint32_t fv = ob->verticesOffset + 3u * i;
              c->tris++;
              float t = tri(&ro, vp(&s->vertices[fv]), vp(&s->vertices[fv + 1]), vp(&s->vertices[fv + 2]));
              if (t > 1e-4f && t < h.t) { h.t = t; h.obj = oi; h.prim = fv; }
            }
          } else {
            float t0; c->inner++;
            int a = (int)(ob->bvhOffset + m->idx[0]), b = (int)(ob->bvhOffset + m->idx[1]);
            if (box(&ro, &s->blas[a], h.t, &t0)) st2[sp2++] = a;
            if (box(&ro, &s->blas[b], h.t, &t0)) st2[sp2++] = b;
          }
          j = st2[--sp2];
        }
      }
    } else {
      float t0; c->inner++;
      int a = (int)n->idx[0], b = (int)n->idx[1];
      if (box(ray, &s->tlas[a], h.t, &t0)) st[sp++] = a;
      if (box(ray, &s->tlas[b], h.t, &t0)) st[sp++] = b;
    }
    idx = st[--sp];
  }
  return h;
}

/* nearer child first; culls against best*(1+rel)+abs; reports the second-nearest distinct primitive's distance */
static hit_t search_near(const scene_t* s, const ray_t* ray, float rel, float abs_, cnt_t* c, float* second) {
  hit_t h = {3.4e38f, 0xffffffffu, 0};
  float sec = 3.4e38f, lim = 3.4e38f;
  int st[STK], sp = 0; st[sp++] = -1; int idx = 0;
  while (idx > -1) {
    const node_t* n = &s->tlas[idx];
    if (n->isLeaf) {
      for (uint32_t oi = n->idx[0]; oi < n->idx[1]; oi++) {
        const object_t* ob = &s->objects[oi];
        c->objs++;
        ray_t ro = mkray(mpoint(ob->worldMatrixInverse, ray->o), mvec(ob->worldMatrixInverse, ray->d));
        int st2[STK], sp2 = 0; st2[sp2++] = -1; int j = (int)ob->bvhOffset;
        while (j > -1) {
          const node_t* m = &s->blas[j];
          if (m->isLeaf) {
            for (uint32_t i = m->idx[0]; i < m->idx[1]; i++) {
              uint32_t fv = ob->verticesOffset + 3u * i;
              c->tris++;
              float t = tri(&ro, vp(&s->vertices[fv]), vp(&s->vertices[fv + 1]), vp(&s->vertices[fv + 2]));
              if (t > 1e-4f && t < 3.0e38f && !(h.obj == oi && h.prim == fv)) {
                if (t < h.t) { sec = h.t; h.t = t; h.obj = oi; h.prim = fv; lim = t * (1.0f + rel) + abs_; }
                else if (t < sec) sec = t;
              }
            }
          } else {
            float ta, tb; c->inner++;
            int a = (int)(ob->bvhOffset + m->idx[0]), b = (int)(ob->bvhOffset + m->idx[1]);
            int pa = box(&ro, &s->blas[a], lim, &ta), pb = box(&ro, &s->blas[b], lim, &tb);
            if (pa && pb) { if (ta < tb) { st2[sp2++] = b; st2[sp2++] = a; } else { st2[sp2++] = a; st2[sp2++] = b; } }
            else if (pa) st2[sp2++] = a; else if (pb) st2[sp2++] = b;
          }
          j = st2[--sp2];
          /* a popped node is not re-tested against the shrunken limit (as the device stack would not) */
        }
      }
    } else {
      float ta, tb; c->inner++;
      int a = (int)n->idx[0], b = (int)n->idx[1];
      int pa = box(ray, &s->tlas[a], lim, &ta), pb = box(ray, &s->tlas[b], lim, &tb);
      if (pa && pb) { if (ta < tb) { st[sp++] = b; st[sp++] = a; } else { st[sp++] = a; st[sp++] = b; } }
      else if (pa) st[sp++] = a; else if (pb) st[sp++] = b;
    }
    idx = st[--sp];
  }
  *second = sec;
  return h;
}

static inline uint32_t rng(uint32_t* s) { *s = *s * 747796405u + 2891336453u; uint32_t w = ((*s >> ((*s >> 28) + 4)) ^ *s) * 277803737u; return (w >> 22) ^ w; }
static inline float rnd(uint32_t* s) { return (rng(s) >> 8) * (1.0f / 16777216.0f); }

/* out[b*12 + k]: per bounce b<16: segments, SH inner/tris/objs, NF inner/tris/objs, BD inner/tris/objs, ambiguous, mismatch */
void order_study(const scene_t* s, const float* cam, float fovY, int w, int h, int frames, int maxBounces,
                 float rel, float abs_, double* out) {
  memset(out, 0, sizeof(double) * 16 * 12);
#pragma omp parallel
  {
    double loc[16 * 12]; memset(loc, 0, sizeof loc);
#pragma omp for schedule(dynamic, 4)
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w; x++)
        for (int f = 0; f < frames; f++) {
          uint32_t seed = (uint32_t)((f * h + y) * w + x) * 9781u + 12345u;
          float th = tanf(fovY * 0.5f), asp = (float)w / (float)h;
          float px = (2.0f * ((x + rnd(&seed)) / w) - 1.0f) * th * asp, py = (1.0f - 2.0f * ((y + rnd(&seed)) / h)) * th;
          v3 o = V(cam[12], cam[13], cam[14]);
          v3 d = normalize(mvec(cam, V(px, py, -1.0f)));
          for (int b = 0; b < maxBounces && b < 16; b++) {
            ray_t ray = mkray(o, d);
            cnt_t cs = {0, 0, 0}, cn = {0, 0, 0}, cb = {0, 0, 0};
            hit_t hs = search_fixed(s, &ray, 3.4e38f, &cs);
            float sec;
            hit_t hn = search_near(s, &ray, rel, abs_, &cn, &sec);
            double* L = loc + b * 12;
            L[0] += 1; L[1] += cs.inner; L[2] += cs.tris; L[3] += cs.objs;
            L[4] += cn.inner; L[5] += cn.tris; L[6] += cn.objs;
            if (hs.obj != 0xffffffffu) {
              hit_t hb = search_fixed(s, &ray, hs.t * (1.0f + rel) + abs_, &cb);
              L[7] += cb.inner; L[8] += cb.tris; L[9] += cb.objs;
              if (hb.obj != hs.obj || hb.prim != hs.prim || hb.t != hs.t) L[11] += 1000000;
            }
            int amb = hn.obj != 0xffffffffu && sec < hn.t * (1.0f + rel) + abs_;
            L[10] += amb;
            if (!amb && (hn.obj != hs.obj || hn.prim != hs.prim || hn.t != hs.t)) L[11] += 1;
            if (hs.obj == 0xffffffffu) break;
            const object_t* ob = &s->objects[hs.obj];
            v3 p0 = mpoint(ob->worldMatrix, vp(&s->vertices[hs.prim])), p1 = mpoint(ob->worldMatrix, vp(&s->vertices[hs.prim + 1])),
               p2 = mpoint(ob->worldMatrix, vp(&s->vertices[hs.prim + 2]));
            v3 n = normalize(cross(sub(p1, p0), sub(p2, p0)));
            v3 pos = add(o, scale(d, hs.t));
            if (ob->emissionFactor[0] + ob->emissionFactor[1] + ob->emissionFactor[2] > 0.0f) break;
            if (ob->transmissionFactor > 0.5f) { o = pos; /* straight through */ }
            else if (ob->metallicFactor > 0.5f) { d = sub(d, scale(n, 2.0f * dot(n, d))); o = pos; }
            else {
              if (dot(n, d) > 0.0f) n = scale(n, -1.0f);
              float r1 = 6.2831853f * rnd(&seed), r2 = rnd(&seed), r2s = sqrtf(r2);
              v3 a = fabsf(n.x) > 0.1f ? V(0, 1, 0) : V(1, 0, 0);
              v3 u = normalize(cross(a, n)), v = cross(n, u);
              d = normalize(add(add(scale(u, cosf(r1) * r2s), scale(v, sinf(r1) * r2s)), scale(n, sqrtf(1.0f - r2))));
              o = pos;
            }
          }
        }
#pragma omp critical
    for (int i = 0; i < 16 * 12; i++) out[i] += loc[i];
  }
}
