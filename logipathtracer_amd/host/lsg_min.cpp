#include "lsg_min.hpp"
#include "jpeg_decode.hpp"
#include "png_decode.hpp"

#include <algorithm>
#include <cstring>
#include <fstream>
#include <sstream>
#include <new>
#include <stdexcept>

#include "minijson.hpp"

namespace lsg {

// ---- linear algebra ---------------------------------------------------------------------

mat4 mat4::translation(vec3 t) {
  mat4 r;
  r.at(3, 0) = t.x;
  r.at(3, 1) = t.y;
  r.at(3, 2) = t.z;
  return r;
}

mat4 mat4::scaling(vec3 s) {
  mat4 r;
  r.at(0, 0) = s.x;
  r.at(1, 1) = s.y;
  r.at(2, 2) = s.z;
  return r;
}

mat4 mat4::rotation(quat q) {
  const float xx = q.x * q.x, yy = q.y * q.y, zz = q.z * q.z;
  const float xy = q.x * q.y, xz = q.x * q.z, yz = q.y * q.z;
  const float wx = q.w * q.x, wy = q.w * q.y, wz = q.w * q.z;
  mat4 r;
  r.at(0, 0) = 1.0f - 2.0f * (yy + zz);
  r.at(0, 1) = 2.0f * (xy + wz);
  r.at(0, 2) = 2.0f * (xz - wy);
  r.at(1, 0) = 2.0f * (xy - wz);
  r.at(1, 1) = 1.0f - 2.0f * (xx + zz);
  r.at(1, 2) = 2.0f * (yz + wx);
  r.at(2, 0) = 2.0f * (xz + wy);
  r.at(2, 1) = 2.0f * (yz - wx);
  r.at(2, 2) = 1.0f - 2.0f * (xx + yy);
  return r;
}

mat4 mat4::rotationAxis(int axis, float a) {
  const float c = std::cos(a), s = std::sin(a);
  mat4 r;
  const int i = (axis + 1) % 3, j = (axis + 2) % 3;
  r.at(i, i) = c;
  r.at(i, j) = s;
  r.at(j, i) = -s;
  r.at(j, j) = c;
  return r;
}

mat4 operator*(const mat4& a, const mat4& b) {
  mat4 r;
  for (int c = 0; c < 4; ++c)
    for (int row = 0; row < 4; ++row) {
      float s = 0.0f;
      for (int k = 0; k < 4; ++k) s += a.at(k, row) * b.at(c, k);
      r.at(c, row) = s;
    }
  return r;
}

// General 4x4 inverse by cofactors (Cramer's rule), single precision.
mat4 inverse(const mat4& a) {
  const float* m = a.m;
  float inv[16];
  inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
  inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
  inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
  inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
  inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
  inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
  inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
  inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
  inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
  inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
  inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
  inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
  inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
  inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
  inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
  inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
  float det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
  float rdet = 1.0f / det;
  mat4 r;
  for (int i = 0; i < 16; ++i) r.m[i] = inv[i] * rdet;
  return r;
}

vec3 transformPoint(const mat4& m, vec3 p) {
  return {m.at(0, 0) * p.x + m.at(1, 0) * p.y + m.at(2, 0) * p.z + m.at(3, 0),
          m.at(0, 1) * p.x + m.at(1, 1) * p.y + m.at(2, 1) * p.z + m.at(3, 1),
          m.at(0, 2) * p.x + m.at(1, 2) * p.y + m.at(2, 2) * p.z + m.at(3, 2)};
}

// ---- Transform / Object -------------------------------------------------------------------

uint64_t Transform::chainVersion() const {
  uint64_t v = version_;
  if (auto p = parent_.lock()) v += p->chainVersion() * 1000003ull;
  return v;
}

mat4 Transform::worldMatrix() {
  seenVersion_ = chainVersion();
  if (auto p = parent_.lock()) {
    // do not clear the parent's own dirty state: it may be watched separately
    Transform copy = *p;
    return copy.worldMatrix() * local_;
  }
  return local_;
}

void Object::addChild(const Ref<Object>& c) {
  children_.push_back(c);
  auto ct = c->getComponent<Transform>();
  auto pt = getComponent<Transform>();
  if (ct && pt) ct->setParent(pt);
}

void Object::traverseDown(const std::function<bool(const Ref<Object>&)>& fn) {
  if (!fn(shared_from_this())) return;
  for (auto& c : children_) c->traverseDown(fn);
}

// ---- glTF loading ----------------------------------------------------------------------------

namespace {

using minijson::Value;

std::string readFile(const std::string& path, bool binary) {
  std::ifstream f(path, binary ? std::ios::binary : std::ios::in);
  if (!f) throw std::runtime_error("failed to open file: " + path);
  std::ostringstream ss;
  ss << f.rdbuf();
  return ss.str();
}

std::string decodeBase64(const std::string& in) {
  std::string out;
  int val = 0, bits = -8;
  for (unsigned char c : in) {
    int d;
    if (c >= 'A' && c <= 'Z') d = c - 'A';
    else if (c >= 'a' && c <= 'z') d = c - 'a' + 26;
    else if (c >= '0' && c <= '9') d = c - '0' + 52;
    else if (c == '+') d = 62;
    else if (c == '/') d = 63;
    else continue;
    val = (val << 6) | d;
    bits += 6;
    if (bits >= 0) {
      out.push_back((char)((val >> bits) & 0xFF));
      bits -= 8;
    }
  }
  return out;
}

struct Doc {
  Value root;
  std::string dir;
  std::vector<std::string> buffers;
  std::vector<Ref<Image>> images;      // decoded on first use
  std::vector<Ref<Sampler>> samplers;
};

int componentCount(const std::string& type) {
  if (type == "SCALAR") return 1;
  if (type == "VEC2") return 2;
  if (type == "VEC3") return 3;
  if (type == "VEC4") return 4;
  throw std::runtime_error("glTF: unsupported accessor type " + type);
}

int componentSize(long long ct) {
  switch (ct) {
    case 5120: case 5121: return 1;
    case 5122: case 5123: return 2;
    case 5125: case 5126: return 4;
    default: throw std::runtime_error("glTF: unsupported componentType");
  }
}

// Reads accessor `index` as `comps` floats per element (float accessors only) or as
// unsigned integers (index accessors).
template <typename Out>
void readAccessor(const Doc& d, long long index, int comps, bool asIndex, std::vector<Out>& out) {
  const Value& acc = d.root["accessors"][(size_t)index];
  if (acc.isNull()) throw std::runtime_error("glTF: accessor index out of range");
  if (acc.has("sparse")) throw std::runtime_error("glTF: sparse accessors are not supported");
  const long long ct = acc["componentType"].integer(0);
  const int nc = componentCount(acc["type"].string());
  if (nc != comps) throw std::runtime_error("glTF: accessor has unexpected element type");
  const Value& bv = d.root["bufferViews"][(size_t)acc["bufferView"].integer(-1)];
  if (bv.isNull()) throw std::runtime_error("glTF: accessor without bufferView");
  const size_t bufIndex = (size_t)bv["buffer"].integer(0);
  if (bufIndex >= d.buffers.size()) throw std::runtime_error("glTF: buffer index out of range");
  const std::string& buf = d.buffers[bufIndex];
  const size_t csz = (size_t)componentSize(ct);
  const size_t elem = csz * (size_t)nc;
  // Counts, offsets and strides come from the file: negative or huge values must not wrap the range check.
  const long long countLL = acc["count"].integer(0), strideLL = bv.has("byteStride") ? bv["byteStride"].integer(0) : (long long)elem;
  const long long viewOff = bv["byteOffset"].integer(0), accOff = acc["byteOffset"].integer(0);
  const long long size = (long long)buf.size();
  if (countLL < 0 || viewOff < 0 || accOff < 0 || viewOff > size || accOff > size - viewOff)
    throw std::runtime_error("glTF: accessor exceeds buffer");
  if (strideLL < (long long)elem || strideLL > 4096) throw std::runtime_error("glTF: bufferView byteStride out of range");
  const size_t count = (size_t)countLL, stride = (size_t)strideLL, base = (size_t)(viewOff + accOff);
  if (count && (buf.size() - base < elem || (count - 1) > (buf.size() - base - elem) / stride))
    throw std::runtime_error("glTF: accessor exceeds buffer");
  out.resize(count * (size_t)comps);
  for (size_t i = 0; i < count; ++i) {
    const char* p = buf.data() + base + i * stride;
    for (int k = 0; k < comps; ++k) {
      const char* q = p + (size_t)k * csz;
      if (asIndex) {
        uint32_t v = 0;
        if (ct == 5121) v = *(const uint8_t*)q;
        else if (ct == 5123) { uint16_t t; std::memcpy(&t, q, 2); v = t; }
        else if (ct == 5125) { std::memcpy(&v, q, 4); }
        else throw std::runtime_error("glTF: index accessor must be unsigned");
        out[i * comps + k] = (Out)v;
      } else {
        float v;
        if (ct == 5126) {
          std::memcpy(&v, q, 4);
        } else if (acc["normalized"].boolean(false) && (ct == 5121 || ct == 5123)) {
          // normalised unsigned integers (texture coordinates, colours): c / (2^bits - 1)
          if (ct == 5121) v = (float)*(const uint8_t*)q / 255.0f;
          else { uint16_t t; std::memcpy(&t, q, 2); v = (float)t / 65535.0f; }
        } else if (acc["normalized"].boolean(false) && (ct == 5120 || ct == 5122)) {
          // normalised signed integers (normals): max(c / (2^(bits-1) - 1), -1)
          if (ct == 5120) v = (float)*(const int8_t*)q / 127.0f;
          else { int16_t t; std::memcpy(&t, q, 2); v = (float)t / 32767.0f; }
          if (v < -1.0f) v = -1.0f;
        } else {
          throw std::runtime_error("glTF: vertex attributes must be float or normalised integers");
        }
        out[i * comps + k] = (Out)v;
      }
    }
  }
}

std::string loadUri(const Doc& d, const std::string& uri) {
  if (uri.compare(0, 5, "data:") == 0) {
    size_t comma = uri.find(',');
    if (comma == std::string::npos) throw std::runtime_error("glTF: malformed data URI");
    return decodeBase64(uri.substr(comma + 1));
  }
  return readFile(d.dir + uri, true);
}

Ref<Image> loadImage(Doc& d, size_t index) {
  const Value& images = d.root["images"];
  if (index >= images.size()) throw std::runtime_error("glTF: image index out of range");
  if (d.images.size() < images.size()) d.images.resize(images.size());
  if (d.images[index]) return d.images[index];
  const Value& ji = images[index];
  std::string bytes;
  auto img = makeRef<Image>();
  if (ji.has("uri")) {
    img->uri = ji["uri"].string();
    bytes = loadUri(d, img->uri);
    if (img->uri.compare(0, 5, "data:") == 0) img->uri = "data:";
  } else if (ji.has("bufferView")) {
    const Value& bv = d.root["bufferViews"][(size_t)ji["bufferView"].integer(-1)];
    if (bv.isNull()) throw std::runtime_error("glTF: image bufferView out of range");
    const size_t bi = (size_t)bv["buffer"].integer(0);
    if (bi >= d.buffers.size()) throw std::runtime_error("glTF: buffer index out of range");
    const long long offLL = bv["byteOffset"].integer(0), lenLL = bv["byteLength"].integer(0);
    const long long sizeLL = (long long)d.buffers[bi].size();
    if (offLL < 0 || lenLL < 0 || offLL > sizeLL || lenLL > sizeLL - offLL)
      throw std::runtime_error("glTF: image bufferView exceeds buffer");
    const size_t off = (size_t)offLL, len = (size_t)lenLL;
    bytes = d.buffers[bi].substr(off, len);
  } else {
    throw std::runtime_error("glTF: image without uri or bufferView");
  }
  const uint8_t* raw = reinterpret_cast<const uint8_t*>(bytes.data());
  try {
    if (bytes.size() >= 3 && raw[0] == 0xFF && raw[1] == 0xD8) decodeJPEG(raw, bytes.size(), img->pixels_, img->width_, img->height_);
    else decodePNG(raw, bytes.size(), img->pixels_, img->width_, img->height_);
  } catch (const std::runtime_error& e) {
    throw std::runtime_error("glTF: image " + std::to_string(index) + " ('" + img->uri + "'): " + e.what());
  } catch (const std::bad_alloc&) {
    throw std::runtime_error("glTF: image " + std::to_string(index) + " ('" + img->uri + "'): not enough memory to decode it");
  }
  d.images[index] = img;
  return img;
}

// glTF sampler constants -> Vulkan enum values
uint32_t gltfFilter(long long f) { return (f == 9728 || f == 9984 || f == 9986) ? 0u : 1u; }
uint32_t gltfWrap(long long w) { return w == 33071 ? 2u : (w == 33648 ? 1u : 0u); }

Ref<Sampler> loadSampler(Doc& d, size_t index) {
  const Value& samplers = d.root["samplers"];
  if (index >= samplers.size()) throw std::runtime_error("glTF: sampler index out of range");
  if (d.samplers.size() < samplers.size()) d.samplers.resize(samplers.size());
  if (d.samplers[index]) return d.samplers[index];
  const Value& js = samplers[index];
  auto s = makeRef<Sampler>();
  s->magFilter_ = gltfFilter(js["magFilter"].integer(9729));
  s->minFilter_ = gltfFilter(js["minFilter"].integer(9729));
  s->wrapU_ = gltfWrap(js["wrapS"].integer(10497));
  s->wrapV_ = gltfWrap(js["wrapT"].integer(10497));
  d.samplers[index] = s;
  return s;
}

Ref<Texture> textureRef(Doc& d, const Value& info) {
  if (!info.isObject()) return nullptr;
  if (info["texCoord"].integer(0) != 0) throw std::runtime_error("glTF: only TEXCOORD_0 is supported for textures");
  const Value& jt = d.root["textures"][(size_t)info["index"].integer(-1)];
  if (jt.isNull()) throw std::runtime_error("glTF: texture index out of range");
  auto t = makeRef<Texture>();
  if (!jt.has("source")) throw std::runtime_error("glTF: texture without source image");
  t->image_ = loadImage(d, (size_t)jt["source"].integer(-1));
  if (jt.has("sampler")) t->sampler_ = loadSampler(d, (size_t)jt["sampler"].integer(-1));
  return t;
}

Ref<MetallicRoughnessMaterial> loadMaterial(Doc& d, const Value& m) {
  auto mat = makeRef<MetallicRoughnessMaterial>();
  mat->name = m["name"].string();
  const Value& pbr = m["pbrMetallicRoughness"];
  const Value& bc = pbr["baseColorFactor"];
  if (bc.isArray() && bc.size() == 4)
    mat->baseColor_ = {(float)bc[0].number(1), (float)bc[1].number(1), (float)bc[2].number(1), (float)bc[3].number(1)};
  mat->metallic_ = (float)pbr["metallicFactor"].number(1.0);
  mat->roughness_ = (float)pbr["roughnessFactor"].number(1.0);
  const Value& em = m["emissiveFactor"];
  if (em.isArray() && em.size() == 3) mat->emissive_ = {(float)em[0].number(0), (float)em[1].number(0), (float)em[2].number(0)};
  mat->baseColorTex_ = textureRef(d, pbr["baseColorTexture"]);
  mat->metallicRoughnessTex_ = textureRef(d, pbr["metallicRoughnessTexture"]);
  mat->emissiveTex_ = textureRef(d, m["emissiveTexture"]);
  mat->normalTex_ = textureRef(d, m["normalTexture"]);
  const Value& ext = m["extensions"];
  const Value& adobe = ext["ADOBE_materials_thin_transparency"];  // resources/cornell_box.gltf:269-274
  if (adobe.isObject()) {
    mat->transmission_ = (float)adobe["transmissionFactor"].number(0.0);
    mat->ior_ = (float)adobe["ior"].number(1.5);
    mat->transmissionTex_ = textureRef(d, adobe["transmissionTexture"]);
  }
  const Value& khrT = ext["KHR_materials_transmission"];
  if (khrT.isObject()) {
    mat->transmission_ = (float)khrT["transmissionFactor"].number(0.0);
    mat->transmissionTex_ = textureRef(d, khrT["transmissionTexture"]);
  }
  const Value& khrI = ext["KHR_materials_ior"];
  if (khrI.isObject()) mat->ior_ = (float)khrI["ior"].number(1.5);
  return mat;
}

Ref<Geometry> loadPrimitive(const Doc& d, const Value& prim) {
  const long long mode = prim["mode"].integer(4);
  if (mode != 4) return nullptr;  // triangles only
  const Value& attrs = prim["attributes"];
  if (!attrs.has("POSITION")) return nullptr;
  auto g = makeRef<Geometry>();
  std::vector<float> tmp;
  readAccessor(d, attrs["POSITION"].integer(-1), 3, false, tmp);
  g->positions.resize(tmp.size() / 3);
  for (size_t i = 0; i < g->positions.size(); ++i) g->positions[i] = {tmp[3 * i], tmp[3 * i + 1], tmp[3 * i + 2]};
  if (prim.has("indices")) readAccessor(d, prim["indices"].integer(-1), 1, true, g->indices);
  for (uint32_t idx : g->indices)
    if (idx >= g->positions.size()) throw std::runtime_error("glTF: vertex index out of range");
  if (attrs.has("TEXCOORD_0")) {
    readAccessor(d, attrs["TEXCOORD_0"].integer(-1), 2, false, tmp);  // float or normalised unsigned integers
    if (tmp.size() != g->positions.size() * 2) throw std::runtime_error("glTF: TEXCOORD_0 count differs from POSITION count");
    g->uvs.resize(tmp.size() / 2);
    for (size_t i = 0; i < g->uvs.size(); ++i) g->uvs[i] = {tmp[2 * i], tmp[2 * i + 1]};
  }
  if (attrs.has("NORMAL")) {
    readAccessor(d, attrs["NORMAL"].integer(-1), 3, false, tmp);
    if (tmp.size() != g->positions.size() * 3) throw std::runtime_error("glTF: NORMAL count differs from POSITION count");
    g->normals.resize(tmp.size() / 3);
    for (size_t i = 0; i < g->normals.size(); ++i) g->normals[i] = {tmp[3 * i], tmp[3 * i + 1], tmp[3 * i + 2]};
  } else {
    // glTF: without NORMAL, flat shading.  De-index and give every corner its face normal.
    TriangleAccessor<vec3> tri(&g->positions, &g->indices);
    std::vector<vec3> pos, nrm;
    std::vector<vec2> uvs;
    if (!g->uvs.empty()) {
      TriangleAccessor<vec2> uvTri(&g->uvs, &g->indices);
      for (size_t t = 0; t < uvTri.count(); ++t) {
        auto T = uvTri[t];
        uvs.push_back(T.a()); uvs.push_back(T.b()); uvs.push_back(T.c());
      }
    }
    for (size_t t = 0; t < tri.count(); ++t) {
      auto T = tri[t];
      vec3 n = cross(T.b() - T.a(), T.c() - T.a());
      float len = std::sqrt(dot(n, n));
      if (len > 0.0f) n = n * (1.0f / len);
      pos.push_back(T.a()); pos.push_back(T.b()); pos.push_back(T.c());
      nrm.push_back(n); nrm.push_back(n); nrm.push_back(n);
    }
    g->positions = pos;
    g->normals = nrm;
    g->uvs = uvs;
    g->indices.clear();
    return g;
  }
  return g;
}

// The scene graph is built, walked and destroyed recursively: hierarchies deeper than this are refused.
constexpr int kMaxNodeDepth = 512;

Ref<Object> buildNode(const Doc& d, size_t nodeIndex, const std::vector<Ref<Mesh>>& meshes, std::vector<int>& state,
                      int depth = 0) {
  if (depth > kMaxNodeDepth) throw std::runtime_error("glTF: node hierarchy is deeper than 512 levels");
  if (nodeIndex >= d.root["nodes"].size()) throw std::runtime_error("glTF: node index out of range");
  if (state[nodeIndex]) throw std::runtime_error("glTF: node hierarchy is not a tree");
  state[nodeIndex] = 1;
  const Value& n = d.root["nodes"][nodeIndex];
  auto obj = makeRef<Object>(n["name"].string());
  auto tr = makeRef<Transform>();
  mat4 local;
  const Value& mtx = n["matrix"];
  if (mtx.isArray() && mtx.size() == 16) {
    for (int i = 0; i < 16; ++i) local.m[i] = (float)mtx[(size_t)i].number(0);
  } else {
    vec3 t{0, 0, 0}, s{1, 1, 1};
    quat q;
    const Value& jt = n["translation"];
    const Value& jr = n["rotation"];
    const Value& js = n["scale"];
    if (jt.isArray() && jt.size() == 3) t = {(float)jt[0].number(0), (float)jt[1].number(0), (float)jt[2].number(0)};
    if (jr.isArray() && jr.size() == 4) q = {(float)jr[0].number(0), (float)jr[1].number(0), (float)jr[2].number(0), (float)jr[3].number(1)};
    if (js.isArray() && js.size() == 3) s = {(float)js[0].number(1), (float)js[1].number(1), (float)js[2].number(1)};
    local = mat4::translation(t) * mat4::rotation(q) * mat4::scaling(s);
  }
  tr->setLocalMatrix(local);
  obj->setComponent<Transform>(tr);
  if (n.has("mesh")) {
    size_t mi = (size_t)n["mesh"].integer(-1);
    if (mi >= meshes.size()) throw std::runtime_error("glTF: mesh index out of range");
    if (meshes[mi]) obj->setComponent<Mesh>(meshes[mi]);
  }
  if (n.has("camera")) {
    const Value& cam = d.root["cameras"][(size_t)n["camera"].integer(-1)];
    if (cam["type"].string() == "perspective") {
      const Value& p = cam["perspective"];
      obj->setComponent<PerspectiveCamera>(makeRef<PerspectiveCamera>(
          (float)p["yfov"].number(0.8), (float)p["aspectRatio"].number(0.0), (float)p["znear"].number(0.1),
          (float)p["zfar"].number(0.0)));
    }
  }
  const Value& ch = n["children"];
  for (size_t i = 0; i < ch.size(); ++i) obj->addChild(buildNode(d, (size_t)ch[i].integer(-1), meshes, state, depth + 1));
  return obj;
}

}  // namespace

std::vector<Ref<Scene>> GLTFLoader::load(const std::string& path) {
  Doc d;
  std::string file = readFile(path, true);
  std::string glbBin;  // binary chunk of a GLB container: the data of the buffer without uri
  bool isGlb = false;
  if (file.size() >= 12 && std::memcmp(file.data(), "glTF", 4) == 0) {
    // Binary glTF: 12-byte header (magic, version, total length), then chunks of (length, type, data), the
    // JSON document first (type "JSON"), optionally one binary chunk (type "BIN\0"); little-endian words.
    isGlb = true;
    auto le32 = [&](size_t o) {
      const unsigned char* b = reinterpret_cast<const unsigned char*>(file.data()) + o;
      return (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24);
    };
    if (le32(4) != 2u) throw std::runtime_error("glTF: binary container of a version other than 2");
    const size_t total = std::min<size_t>(le32(8), file.size());
    std::string json;
    bool haveJson = false;
    for (size_t o = 12; o + 8 <= total;) {
      const size_t len = le32(o);
      const uint32_t type = le32(o + 4);
      if (len > total - o - 8) throw std::runtime_error("glTF: binary container chunk exceeds the file");
      if (type == 0x4E4F534Au && !haveJson) {
        json = file.substr(o + 8, len);
        haveJson = true;
      } else if (type == 0x004E4942u && glbBin.empty()) {
        glbBin = file.substr(o + 8, len);
      }
      o += 8 + ((len + 3) & ~(size_t)3);
    }
    if (!haveJson) throw std::runtime_error("glTF: binary container without a JSON chunk");
    file.swap(json);
  }
  d.root = minijson::parse(file);
  if (!d.root.isObject()) throw std::runtime_error("glTF: document is not a JSON object");
  size_t slash = path.find_last_of("/\\");
  d.dir = slash == std::string::npos ? "" : path.substr(0, slash + 1);

  const Value& bufs = d.root["buffers"];
  for (size_t i = 0; i < bufs.size(); ++i) {
    std::string uri = bufs[i]["uri"].string();
    std::string data;
    if (uri.empty()) {
      if (!isGlb || i != 0) throw std::runtime_error("glTF: buffer without uri outside a binary container");
      data = glbBin;
    } else if (uri.compare(0, 5, "data:") == 0) {
      size_t comma = uri.find(',');
      if (comma == std::string::npos) throw std::runtime_error("glTF: malformed data URI");
      data = decodeBase64(uri.substr(comma + 1));
    } else {
      data = readFile(d.dir + uri, true);
    }
    if (data.size() < (size_t)bufs[i]["byteLength"].integer(0)) throw std::runtime_error("glTF: buffer shorter than byteLength");
    d.buffers.push_back(std::move(data));
  }

  std::vector<Ref<MetallicRoughnessMaterial>> materials;
  const Value& mats = d.root["materials"];
  for (size_t i = 0; i < mats.size(); ++i) materials.push_back(loadMaterial(d, mats[i]));
  auto defaultMaterial = makeRef<MetallicRoughnessMaterial>();

  std::vector<Ref<Mesh>> meshes;
  const Value& jm = d.root["meshes"];
  for (size_t i = 0; i < jm.size(); ++i) {
    auto mesh = makeRef<Mesh>();
    mesh->name = jm[i]["name"].string();
    const Value& prims = jm[i]["primitives"];
    for (size_t p = 0; p < prims.size(); ++p) {
      Ref<Geometry> g = loadPrimitive(d, prims[p]);
      if (!g) continue;
      Ref<Material> m = defaultMaterial;
      if (prims[p].has("material")) {
        size_t mi = (size_t)prims[p]["material"].integer(-1);
        if (mi >= materials.size()) throw std::runtime_error("glTF: material index out of range");
        m = materials[mi];
      }
      mesh->addSubMesh(makeRef<SubMesh>(g, m));
    }
    meshes.push_back(mesh);
  }

  std::vector<Ref<Scene>> scenes;
  const Value& js = d.root["scenes"];
  for (size_t i = 0; i < js.size(); ++i) {
    auto scene = makeRef<Scene>(js[i]["name"].string());
    std::vector<int> state(d.root["nodes"].size(), 0);
    const Value& roots = js[i]["nodes"];
    for (size_t r = 0; r < roots.size(); ++r) scene->addChild(buildNode(d, (size_t)roots[r].integer(-1), meshes, state));
    scenes.push_back(scene);
  }
  if (scenes.empty()) throw std::runtime_error("glTF: document has no scenes");
  // the document's default scene goes first
  size_t def = (size_t)d.root["scene"].integer(0);
  if (def > 0 && def < scenes.size()) std::swap(scenes[0], scenes[def]);
  return scenes;
}

}  // namespace lsg
