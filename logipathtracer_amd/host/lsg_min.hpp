// Minimal scene graph standing in for the LogiSceneGraph ("lsg") library the
// reference links against (libs/LogiSceneGraph is an empty submodule there).
// Only what RendererPT / PTSceneConverter / Main.cpp of the reference touch is
// provided, under the same names: Scene, Object, Transform, PerspectiveCamera,
// Mesh, SubMesh, Geometry triangle accessors, MetallicRoughnessMaterial,
// GLTFLoader, and the bvh builders (lsg_bvh.hpp).
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <functional>
#include <memory>
#include <string>
#include <typeindex>
#include <unordered_map>
#include <vector>

namespace lsg {

template <typename T>
using Ref = std::shared_ptr<T>;
template <typename T, typename... Args>
Ref<T> makeRef(Args&&... args) {
  return std::make_shared<T>(std::forward<Args>(args)...);
}
template <typename To, typename From>
Ref<To> dynamicRefCast(const Ref<From>& r) {
  return std::dynamic_pointer_cast<To>(r);
}

// ---- small linear algebra (column-major, single precision) ----------------------

struct vec2 { float x = 0, y = 0; };
struct vec3 {
  float x = 0, y = 0, z = 0;
  float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
struct vec4 { float x = 0, y = 0, z = 0, w = 0; };
struct quat { float x = 0, y = 0, z = 0, w = 1; };

inline vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3 operator*(vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline vec3 cross(vec3 a, vec3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

struct mat4 {
  float m[16];  // m[col*4+row]
  mat4() : m{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1} {}
  float& at(int col, int row) { return m[col * 4 + row]; }
  float at(int col, int row) const { return m[col * 4 + row]; }
  static mat4 identity() { return mat4(); }
  static mat4 translation(vec3 t);
  static mat4 scaling(vec3 s);
  static mat4 rotation(quat q);
  static mat4 rotationAxis(int axis, float radians);
};
mat4 operator*(const mat4& a, const mat4& b);
mat4 inverse(const mat4& a);
vec3 transformPoint(const mat4& m, vec3 p);

template <typename T>
class AABB {
 public:
  AABB() { reset(); }
  AABB(vec3 mn, vec3 mx) : min_(mn), max_(mx) {}
  void reset() {
    min_ = {INFINITY, INFINITY, INFINITY};
    max_ = {-INFINITY, -INFINITY, -INFINITY};
  }
  void expand(vec3 p) {
    min_ = {std::fmin(min_.x, p.x), std::fmin(min_.y, p.y), std::fmin(min_.z, p.z)};
    max_ = {std::fmax(max_.x, p.x), std::fmax(max_.y, p.y), std::fmax(max_.z, p.z)};
  }
  // (component-wise, so that an empty box — min = +inf, max = -inf — leaves the other unchanged; expanding by its two
  // corners as points would turn the result into the whole space)
  void expand(const AABB& o) {
    min_ = {std::fmin(min_.x, o.min_.x), std::fmin(min_.y, o.min_.y), std::fmin(min_.z, o.min_.z)};
    max_ = {std::fmax(max_.x, o.max_.x), std::fmax(max_.y, o.max_.y), std::fmax(max_.z, o.max_.z)};
  }
  const vec3& min() const { return min_; }
  const vec3& max() const { return max_; }
  vec3 center() const { return (min_ + max_) * 0.5f; }
  float area() const {
    vec3 d = max_ - min_;
    if (d.x < 0 || d.y < 0 || d.z < 0) return 0.0f;
    return 2.0f * (d.x * d.y + d.y * d.z + d.z * d.x);
  }
  // Box of the eight transformed corners (src/PTSceneConverter.cpp:122).
  AABB transform(const mat4& m) const {
    AABB out;
    for (int i = 0; i < 8; ++i)
      out.expand(transformPoint(m, {(i & 1) ? max_.x : min_.x, (i & 2) ? max_.y : min_.y, (i & 4) ? max_.z : min_.z}));
    return out;
  }

 private:
  vec3 min_, max_;
};

// ---- components ---------------------------------------------------------------------

class Object;

class Component {
 public:
  virtual ~Component() = default;
};

class Transform : public Component {
 public:
  void setLocalMatrix(const mat4& m) { local_ = m; touch(); }
  const mat4& localMatrix() const { return local_; }
  void setParent(const std::weak_ptr<Transform>& p) { parent_ = p; touch(); }
  // World matrix = parent's world matrix * local matrix.  Reading it clears the dirty state.
  mat4 worldMatrix();
  // True if this transform or an ancestor changed since worldMatrix() was last read
  // (src/RendererPT.cpp:575).
  bool isWorldMatrixDirty() const { return chainVersion() != seenVersion_; }
  // Moves / turns in the object's own axes (src/Main.cpp:57-93).
  void translateX(float d) { local_ = local_ * mat4::translation({d, 0, 0}); touch(); }
  void translateY(float d) { local_ = local_ * mat4::translation({0, d, 0}); touch(); }
  void translateZ(float d) { local_ = local_ * mat4::translation({0, 0, d}); touch(); }
  void rotateX(float a) { local_ = local_ * mat4::rotationAxis(0, a); touch(); }
  void rotateY(float a) { local_ = local_ * mat4::rotationAxis(1, a); touch(); }
  void rotateZ(float a) { local_ = local_ * mat4::rotationAxis(2, a); touch(); }

 private:
  void touch() { ++version_; }
  uint64_t chainVersion() const;
  mat4 local_;
  std::weak_ptr<Transform> parent_;
  uint64_t version_ = 1;
  uint64_t seenVersion_ = 0;
};

class PerspectiveCamera : public Component {
 public:
  PerspectiveCamera(float fovY, float aspect, float znear, float zfar)
      : fov_(fovY), aspect_(aspect), near_(znear), far_(zfar) {}
  float fov() const { return fov_; }  // vertical field of view, radians (glTF yfov)
  float aspectRatio() const { return aspect_; }
  float nearPlane() const { return near_; }
  float farPlane() const { return far_; }

 private:
  float fov_, aspect_, near_, far_;
};

// Decoded texture image, always RGBA8 (what the reference uploads as one mip level,
// src/PTSceneConverter.cpp:248-283).
class Image {
 public:
  uint32_t width() const { return width_; }
  uint32_t height() const { return height_; }
  size_t pixelSize() const { return 4; }
  const void* rawPixelData() const { return pixels_.data(); }

  std::string uri;
  uint32_t width_ = 0, height_ = 0;
  std::vector<uint8_t> pixels_;  // width*height*4, row 0 first
};

// Sampler state with Vulkan enum values (vk::Filter: 0 nearest, 1 linear;
// vk::SamplerAddressMode: 0 repeat, 1 mirrored repeat, 2 clamp to edge).  glTF only
// describes two wrap axes; the third keeps its default.
class Sampler {
 public:
  uint32_t magFilter() const { return magFilter_; }
  uint32_t minFilter() const { return minFilter_; }
  uint32_t wrappingU() const { return wrapU_; }
  uint32_t wrappingV() const { return wrapV_; }
  uint32_t wrappingW() const { return wrapW_; }

  uint32_t magFilter_ = 1, minFilter_ = 1;
  uint32_t wrapU_ = 0, wrapV_ = 0, wrapW_ = 0;
};

class Texture {
 public:
  Ref<Image> image() const { return image_; }
  Ref<Sampler> sampler() const { return sampler_; }
  Ref<Image> image_;
  Ref<Sampler> sampler_;  // may be null (texture without "sampler")
};

class Material {
 public:
  virtual ~Material() = default;
  std::string name;
};

// glTF 2.0 pbrMetallicRoughness plus thin-transparency / transmission extensions.
class MetallicRoughnessMaterial : public Material {
 public:
  const vec4& baseColorFactor() const { return baseColor_; }
  const vec3& emissiveFactor() const { return emissive_; }
  float metallicFactor() const { return metallic_; }
  float roughnessFactor() const { return roughness_; }
  float transmissionFactor() const { return transmission_; }
  float ior() const { return ior_; }
  Ref<Texture> baseColorTex() const { return baseColorTex_; }
  Ref<Texture> emissiveTex() const { return emissiveTex_; }
  Ref<Texture> metallicRoughnessTex() const { return metallicRoughnessTex_; }
  Ref<Texture> transmissionTexture() const { return transmissionTex_; }
  Ref<Texture> normalTex() const { return normalTex_; }

  vec4 baseColor_{1, 1, 1, 1};
  vec3 emissive_{0, 0, 0};
  float metallic_ = 1.0f;   // glTF defaults
  float roughness_ = 1.0f;
  float transmission_ = 0.0f;
  float ior_ = 1.5f;
  Ref<Texture> baseColorTex_, emissiveTex_, metallicRoughnessTex_, transmissionTex_, normalTex_;
};

template <typename V>
class Triangle {
 public:
  Triangle(V a, V b, V c) : v_{a, b, c} {}
  const V& a() const { return v_[0]; }
  const V& b() const { return v_[1]; }
  const V& c() const { return v_[2]; }

 private:
  V v_[3];
};

// Triangle view over an attribute array, optionally through an index buffer.
template <typename V>
class TriangleAccessor {
 public:
  TriangleAccessor(const std::vector<V>* data, const std::vector<uint32_t>* indices) : data_(data), indices_(indices) {}
  size_t count() const { return (indices_ && !indices_->empty() ? indices_->size() : data_->size()) / 3; }
  Triangle<V> operator[](size_t t) const {
    if (indices_ && !indices_->empty())
      return {(*data_)[(*indices_)[3 * t]], (*data_)[(*indices_)[3 * t + 1]], (*data_)[(*indices_)[3 * t + 2]]};
    return {(*data_)[3 * t], (*data_)[3 * t + 1], (*data_)[3 * t + 2]};
  }

 private:
  const std::vector<V>* data_;
  const std::vector<uint32_t>* indices_;
};

class Geometry {
 public:
  std::vector<vec3> positions;
  std::vector<vec3> normals;
  std::vector<vec2> uvs;  // TEXCOORD_0, may be empty
  std::vector<uint32_t> indices;

  bool hasUv(unsigned set) const { return set == 0 && !uvs.empty(); }
  Ref<TriangleAccessor<vec3>> getTrianglePositionAccessor() const {
    return makeRef<TriangleAccessor<vec3>>(&positions, &indices);
  }
  Ref<TriangleAccessor<vec3>> getTriangleNormalAccessor() const {
    return makeRef<TriangleAccessor<vec3>>(&normals, &indices);
  }
  Ref<TriangleAccessor<vec2>> getTriangleUVAccessor(unsigned) const {
    return makeRef<TriangleAccessor<vec2>>(&uvs, &indices);
  }
};

class SubMesh {
 public:
  SubMesh(Ref<Geometry> g, Ref<Material> m) : geometry_(std::move(g)), material_(std::move(m)) {}
  const Ref<Geometry>& geometry() const { return geometry_; }
  const Ref<Material>& material() const { return material_; }

 private:
  Ref<Geometry> geometry_;
  Ref<Material> material_;
};

class Mesh : public Component {
 public:
  const std::vector<Ref<SubMesh>>& subMeshes() const { return subMeshes_; }
  void addSubMesh(Ref<SubMesh> s) { subMeshes_.push_back(std::move(s)); }
  std::string name;

 private:
  std::vector<Ref<SubMesh>> subMeshes_;
};

// ---- objects -------------------------------------------------------------------------

class Object : public std::enable_shared_from_this<Object> {
 public:
  explicit Object(std::string name = "") : name_(std::move(name)) {}
  const std::string& name() const { return name_; }
  const std::vector<Ref<Object>>& children() const { return children_; }
  void addChild(const Ref<Object>& c);

  template <typename T>
  Ref<T> getComponent() const {
    auto it = components_.find(std::type_index(typeid(T)));
    return it == components_.end() ? nullptr : std::static_pointer_cast<T>(it->second);
  }
  template <typename T>
  void setComponent(Ref<T> c) {
    components_[std::type_index(typeid(T))] = std::move(c);
  }
  // Pre-order walk; the callback returns false to stop descending below a node.
  void traverseDown(const std::function<bool(const Ref<Object>&)>& fn);

 private:
  std::string name_;
  std::vector<Ref<Object>> children_;
  std::unordered_map<std::type_index, Ref<Component>> components_;
};

class Scene {
 public:
  explicit Scene(std::string name = "") : name_(std::move(name)) {}
  const std::string& name() const { return name_; }
  const std::vector<Ref<Object>>& children() const { return children_; }
  void addChild(const Ref<Object>& c) { children_.push_back(c); }
  void traverseDown(const std::function<bool(const Ref<Object>&)>& fn) {
    for (auto& c : children_) c->traverseDown(fn);
  }

 private:
  std::string name_;
  std::vector<Ref<Object>> children_;
};

// glTF 2.0 (.gltf + external or base64 buffers) -> scenes.  Throws std::runtime_error
// on malformed or unsupported input (src/Main.cpp:16-17).
class GLTFLoader {
 public:
  std::vector<Ref<Scene>> load(const std::string& path);
};

}  // namespace lsg
