// scene_loader.cpp — see scene_loader.h. Compiled with -ffp-contract=off: the float32 matrix arithmetic
// below is specified operation by operation (same order as rtamd/scenes.py).
#include "scene_loader.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <functional>
#include <map>
#include <stdexcept>

#include "json.h"
#include "jpeg.h"
#include "png.h"

namespace rthost {
namespace {

struct M4 {
    float m[16]; // column-major: m[c*4 + r]
};
M4 identity() {
    M4 r{};
    r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0f;
    return r;
}
M4 mul(const M4& a, const M4& b) { // ((a0r*bc0 + a1r*bc1) + a2r*bc2) + a3r*bc3
    M4 o{};
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) {
            float acc = a.m[0 * 4 + r] * b.m[c * 4 + 0];
            for (int k = 1; k < 4; ++k) acc = acc + a.m[k * 4 + r] * b.m[c * 4 + k];
            o.m[c * 4 + r] = acc;
        }
    return o;
}
M4 translate(const float t[3]) {
    M4 r = identity();
    r.m[12] = t[0], r.m[13] = t[1], r.m[14] = t[2];
    return r;
}
M4 scale(const float s[3]) {
    M4 r = identity();
    r.m[0] = s[0], r.m[5] = s[1], r.m[10] = s[2];
    return r;
}
M4 from_quat(const float q[4]) { // (x, y, z, w); glm::mat4_cast arithmetic
    const float x = q[0], y = q[1], z = q[2], w = q[3];
    const float qxx = x * x, qyy = y * y, qzz = z * z, qxz = x * z, qxy = x * y, qyz = y * z, qwx = w * x, qwy = w * y, qwz = w * z;
    M4 r = identity();
    r.m[0] = 1.0f - 2.0f * (qyy + qzz);
    r.m[1] = 2.0f * (qxy + qwz);
    r.m[2] = 2.0f * (qxz - qwy);
    r.m[4] = 2.0f * (qxy - qwz);
    r.m[5] = 1.0f - 2.0f * (qxx + qzz);
    r.m[6] = 2.0f * (qyz + qwx);
    r.m[8] = 2.0f * (qxz + qwy);
    r.m[9] = 2.0f * (qyz - qwx);
    r.m[10] = 1.0f - 2.0f * (qxx + qyy);
    return r;
}
// GeometryData::obj_to_world = transpose(inverse(mat3(global))) (src/scene.cpp:502)
void normal_matrix(const M4& g, float out[9]) {
    const float m00 = g.m[0], m01 = g.m[1], m02 = g.m[2];
    const float m10 = g.m[4], m11 = g.m[5], m12 = g.m[6];
    const float m20 = g.m[8], m21 = g.m[9], m22 = g.m[10];
    const float c00 = m11 * m22 - m21 * m12;
    const float c01 = m10 * m22 - m12 * m20;
    const float c02 = m10 * m21 - m11 * m20;
    const float det = (m00 * c00 - m01 * c01) + m02 * c02;
    const float inv_det = 1.0f / det;
    float inv[3][3]; // inv[c][r]
    inv[0][0] = c00 * inv_det;
    inv[0][1] = (-(m01 * m22 - m21 * m02)) * inv_det;
    inv[0][2] = (m01 * m12 - m11 * m02) * inv_det;
    inv[1][0] = (-c01) * inv_det;
    inv[1][1] = (m00 * m22 - m20 * m02) * inv_det;
    inv[1][2] = (-(m00 * m12 - m10 * m02)) * inv_det;
    inv[2][0] = c02 * inv_det;
    inv[2][1] = (-(m00 * m21 - m20 * m01)) * inv_det;
    inv[2][2] = (m00 * m11 - m10 * m01) * inv_det;
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) out[c * 3 + r] = inv[r][c];
}

struct Glb {
    Json js;
    std::vector<uint8_t> bin;
};

Glb read_glb(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("Failed to load .glTF : cannot open " + path);
    std::vector<uint8_t> b((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    auto u32 = [&](size_t o) { return (uint32_t)b[o] | ((uint32_t)b[o + 1] << 8) | ((uint32_t)b[o + 2] << 16) | ((uint32_t)b[o + 3] << 24); };
    if (b.size() < 20 || u32(0) != 0x46546C67u || u32(4) != 2u) throw std::runtime_error("Failed to load .glTF : not a GLB v2 file");
    const size_t total = std::min<size_t>(u32(8), b.size());
    Glb g;
    size_t off = 12;
    bool have_json = false;
    while (off + 8 <= total) {
        const uint32_t len = u32(off), type = u32(off + 4);
        if (off + 8 + (size_t)len > b.size()) throw std::runtime_error("Failed to load .glTF : truncated chunk");
        if (type == 0x4E4F534Au) {
            g.js = JsonParser(std::string((const char*)b.data() + off + 8, len)).parse();
            have_json = true;
        } else if (type == 0x004E4942u) {
            g.bin.assign(b.begin() + (long)off + 8, b.begin() + (long)off + 8 + len);
        }
        off += 8 + (size_t)len;
    }
    if (!have_json) throw std::runtime_error("Failed to load .glTF : no JSON chunk");
    return g;
}

size_t comp_size(int t) {
    switch (t) {
    case 5120: case 5121: return 1;
    case 5122: case 5123: return 2;
    case 5125: case 5126: return 4;
    default: throw std::runtime_error("glTF: unsupported component type " + std::to_string(t));
    }
}
int type_count(const std::string& t) {
    if (t == "SCALAR") return 1;
    if (t == "VEC2") return 2;
    if (t == "VEC3") return 3;
    if (t == "VEC4") return 4;
    if (t == "MAT4") return 16;
    throw std::runtime_error("glTF: unsupported accessor type " + t);
}

struct View {
    const uint8_t* base;
    size_t stride, count;
    int comp, ncomp;
};
View accessor(const Glb& g, int index) {
    const Json& acc = g.js.at("accessors").at((size_t)index);
    const Json& bv = g.js.at("bufferViews").at((size_t)acc.int_or("bufferView", 0));
    View v;
    v.comp = acc.int_or("componentType", 0);
    v.ncomp = type_count(acc.at("type").str);
    // every number is validated against the BIN chunk's size BEFORE it is cast, and the range check is written without
    // sums or products that could wrap around in size_t (count = 2^61 with stride 8 used to pass)
    const size_t size = g.bin.size();
    v.count = acc.size_or("count", 0, size);
    const size_t off_bv = bv.size_or("byteOffset", 0, size), off_acc = acc.size_or("byteOffset", 0, size);
    const size_t elem = comp_size(v.comp) * (size_t)v.ncomp;
    const size_t bs = bv.size_or("byteStride", 0, size);
    v.stride = bs ? bs : elem;
    if (off_bv > size - off_acc) throw std::runtime_error("glTF: accessor out of bounds");
    const size_t start = off_bv + off_acc;
    if (v.count && !(elem <= size && start <= size - elem && (v.count - 1) <= (size - start - elem) / v.stride))
        throw std::runtime_error("glTF: accessor out of bounds");
    v.base = g.bin.data() + start;
    return v;
}
void read_floats(const View& v, int ncomp, std::vector<float>& out) {
    if (v.comp != 5126 || v.ncomp != ncomp) throw std::runtime_error("glTF: POSITION/NORMAL/TEXCOORD_0 must be float");
    for (size_t i = 0; i < v.count; ++i) {
        float tmp[4];
        std::memcpy(tmp, v.base + i * v.stride, sizeof(float) * (size_t)ncomp);
        out.insert(out.end(), tmp, tmp + ncomp);
    }
}

} // namespace

rt_scene_desc LoadedScene::desc() const {
    rt_scene_desc d{};
    d.n_vertices = (uint32_t)(positions.size() / 3);
    d.positions = positions.data(), d.normals = normals.data(), d.uvs = uvs.data();
    d.n_triangles = (uint32_t)(indices.size() / 3);
    d.indices = indices.data(), d.tri_instance = tri_instance.data();
    d.n_instances = (uint32_t)instances.size(), d.instances = instances.data();
    d.n_materials = (uint32_t)materials.size(), d.materials = materials.data();
    d.n_layers = n_layers, d.textures = n_layers ? textures.data() : nullptr;
    d.sky[0] = sky[0], d.sky[1] = sky[1], d.sky[2] = sky[2];
    return d;
}

LoadedScene load_glb(const std::string& path, bool verbose) {
    const Glb g = read_glb(path);
    const Json& js = g.js;
    LoadedScene sc;
    {
        size_t slash = path.find_last_of("/\\"), dot = path.find_last_of('.');
        sc.name = path.substr(slash == std::string::npos ? 0 : slash + 1,
                              (dot == std::string::npos || (slash != std::string::npos && dot < slash) ? path.size() : dot) -
                                  (slash == std::string::npos ? 0 : slash + 1));
    }
    const Json empty_arr = [] { Json j; j.kind = Json::Array; return j; }();
    const Json empty_obj = [] { Json j; j.kind = Json::Object; return j; }();
    auto arr = [&](const char* k) -> const Json& { return js.has(k) ? js.at(k) : empty_arr; };

    // ---- images -> 512x512 layers (Scene::load_images src/scene.cpp:148-162, ImageManager :39-100) ----
    const Json& images = arr("images");
    if (verbose) std::printf("Loading %zu images\n", images.size());
    if (images.size() > RT_TEX_MAX_LAYERS) throw std::runtime_error("Too many images uploaded"); // :41-44
    sc.n_layers = (uint32_t)images.size();
    sc.textures.assign((size_t)sc.n_layers * RT_TEX_SIZE * RT_TEX_SIZE * 4, 255);
    for (size_t i = 0; i < images.size(); ++i) {
        const Json& im = images.at(i);
        if (!im.has("bufferView")) throw std::runtime_error("glTF: image " + std::to_string(i) + " is not embedded in the GLB");
        const Json& bv = js.at("bufferViews").at((size_t)im.int_or("bufferView", 0));
        const size_t off = bv.size_or("byteOffset", 0, g.bin.size()), len = bv.size_or("byteLength", 0, g.bin.size());
        if (!(len <= g.bin.size() && off <= g.bin.size() - len)) throw std::runtime_error("glTF: image buffer view out of bounds");
        uint32_t w = 0, h = 0;
        std::vector<uint8_t> rgba;
        decode_image_rgba8(g.bin.data() + off, len, w, h, rgba);
        resize_to_layer(rgba.data(), w, h, sc.textures.data() + i * RT_TEX_SIZE * RT_TEX_SIZE * 4);
        if (verbose) std::printf("Resized image %zu from %ux%u to %ux%u\n", i, w, h, RT_TEX_SIZE, RT_TEX_SIZE);
    }
    if (verbose) std::printf("Baked %zu images into array\n", images.size());

    // ---- materials: classification of src/scene.cpp:188-254, created in order of first use ----
    const Json& gl_mats = arr("materials");
    std::map<int, uint32_t> mat_ids;
    auto material_for = [&](int idx) -> uint32_t {
        auto it = mat_ids.find(idx);
        if (it != mat_ids.end()) return it->second;
        const Json& gm = idx >= 0 ? gl_mats.at((size_t)idx) : empty_obj; // no material: glTF defaults (tiny_gltf.h:739-743)
        const Json& pbr = gm.has("pbrMetallicRoughness") ? gm.at("pbrMetallicRoughness") : empty_obj;
        double base[3] = {1.0, 1.0, 1.0};
        if (pbr.has("baseColorFactor"))
            for (int k = 0; k < 3; ++k) base[k] = pbr.at("baseColorFactor").at((size_t)k).num;
        const double metallic = pbr.number_or("metallicFactor", 1.0), rough = pbr.number_or("roughnessFactor", 1.0);
        const Json& ext = gm.has("extensions") ? gm.at("extensions") : empty_obj;
        double strength = 0.0; // emissive strength defaults to 0 unless the extension is present (:198-206)
        if (ext.has("KHR_materials_emissive_strength")) strength = ext.at("KHR_materials_emissive_strength").number_or("emissiveStrength", 0.0);
        double ef[3] = {0.0, 0.0, 0.0};
        if (gm.has("emissiveFactor"))
            for (int k = 0; k < 3; ++k) ef[k] = gm.at("emissiveFactor").at((size_t)k).num;
        bool has_tex = false;
        uint32_t tex_layer = 0;
        if (pbr.has("baseColorTexture")) {
            const int ti = pbr.at("baseColorTexture").int_or("index", -1);
            if (ti >= 0) {
                has_tex = true;
                tex_layer = (uint32_t)js.at("textures").at((size_t)ti).int_or("source", 0);
                if (tex_layer >= sc.n_layers) throw std::runtime_error("glTF: texture source out of range");
            }
        }
        rt_material m{};
        if (ext.has("KHR_materials_ior") && ext.has("KHR_materials_transmission")) {
            m.type = RT_MAT_DIELECTRIC, m.tex_kind = RT_TEX_COLOR;
            m.color[0] = m.color[1] = m.color[2] = 0.8f;
            m.roughness = 0.0f;
            m.ior = (float)ext.at("KHR_materials_ior").number_or("ior", 1.5);
            if (verbose) std::printf("Dielectric: ior=%g\n", m.ior);
        } else {
            m.type = metallic > (double)0.01f ? RT_MAT_METALLIC : RT_MAT_DIFFUSE; // double vs float-literal compare, as src/scene.cpp:219
            m.tex_kind = has_tex ? RT_TEX_IMAGE : RT_TEX_COLOR;
            m.tex_layer = has_tex ? tex_layer : 0u;
            for (int k = 0; k < 3; ++k) m.color[k] = (float)base[k], m.emissive[k] = (float)ef[k] * (float)strength;
            m.roughness = m.type == RT_MAT_METALLIC ? (float)rough : 0.0f;
            m.ior = 1.5f;
            if (verbose)
                std::printf(m.type == RT_MAT_METALLIC ? "Metallic: roughness=%g\n" : "Diffuse: albedo=(%g, %g, %g)\n",
                            m.type == RT_MAT_METALLIC ? m.roughness : m.color[0], m.color[1], m.color[2]);
        }
        const uint32_t id = (uint32_t)sc.materials.size();
        sc.materials.push_back(m);
        mat_ids[idx] = id;
        return id;
    };

    // ---- meshes: vertex data appended once per primitive (Scene::load_primitives :164-442) ----
    struct Prim {
        uint32_t vertex_base;
        std::vector<uint32_t> indices;
        uint32_t material;
    };
    std::vector<std::vector<Prim>> meshes;
    const Json& gl_meshes = arr("meshes");
    for (size_t mi = 0; mi < gl_meshes.size(); ++mi) {
        std::vector<Prim> prims;
        const Json& pl = gl_meshes.at(mi).at("primitives");
        for (size_t pi = 0; pi < pl.size(); ++pi) {
            const Json& p = pl.at(pi);
            const Json& at = p.at("attributes");
            if (!at.has("POSITION") || !at.has("NORMAL") || !at.has("TEXCOORD_0") || !p.has("indices"))
                throw std::runtime_error("glTF: primitives need POSITION, NORMAL, TEXCOORD_0 and indices"); // asserts :256-276
            Prim pr;
            pr.vertex_base = (uint32_t)(sc.positions.size() / 3);
            const View vp = accessor(g, at.int_or("POSITION", 0)), vn = accessor(g, at.int_or("NORMAL", 0)), vt = accessor(g, at.int_or("TEXCOORD_0", 0));
            if (vn.count != vp.count || vt.count != vp.count) throw std::runtime_error("glTF: attribute counts differ");
            read_floats(vp, 3, sc.positions), read_floats(vn, 3, sc.normals), read_floats(vt, 2, sc.uvs);
            const View vi = accessor(g, p.int_or("indices", 0));
            if (vi.ncomp != 1 || vi.count % 3) throw std::runtime_error("glTF: indices must be a SCALAR triangle list");
            pr.indices.resize(vi.count);
            for (size_t k = 0; k < vi.count; ++k) {
                const uint8_t* s = vi.base + k * vi.stride;
                uint32_t v;
                if (vi.comp == 5125) std::memcpy(&v, s, 4);
                else if (vi.comp == 5123) { uint16_t t; std::memcpy(&t, s, 2); v = t; }
                else if (vi.comp == 5121) v = *s;
                else throw std::runtime_error("Index component type " + std::to_string(vi.comp) + " not supported!");
                if (v >= vp.count) throw std::runtime_error("glTF: vertex index out of range");
                pr.indices[k] = v;
            }
            pr.material = material_for(p.int_or("material", -1));
            prims.push_back(std::move(pr));
        }
        meshes.push_back(std::move(prims));
    }

    // ---- scene extras: sky (src/scene.cpp:80-94) ----
    const Json& scenes = js.at("scenes");
    const int def = js.int_or("scene", -1);
    const Json& scene = scenes.at((size_t)(def > -1 ? def : 0));
    if (scene.has("extras")) {
        const Json& ex = scene.at("extras");
        if (ex.has("sky_color") && ex.at("sky_color").is_array() && ex.at("sky_color").size() == 3)
            for (int k = 0; k < 3; ++k) sc.sky[k] = (float)ex.at("sky_color").at((size_t)k).num;
        if (ex.has("sky_strength") && ex.at("sky_strength").is_number())
            for (int k = 0; k < 3; ++k) sc.sky[k] = sc.sky[k] * (float)ex.at("sky_strength").num;
    }

    // ---- node hierarchy (Scene::load_node :444-510, Scene::node_global_matrix :137-146) ----
    // local = T * R * S * matrix (:18-21). A node's GLOBAL matrix is composed exactly as the reference composes it, from the node UP:
    //     m = local * scale(global_scale);   for every ancestor, nearest first:  m = ancestor.local * m
    // (global_scale is the reference's (1, 1, 1) default, :88.) Matrix products do not associate in fp32: composing from the root down —
    // (root * parent) * local, as rounds 1-4 did — gives other low bits in every transform and normal matrix of a hierarchy three or more deep.
    const Json& nodes = arr("nodes");
    struct NodeRec {
        bool loaded = false;
        int parent = -1;
        M4 local;
    };
    std::vector<NodeRec> recs(nodes.size());
    struct Placed {
        size_t node;
        int mesh;
        M4 global;
    };
    std::vector<Placed> placed;
    int cam_node = -1;
    auto global_of = [&](size_t ni) {
        const float ones[3] = {1.0f, 1.0f, 1.0f};
        M4 m = mul(recs[ni].local, scale(ones));
        int depth = 0;
        for (int p = recs[ni].parent; p >= 0; p = recs[(size_t)p].parent) {
            if (++depth > 256) throw std::runtime_error("glTF: node hierarchy too deep (cycle?)");
            m = mul(recs[(size_t)p].local, m);
        }
        return m;
    };
    std::function<void(size_t, int, int)> visit = [&](size_t ni, int parent, int depth) {
        if (depth > 256) throw std::runtime_error("glTF: node hierarchy too deep (cycle?)");
        const Json& n = nodes.at(ni);
        float t[3] = {0, 0, 0}, q[4] = {0, 0, 0, 1}, s[3] = {1, 1, 1};
        if (n.has("translation") && n.at("translation").size() == 3) for (int k = 0; k < 3; ++k) t[k] = (float)n.at("translation").at((size_t)k).num;
        if (n.has("rotation") && n.at("rotation").size() == 4) for (int k = 0; k < 4; ++k) q[k] = (float)n.at("rotation").at((size_t)k).num;
        if (n.has("scale") && n.at("scale").size() == 3) for (int k = 0; k < 3; ++k) s[k] = (float)n.at("scale").at((size_t)k).num;
        M4 mm = identity();
        if (n.has("matrix") && n.at("matrix").size() == 16)
            for (int k = 0; k < 16; ++k) mm.m[k] = (float)n.at("matrix").at((size_t)k).num;
        recs[ni].local = mul(mul(mul(translate(t), from_quat(q)), scale(s)), mm); // ((T * R) * S) * matrix, the identity included (:18-21)
        recs[ni].parent = parent, recs[ni].loaded = true;
        if (n.has("camera")) cam_node = (int)ni;
        if (n.has("children"))
            for (size_t c = 0; c < n.at("children").size(); ++c) visit(n.at("children").at(c).index_value(nodes.size()), (int)ni, depth + 1);
        if (n.has("mesh")) placed.push_back({ni, n.int_or("mesh", 0), global_of(ni)}); // after the children, with the ancestors' locals already set (:472-492)
    };
    if (scene.has("nodes"))
        for (size_t r = 0; r < scene.at("nodes").size(); ++r) visit(scene.at("nodes").at(r).index_value(nodes.size()), -1, 0);
    // instances attach in NODE INDEX order (src/scene.cpp:101-106), primitives in order
    std::stable_sort(placed.begin(), placed.end(), [](const Placed& a, const Placed& b) { return a.node < b.node; });
    for (const Placed& pl : placed) {
        if (pl.mesh < 0 || (size_t)pl.mesh >= meshes.size()) throw std::runtime_error("glTF: node mesh out of range");
        for (const Prim& pr : meshes[(size_t)pl.mesh]) {
            rt_instance inst{};
            std::memcpy(inst.transform, pl.global.m, sizeof(inst.transform));
            normal_matrix(pl.global, inst.normal_mat);
            inst.material = pr.material;
            const uint32_t id = (uint32_t)sc.instances.size();
            sc.instances.push_back(inst);
            for (uint32_t v : pr.indices) sc.indices.push_back(v + pr.vertex_base);
            sc.tri_instance.insert(sc.tri_instance.end(), pr.indices.size() / 3, id);
        }
    }

    // ---- camera (src/scene.cpp:109-128). The reference reads uninitialised fields when the file has no
    // camera (Appendix A F6); the two shipped assets get the documented defaults instead. ----
    if (cam_node >= 0) {
        const Json& cn = nodes.at((size_t)cam_node);
        // position = the global matrix's translation column; direction = normalize(quat_cast(global) * (0, 0, -1)) (:113-121): the rotation is
        // taken out of the matrix by GLM's quat_cast (largest-component form), the forward vector is turned by q * v = v + 2 (w (q x v) + q x (q x v))
        // and normalised as glm::normalize does (v * 1 / sqrt(v . v)). Under scale or shear in the camera's chain this is NOT minus the matrix's
        // third column normalised (rounds 1-4); glm itself is un-vendored: its published algorithms are restated, in float.
        const M4 g = global_of((size_t)cam_node);
        auto M = [&](int c, int r) { return g.m[c * 4 + r]; };
        const float fx = M(0, 0) - M(1, 1) - M(2, 2), fy = M(1, 1) - M(0, 0) - M(2, 2), fz = M(2, 2) - M(0, 0) - M(1, 1), fw = M(0, 0) + M(1, 1) + M(2, 2);
        int biggest = 0;
        float four = fw;
        if (fx > four) four = fx, biggest = 1;
        if (fy > four) four = fy, biggest = 2;
        if (fz > four) four = fz, biggest = 3;
        const float big = std::sqrt(four + 1.0f) * 0.5f, mult = 0.25f / big;
        float qw, qx, qy, qz;
        switch (biggest) {
        case 0: qw = big, qx = (M(1, 2) - M(2, 1)) * mult, qy = (M(2, 0) - M(0, 2)) * mult, qz = (M(0, 1) - M(1, 0)) * mult; break;
        case 1: qw = (M(1, 2) - M(2, 1)) * mult, qx = big, qy = (M(0, 1) + M(1, 0)) * mult, qz = (M(2, 0) + M(0, 2)) * mult; break;
        case 2: qw = (M(2, 0) - M(0, 2)) * mult, qx = (M(0, 1) + M(1, 0)) * mult, qy = big, qz = (M(1, 2) + M(2, 1)) * mult; break;
        default: qw = (M(0, 1) - M(1, 0)) * mult, qx = (M(2, 0) + M(0, 2)) * mult, qy = (M(1, 2) + M(2, 1)) * mult, qz = big; break;
        }
        const float v[3] = {0.0f, 0.0f, -1.0f};
        const float uv[3] = {qy * v[2] - v[1] * qz, qz * v[0] - v[2] * qx, qx * v[1] - v[0] * qy};           // cross(q.xyz, v)
        const float uuv[3] = {qy * uv[2] - uv[1] * qz, qz * uv[0] - uv[2] * qx, qx * uv[1] - uv[0] * qy};   // cross(q.xyz, uv)
        float d[3];
        for (int k = 0; k < 3; ++k) d[k] = v[k] + ((uv[k] * qw) + uuv[k]) * 2.0f;
        const float inv = 1.0f / std::sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
        for (int k = 0; k < 3; ++k) sc.camera_direction[k] = d[k] * inv, sc.camera_position[k] = g.m[12 + k];
        const double yfov = js.at("cameras").at((size_t)cn.int_or("camera", 0)).at("perspective").number_or("yfov", 0.8);
        sc.camera_focal_length = (float)(1.0 / std::tan(yfov / 2.0));
        sc.has_camera = true;
    } else if (sc.name == "triangle") {
        const float p[3] = {-3.9f, 0.244f, 0.218f}, d[3] = {1.0f, 0.0f, 0.0f};
        std::memcpy(sc.camera_position, p, 12), std::memcpy(sc.camera_direction, d, 12);
        sc.camera_focal_length = 1.0f;
    } else { // "cube" and anything else without a camera: eye at the origin looking down -z
        const float p[3] = {0.0f, 0.0f, 0.0f}, d[3] = {0.0f, 0.0f, -1.0f};
        std::memcpy(sc.camera_position, p, 12), std::memcpy(sc.camera_direction, d, 12);
        sc.camera_focal_length = 1.0f;
    }
    return sc;
}

} // namespace rthost
