// m2s_host.h — internal host-side types of the scene I/O layer (.glb loader, PNG decoder, .ply reader).
// The public boundary is include/m2s.h.
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "../../include/m2s.h"

namespace m2s_host {

// std::vector without the zero fill of resize(): the loader's vertex arrays (hundreds of MB) are written completely, and in
// parallel, right after they are sized — a value-initialising resize would touch every page once more, on one thread.
template <class T>
struct DefaultInit : std::allocator<T> {
    template <class U> struct rebind { using other = DefaultInit<U>; };
    using std::allocator<T>::allocator;
    template <class U> void construct(U* p) noexcept { ::new (static_cast<void*>(p)) U; }
    template <class U, class... A> void construct(U* p, A&&... a) { ::new (static_cast<void*>(p)) U(std::forward<A>(a)...); }
};

struct Image {
    uint32_t width = 0, height = 0;
    std::vector<uint8_t> rgba;  // RGBA8, row 0 first
};

// PNG -> RGBA8 (tiny_gltf/stb_image behaviour: always 4 components).  false + err on failure.
bool decode_png(const uint8_t* data, size_t len, Image& img, std::string& err);
// JPEG (baseline / progressive Huffman, 8 bit, 1 or 3 components) -> RGBA8, bit-identical to stb_image's decoder.
bool decode_jpeg(const uint8_t* data, size_t len, Image& img, std::string& err);

// == utils::Mesh + its meshToTextureData entry after SceneManager::loadModel
struct HostMesh {
    std::string name;
    std::vector<float, DefaultInit<float>> vertices;   // 17 floats per vertex, the VBO of SceneManager::setupMeshBuffers
    float bbox_min[3], bbox_max[3];
    float base_color[4];
    int tex_image[3] = { -1, -1, -1 };  // index into HostScene::images (albedo, normal, MR)
};

struct HostScene {
    std::vector<HostMesh> meshes;
    std::vector<Image> images;         // decoded once, shared between materials
    std::vector<m2s_mesh> c_meshes;    // view handed to m2s_upload_scene
    std::vector<std::string> warnings;
};

// SceneManager::parseGltfFile + setupMeshBuffers + loadTextures for a .glb file.
bool load_glb(const std::string& path, HostScene& scene, std::string& err);

}  // namespace m2s_host
