// flatten.h -- host flattener: object graph of the scene description -> linear arrays.
#ifndef RT_FLATTEN_H
#define RT_FLATTEN_H

#include <memory>
#include <new>
#include <string>
#include <utility>
#include <vector>
#include "device_types.h"

namespace rt {

struct Box { // aabb.rs:10-14 as xmin xmax ymin ymax zmin zmax
    double b[6];
};

// A vector whose resize() leaves trivial elements uninitialised: the big node arrays are written exactly once, by the
// threads that build them, and value-initialising hundreds of megabytes first costs a single-threaded pass over fresh pages.
template <class T>
struct DefaultInit : std::allocator<T> {
    template <class U> struct rebind { using other = DefaultInit<U>; };
    template <class U> void construct(U* p) noexcept { ::new (static_cast<void*>(p)) U; }
    template <class U, class... A> void construct(U* p, A&&... a) { ::new (static_cast<void*>(p)) U(std::forward<A>(a)...); }
};
template <class T> using RawVec = std::vector<T, DefaultInit<T>>;

struct BuildItem { Box box; uint32_t ref; uint32_t medium; }; // one object of a BVH under construction: bounding_box(), reference, holds-a-medium flag

struct FlatScene {
    std::vector<DNode> nodes;
    std::vector<WNode> wnodes;
    RawVec<QNode> qnodes;              // 4-wide collapse of the BVH at fast_root (empty otherwise)
    RawVec<PNode> pnodes;              // packed 48-byte copy of the tree at fast_root (node 0 is the root); empty unless all its leaves are triangles
    RawVec<PTri> ptris;               // its triangles in leaf order
    RawVec<CQNode> cqnodes;            // the same nodes with 8-bit boxes (same indices); empty when the grid would be too coarse
    std::vector<DSphere> spheres;
    std::vector<DQuad> quads;
    RawVec<DTri> tris;
    std::vector<DList> lists;
    std::vector<uint32_t> list_items;
    std::vector<DXform> xforms;
    std::vector<DMedium> media;
    std::vector<DMaterial> materials;
    std::vector<DTexture> textures;
    std::vector<rt_image> images;
    std::vector<uint8_t> image_bytes;
    std::vector<rt_perlin> perlins;
    std::vector<DLight> lights;
    std::vector<std::pair<uint32_t, uint32_t>> quad_runs; // (REF_QUADS reference used in the 4-wide nodes, the list it stands for)
    std::vector<Box> object_bounds; // bounding_box() of every object of the description
    uint32_t world = RT_REF_NONE;
    int32_t n_lights = -1;
    int32_t lights_is_list = 0;
    uint32_t max_stack = 0;
    uint32_t fast_root = RT_REF_NONE; // set when the world is one BVH whose leaves are all primitives
    uint32_t fast_stack = 0;          // stack entries the primitive-BVH kernel needs
    // RT_HIT_ORDERED | RT_HIT_EXACT_CULL on such a world: the collapse of the reference's tree (wide nodes exact_first ..
    // exact_first + exact_count of qnodes, entered at exact_root) and, per node and slot, the f64 box Aabb::hit is
    // evaluated on before that child is entered (ebox: 6 doubles per slot, 24 per node, indexed from exact_first)
    std::vector<double> ebox;
    uint32_t exact_root = RT_REF_NONE, exact_first = 0, exact_count = 0, exact_stack = 0;
    bool fast_tree = false;           // fast_root is a traversal-only tree (fast_tree.cpp): leaves may name a triangle more than once, with partial boxes
    bool fast_partial = false;        // fast_root covers only the FIRST member of the world list (rebuilt as one BVH over its
                                      // primitives); `world_rest` is the list of the remaining members, traversed afterwards
                                      // with the first member's closest hit as the interval's upper end (list order: hittable_list.rs:46-57)
    uint32_t world_rest = RT_REF_NONE;
    float q_extent = 0.0f;            // largest |coordinate| in qnodes
    double fast_leaf_overlap = 0.0;   // packed tree: sum of the leaf boxes' surface areas over the root's, i.e. the number of
                                      // triangle boxes a random line through the scene's box crosses
    double quant_inflation = 0.0;     // mean surface-area ratio of the 8-bit child boxes to their f32 boxes
    bool has_medium = false;
    bool track_uv = false;            // some constant medium's phase function reads u,v: traversal keeps the record history (traverse.h UvTrack)
    int medium_depth = 0;             // 2: some medium's boundary holds a medium (constant_medium.rs:48-104 takes any Hittable)
    bool uses_uv = false;             // an image texture exists: shading needs the spheres' (u,v)
    bool has_textures = false;        // some material's texture is not a solid colour (checker, image, noise)
    Box world_bounds;
};

// BvhNode::from_objects on the current HIP device (bvh_build.hip): same nodes as the host recursion, byte for byte.
int build_bvh_device(const BuildItem* items, uint64_t n, uint64_t base, DNode* nodes, WNode* wnodes, std::string& err);

// Returns RT_OK or a negative RT_E_* code with `err` set.  device_bvh_min > 0: BVHs over at least that many objects
// (none of them holding a constant medium) are built on the current HIP device.
// fast_tree_min > 0: a world that is one BVH over at least that many triangles (a Mesh) also gets a traversal-only tree
// for the nearest-first kernel (fast_tree.cpp); fast_root then names that tree.
int flatten_scene(const rt_scene_desc* desc, FlatScene& out, std::string& err, uint64_t device_bvh_min = 0, uint64_t fast_tree_min = 0);

// Camera::initialize (camera.rs:161-205).
int camera_initialize(const rt_camera* cam, rt_camera_derived* out, std::string& err);

} // namespace rt
#endif
