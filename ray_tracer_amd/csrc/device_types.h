// device_types.h -- layout of the flattened scene in HBM, shared by the host
// flattener (flatten.cpp) and the HIP kernels (mi355rt.hip).
//
// Every record that a traversal lane fetches on its own is a multiple of 16 bytes
// and 16-byte aligned, so one lane reads it with dwordx4 loads; BVH nodes are one
// 64-byte record (half a 128-byte line, never straddling one).
#ifndef RT_DEVICE_TYPES_H
#define RT_DEVICE_TYPES_H

#include <stdint.h>
#include "../../include/mi355rt.h"

namespace rt {

// Internal reference kinds that only live on the traversal stack.
constexpr uint32_t REF_QNODE = 7u;     // 4-wide node `index` of qnodes / cqnodes (ordered traversal only)
constexpr uint32_t REF_LISTCONT = 8u;  // resume list `index` at the position stored one entry below
constexpr uint32_t REF_POPX = 9u;      // leave a transform: restore context `index`
constexpr uint32_t REF_QUADS = 10u;    // a list that is nothing but 2..16 consecutive quads (the six sides of a box): index =
                                       // (count - 1) << 24 | first quad; stands in for the list in the 4-wide nodes, so the
                                       // walk needs neither the list header nor its items
constexpr uint32_t REF_LISTEND = 11u;  // (record-history walks only) the list entered below is done: settle whose u,v the enclosing record holds
constexpr uint32_t REF_OTHER = 15u;    // light-list member without pdf_value/random (hittable.rs:51-56 defaults)

constexpr int STACK_MAX = 32;          // traversal stack entries per lane (LDS)
constexpr int XFORM_DEPTH_MAX = 8;     // nested Translate/RotateY levels

inline constexpr uint32_t make_ref(uint32_t kind, uint32_t index) { return (kind << 28) | (index & 0x0FFFFFFFu); }

struct alignas(16) DSphere {   // 64 B (sphere.rs:12-17; center is the ray center1 + t*(center2-center1))
    double c[3];
    double dc[3];
    double radius;
    int32_t mat;
    int32_t _pad;
};

struct alignas(16) DQuad {     // 144 B (quad.rs:15-25 without the bbox)
    double q[3];
    double u[3];
    double v[3];
    double w[3];
    double n[3];               // unit normal
    double d;
    double area;
    int32_t mat;
    int32_t _pad;
};

// The three vertices the intersector reads (triangle.rs:113-115): 72 bytes in a 128-byte slot, so a
// triangle test fetches exactly one line from HBM (packed at 72 bytes every second record straddles two,
// and the primitive-BVH kernel is bound by the number of lines it pulls through the fabric).
// The unit normal and the material, read once per accepted hit, ride in the same line.
struct alignas(128) DTri {
    double v0[3];
    double v1[3];
    double v2[3];
    double n[3];               // unit normal (triangle.rs:125)
    int32_t mat;
    int32_t _pad0;
    double _pad[3];
};

typedef rt_bvh_node DNode;     // 64 B

// Wide node of the ordered traversal: the boxes of BOTH children of BVH node i, rounded outward
// to f32, so one 64-byte fetch decides both subtrees before either is entered.
struct alignas(16) WNode {     // 64 B
    float lb[6];               // left child box: xmin xmax ymin ymax zmin zmax
    float rb[6];               // right child box
    uint32_t left, right;      // same references as DNode
    uint32_t meta;             // bit 0: subtree holds a constant medium (keep reference order);
                               // bit 1: right is the same object as left (bvh.rs:41-44)
    uint32_t qroot;            // root of a BVH without media: 1 + index of its 4-wide root node, else 0
};

// 4-wide node of the primitive-BVH kernel: two levels of the binary tree collapsed into one
// 128-byte record (one cache line), boxes rounded outward to f32, structure-of-arrays per axis.
struct alignas(128) QNode {
    float lo[3][4];            // [axis][child] box minimum
    float hi[3][4];            // [axis][child] box maximum
    uint32_t child[4];         // REF_QNODE|q-index, any other reference (primitive, list, instance), or RT_REF_NONE (empty slot)
    uint32_t _pad[4];
};

// The same 4-wide node in 64 bytes, read with four 16-byte requests instead of seven.  Child boxes
// are 8-bit grid coordinates on the node's own grid: plane = origin[a] + q * 2^(exp[a] - 127), minima
// rounded down and maxima rounded up, so every stored box contains the f32 box it came from.
struct alignas(64) CQNode {
    float origin[3];           // lower corner of the union of the child boxes
    uint8_t exp[3];            // biased exponent of the grid step per axis
    uint8_t _pad0;
    uint32_t child[4];         // as QNode::child
    uint8_t qlo[3][4];         // [axis][child]
    uint8_t qhi[3][4];
    uint32_t _pad[2];
};

// The 4-wide tree of a triangle mesh once more, in 48 bytes per node (three 16-byte requests instead of four: the
// primitive-BVH kernel is bound by the number of divergent requests its lanes put to the vector L1).  The children
// of a node lie next to each other, so two base indices address them: inner children first, in slots
// 0 .. n_inner-1, at pnodes[node_base + slot]; then the triangles, at ptris[leaf_base + slot - n_inner].  Boxes as
// in CQNode.  The array is dense (the CQNode array is indexed like the binary tree and half empty).
struct alignas(16) PNode {
    float origin[3];
    uint8_t exp[3];
    uint8_t counts;            // low nibble: inner children, high nibble: all children
    uint8_t qlo[3][4];         // [axis][slot]
    uint8_t qhi[3][4];
    uint32_t node_base;        // REF_QNODE << 28 | index of the first inner child
    uint32_t leaf_base;        // (RT_REF_TRI << 28 | index of the first triangle child in ptris) - n_inner
};
static_assert(sizeof(PNode) == 48, "three 16-byte requests");

// Triangles in the order the packed tree's leaves name them; `orig` is the index in DScene::tris, which is what a
// hit reports.  One line per triangle, as DTri.
struct alignas(128) PTri {
    double v0[3];
    double v1[3];
    double v2[3];
    uint32_t orig;
    uint32_t _pad;
};

struct DList { uint32_t first, count; };

struct alignas(8) DXform {     // hittable.rs:61-65 / 97-102
    int32_t kind;              // 0 translate (a = offset), 1 rotate_y (a[0] = sin, a[1] = cos)
    int32_t parent;            // context id of the enclosing transform chain (0 = world)
    uint32_t child;
    int32_t depth;             // number of transforms from the world to here (>= 1)
    double a[3];
};

struct alignas(8) DMedium {    // constant_medium.rs:14-18
    uint32_t boundary;
    int32_t mat;
    double neg_inv_density;
    int32_t object;            // object id in the description (reported as the primitive index)
    int32_t _pad;
};

struct alignas(8) DMaterial {
    int32_t kind;
    int32_t tex;               // -1: the texture is a solid colour and `albedo` holds it (SolidColor::value, texture.rs:28-30)
    double albedo[3];
    double fuzz;
    double ior;
};

struct alignas(8) DTexture {
    int32_t kind, even, odd, image, perlin, _pad;
    double color[3];
    double scale;              // CHECKER: inv_scale = 1/scale (texture.rs:43); NOISE: scale
};

struct DLight { uint32_t kind; uint32_t index; }; // RT_REF_SPHERE / RT_REF_QUAD / REF_OTHER

// Pointers are device pointers once uploaded (host pointers in a HOST_ONLY scene).
struct DScene {
    const DNode* nodes;
    const WNode* wnodes;
    const QNode* qnodes;
    const CQNode* cqnodes;
    const PNode* pnodes;       // packed tree of the primitive-BVH kernel (null unless every leaf is a triangle)
    const PTri* ptris;
    const DSphere* spheres;
    const DQuad* quads;
    const DTri* tris;
    const DList* lists;
    const uint32_t* list_items;
    const DXform* xforms;
    const DMedium* media;
    const DMaterial* materials;
    const DTexture* textures;
    const rt_image* images;
    const uint8_t* image_bytes;
    const rt_perlin* perlins;
    const DLight* lights;
    uint32_t world;
    int32_t n_lights;          // -1: no light list (material pdf only)
    int32_t lights_is_list;    // 1: HittableList semantics (uniform pick / mean pdf), 0: a single object
    float q_extent;            // largest |coordinate| stored in the 4-wide nodes (margin of the f32 slab test)
    uint32_t uses_uv;          // some texture reads the surface coordinates u,v (an image texture exists)
    uint32_t world_rest;       // the world list without its first member (RT_REF_NONE unless that member is traced by the primitive-BVH kernel)
    uint32_t n_xforms;         // entries of `xforms`
    uint32_t n_perlins;        // entries of `perlins`
    const double* ebox;        // exact-cull tree: per 4-wide node from exact_first on and slot, the f64 box Aabb::hit is evaluated on (6 doubles)
    uint32_t exact_first;
};

} // namespace rt
#endif
