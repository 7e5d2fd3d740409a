/*
 * mi355rt.h -- C ABI of libmi355rt.so: BVH traversal + closest-hit intersection
 * and the wavefront path tracer driven by the camera's per-pixel sampling loop,
 * executed by hand-written HIP kernels for gfx950 (MI355X).
 *
 * The reference (a single Rust crate, /root/reference/src) has no FFI surface;
 * its seam for this path is the trait `Hittable` (hittable.rs:37-57), the
 * constructor `BvhNode::new` (bvh.rs:14-18) and `Camera::render`
 * (camera.rs:78-158).  Every entry point below names the reference interface it
 * replaces.  All geometry is f64, exactly as in the reference.
 *
 * Conventions
 *   - plain C, POD structs, no callbacks, no torch types;
 *   - every function returns 0 on success or a negative RT_E_* code; the text of
 *     the last error on the calling thread is returned by rt_last_error();
 *   - host buffers are owned by the caller and copied during the call; device
 *     memory behind `rt_scene*` is owned by the library;
 *   - functions whose name ends in `_device` take DEVICE pointers and enqueue on
 *     the HIP stream handed in (0 = the null stream) without synchronising.
 */
#ifndef MI355RT_H
#define MI355RT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_OK             0
#define RT_E_INVALID     -1   /* malformed description (NaN bbox, bad index, empty BVH ...) */
#define RT_E_HIP         -2   /* HIP runtime error (no device, out of memory, launch failure) */
#define RT_E_UNSUPPORTED -3   /* valid in the reference but outside what the kernels implement */
#define RT_E_TOO_LARGE   -4   /* exceeds a stated fixed limit */
#define RT_E_BUSY        -5   /* a shared resource is still in use by work enqueued elsewhere (rt_closest_hit_device_noalloc) */

typedef struct rt_scene rt_scene; /* opaque: flattened scene + device buffers */

/* ------------------------------------------------------------------ textures
 * texture.rs:6-130.  A texture is referenced by its index in rt_scene_desc.textures. */
enum { RT_TEX_SOLID = 0, RT_TEX_CHECKER = 1, RT_TEX_IMAGE = 2, RT_TEX_NOISE = 3 };

typedef struct rt_texture {
    int32_t kind;
    int32_t even;     /* CHECKER: texture id used where the cell sum is even (texture.rs:64-70) */
    int32_t odd;      /* CHECKER: texture id used otherwise */
    int32_t image;    /* IMAGE: index into images[], -1 = no image loaded (solid cyan, texture.rs:89-91) */
    int32_t perlin;   /* NOISE: index into perlins[] */
    int32_t _pad;
    double  color[3]; /* SOLID: albedo (texture.rs:27-31) */
    double  scale;    /* CHECKER: cell size, inv_scale = 1.0/scale (texture.rs:41-55); NOISE: frequency (texture.rs:126-129) */
} rt_texture;

/* RGB8 image, row-major, 3*width bytes per row (rtw_stb_image.rs:110,155-171). */
typedef struct rt_image {
    int32_t  width;
    int32_t  height;
    uint64_t offset; /* byte offset of the first pixel in rt_scene_desc.image_bytes */
} rt_image;

/* Perlin tables built by the host with its scene RNG (perlin.rs:18-53). */
typedef struct rt_perlin {
    double  randvec[256][3];
    int32_t perm_x[256];
    int32_t perm_y[256];
    int32_t perm_z[256];
} rt_perlin;

/* ----------------------------------------------------------------- materials
 * material.rs:58-301. */
enum {
    RT_MAT_LAMBERTIAN = 0,    /* tex */
    RT_MAT_METAL = 1,         /* albedo, fuzz (clamped to <=1 as material.rs:130) */
    RT_MAT_DIELECTRIC = 2,    /* ior */
    RT_MAT_DIFFUSE_LIGHT = 3, /* tex, emits on the front face only (material.rs:250-256) */
    RT_MAT_ISOTROPIC = 4      /* tex */
};

typedef struct rt_material {
    int32_t kind;
    int32_t tex;
    double  albedo[3];
    double  fuzz;
    double  ior;
} rt_material;

/* ---------------------------------------------------------------- primitives */
/* sphere.rs:25-45: static (moving = 0, center2 ignored) or linearly moving. */
typedef struct rt_sphere {
    double  center[3];
    double  center2[3];
    double  radius;
    int32_t mat;
    int32_t moving;
} rt_sphere;

/* quad.rs:28-49: parallelogram q, q+u, q+v, q+u+v. */
typedef struct rt_quad {
    double  q[3];
    double  u[3];
    double  v[3];
    int32_t mat;
    int32_t _pad;
} rt_quad;

/* triangle.rs:122-149. */
typedef struct rt_triangle {
    double  v0[3];
    double  v1[3];
    double  v2[3];
    int32_t mat;
    int32_t _pad;
} rt_triangle;

/* ------------------------------------------------------------- object graph
 * One entry per `Arc<dyn Hittable>` of the reference.  Objects may be shared
 * (the graph is a DAG), e.g. a sphere that is both in the world and the
 * boundary of a medium (main.rs:708-720). */
enum {
    RT_OBJ_SPHERE = 0,          /* a = sphere index */
    RT_OBJ_QUAD = 1,            /* a = quad index */
    RT_OBJ_TRIANGLE = 2,        /* a = triangle index */
    RT_OBJ_LIST = 3,            /* HittableList (hittable_list.rs): children [a, a+b) */
    RT_OBJ_BVH = 4,             /* BvhNode::new(list) (bvh.rs:14-77): children [a, a+b) */
    RT_OBJ_TRANSLATE = 5,       /* hittable.rs:61-94: a = child object, param = offset */
    RT_OBJ_ROTATE_Y = 6,        /* hittable.rs:97-187: a = child object, param[0] = degrees */
    RT_OBJ_CONSTANT_MEDIUM = 7, /* constant_medium.rs: a = boundary object, param[0] = density, mat = isotropic material */
    RT_OBJ_OBJMODEL = 8         /* obj_loader.rs:13-106: triangles [a, a+b), linear scan */
};

/* What a LIST / BVH child range indexes. */
enum {
    RT_CHILD_OBJECTS = 0,  /* children[a .. a+b) are object ids */
    RT_CHILD_SPHERES = 1,  /* spheres[a .. a+b) directly */
    RT_CHILD_QUADS = 2,
    RT_CHILD_TRIANGLES = 3 /* BVH + TRIANGLES is the reference's Mesh (mesh.rs:95-183) */
};

typedef struct rt_object {
    int32_t kind;
    int32_t a;
    int32_t b;
    int32_t child_kind;
    int32_t mat;
    int32_t _pad;
    double  param[3];
} rt_object;

typedef struct rt_scene_desc {
    const rt_sphere*   spheres;     uint64_t n_spheres;
    const rt_quad*     quads;       uint64_t n_quads;
    const rt_triangle* triangles;   uint64_t n_triangles;
    const rt_object*   objects;     uint64_t n_objects;
    const int32_t*     children;    uint64_t n_children;
    const rt_material* materials;   uint64_t n_materials;
    const rt_texture*  textures;    uint64_t n_textures;
    const rt_image*    images;      uint64_t n_images;
    const uint8_t*     image_bytes; uint64_t n_image_bytes;
    const rt_perlin*   perlins;     uint64_t n_perlins;
    int32_t world;  /* object id passed as `world` to Camera::render (camera.rs:80) */
    int32_t lights; /* object id passed as `lights` (camera.rs:81), or -1: material PDF only */
} rt_scene_desc;

/* rt_scene_create flags */
#define RT_SCENE_HOST_ONLY 1u /* flatten and validate only; no device is touched (for CPU-side checks) */

/* Replaces building the object graph + BvhNode::new (bvh.rs:14-77, aabb.rs:37-136,
 * hittable.rs:67-76,104-143, quad.rs:28-61,139-193): validates the description,
 * builds every BVH in the reference's split order, flattens the graph into
 * linear 64-byte-aligned arrays and uploads them to the current HIP device. */
int rt_scene_create(const rt_scene_desc* desc, uint32_t flags, rt_scene** out);
int rt_scene_destroy(rt_scene* scene);

/* Flattened-node introspection (host copy), used to check split order one-to-one
 * with the reference's tree.  A reference is (kind << 28) | index. */
#define RT_REF_NODE    0u
#define RT_REF_SPHERE  1u
#define RT_REF_QUAD    2u
#define RT_REF_TRI     3u
#define RT_REF_LIST    4u
#define RT_REF_XFORM   5u
#define RT_REF_MEDIUM  6u
#define RT_REF_NONE    0xFFFFFFFFu
#define RT_REF_KIND(r)  ((uint32_t)(r) >> 28)
#define RT_REF_INDEX(r) ((uint32_t)(r) & 0x0FFFFFFFu)

typedef struct rt_bvh_node { /* 64 bytes */
    double   bounds[6]; /* xmin xmax ymin ymax zmin zmax (aabb.rs:10-14) */
    uint32_t left;      /* reference: bvh.rs:8 */
    uint32_t right;     /* reference: bvh.rs:9; equals left for a one-object span (bvh.rs:41-44) */
    uint32_t flags;     /* bit 0: subtree contains a constant medium; bits 1-2: longest axis of the box (bvh.rs:32), along which the objects were split */
    uint32_t _pad;
} rt_bvh_node;

typedef struct rt_scene_info {
    uint64_t n_nodes;
    uint64_t n_lists;
    uint64_t n_xforms;
    uint64_t n_media;
    uint64_t n_lights;
    uint32_t world_ref;
    uint32_t max_stack;    /* traversal stack entries the scene can need */
    uint64_t device_bytes; /* bytes resident in HBM for this scene */
    double   world_bounds[6];
} rt_scene_info;

int rt_scene_get_info(const rt_scene* scene, rt_scene_info* info);
/* Copies nodes [first, first+count) of the flattened node array. */
int rt_scene_get_nodes(const rt_scene* scene, uint64_t first, uint64_t count, rt_bvh_node* out);
/* The 4-wide nodes the primitive-BVH kernel traverses when the world is one BVH over primitives
 * (two levels of bvh.rs:20-77 collapsed per node): for every child the box rounded outward to f32
 * and, when the scene also carries the 64-byte nodes, the box decoded from their 8-bit grid.
 * Used to check that each stored box contains the reference's box of that child. */
typedef struct rt_wide_node {
    uint32_t child[4];     /* RT_REF_NODE | wide index, a primitive reference, or RT_REF_NONE */
    double   box32[4][6];  /* xmin xmax ymin ymax zmin zmax per child (f32 values) */
    double   box8[4][6];   /* decoded 8-bit boxes; valid only when has_box8 != 0 */
    int32_t  has_box8;
    int32_t  _pad;
} rt_wide_node;
/* Number of wide nodes (0 when the scene has no primitive-only BVH world); node 0 is the root. */
int rt_scene_wide_node_count(const rt_scene* scene, uint64_t* count);
int rt_scene_get_wide_nodes(const rt_scene* scene, uint64_t first, uint64_t count, rt_wide_node* out);
/* The same tree in the packed form the primitive-BVH kernel reads for a triangle mesh (48-byte nodes, children
 * addressed from two base indices, triangles in leaf order), decoded: child = RT_REF_NODE | packed index or
 * RT_REF_TRI | index of the triangle in the description; box8 as above; box32 is zero.  `leaf_overlap`: sum of the
 * triangles' box areas over the world box's (how many triangle boxes a random line crosses).  Count 0: no packed tree. */
int rt_scene_packed_node_count(const rt_scene* scene, uint64_t* count, double* leaf_overlap);
/* The tree the nearest-first kernel walks when the world is one BVH over primitives: index of its root in the wide-node
 * array, the number of triangle references in its packed form (0 without one), and whether it is a traversal-only tree
 * (a mesh of MI355RT_FAST_TREE_MIN = 1024 triangles or more: surface-area heuristic with spatial splits instead of
 * the collapse of the reference's median-split tree, bvh.rs:20-77, which RT_HIT_REFERENCE_ORDER and RT_HIT_EXACT_CULL
 * keep walking; a triangle may then be the child of several nodes, each with the box of a part of it).  Any of the
 * three out pointers may be NULL.  RT_E_INVALID when the world is not such a BVH. */
int rt_scene_fast_tree_info(const rt_scene* scene, uint64_t* wide_root, uint64_t* references, int32_t* traversal_only);
int rt_scene_get_packed_nodes(const rt_scene* scene, uint64_t first, uint64_t count, rt_wide_node* out);
/* bounding_box() of an object of the description (hittable.rs:50), as 6 doubles. */
int rt_scene_object_bounds(const rt_scene* scene, int32_t object, double out[6]);

/* Replaces `tobj::load_obj(path, {triangulate, single_index})` and the vertex loops of Mesh::from_obj
 * (mesh.rs:101-158) and ObjModel::load (obj_loader.rs:21-83): reads a Wavefront OBJ file into triangles.
 * Positions are parsed to f32 and widened to f64 as tobj + mesh.rs:137-153 do; faces are fan-triangulated;
 * all models of the file are concatenated.  scale_offset = NULL keeps the positions (Mesh::from_obj);
 * otherwise {scale, offset x, y, z}: every component becomes p * scale + offset in f64 (obj_loader.rs:51-67).
 * The array is allocated by the library (host memory); release it with rt_obj_free.  Use the triangles as
 * RT_OBJ_BVH + RT_CHILD_TRIANGLES (Mesh) or RT_OBJ_OBJMODEL (ObjModel) of a scene description. */
int  rt_obj_load(const char* path, const double* scale_offset, int32_t mat, rt_triangle** triangles, uint64_t* n_triangles);
void rt_obj_free(rt_triangle* triangles);

/* ------------------------------------------------------------- closest hit
 * Batched form of `world.hit(&ray, Interval::new(tmin, tmax), &mut rec)`
 * (hittable.rs:49; caller camera.rs:258).  Structure-of-arrays, f64. */
typedef struct rt_ray_batch {
    const double* ox; const double* oy; const double* oz; /* ray.rs:6 */
    const double* dx; const double* dy; const double* dz; /* ray.rs:7, not normalised */
    const double* time;                                   /* ray.rs:8; NULL = 0.0 */
    const double* tmin;                                   /* NULL = 0.001 (camera.rs:258) */
    const double* tmax;                                   /* NULL = +infinity */
} rt_ray_batch;

typedef struct rt_hit_batch {
    double*   t;      /* HitRecord.t; +infinity where nothing was hit */
    uint32_t* prim;   /* (kind << 28) | index of the primitive in the description, RT_REF_NONE on a miss;
                         kinds RT_REF_SPHERE / _QUAD / _TRI / _MEDIUM (index = object id of the medium) */
    double*   u;      /* HitRecord.u, v (hittable.rs:19-20).  A constant-medium hit never writes them in the reference
                         (constant_medium.rs:96-101): reported as 0, except in scenes where a medium's phase function
                         reads u,v (image texture), where they are what the reference's record would hold */
    double*   v;
    uint8_t*  front_face;
    int32_t*  mat;    /* material id of the hit */
    double*   nx; double* ny; double* nz; /* HitRecord.normal; any of these 6 may be NULL */
    double*   px; double* py; double* pz; /* HitRecord.p */
} rt_hit_batch;

/* rt_closest_hit flags */
#define RT_HIT_REFERENCE_ORDER 0u /* left-then-right, list order: visit order of bvh.rs:118-140 */
#define RT_HIT_ORDERED         1u /* near-child-first traversal over conservative boxes; same closest hit, ties may resolve to another
                                     primitive at equal t.  One deviation beyond ties: a ray whose entry into a box coincides exactly
                                     with its exit (through an edge or a corner) is culled there by the reference (aabb.rs:102) but not
                                     by the conservative test, so a primitive touching that edge can be hit here and not there. */
#define RT_HIT_COUNTERS        2u /* also accumulate node-visit / primitive-test counters (slower) */
#define RT_HIT_EXACT_CULL      4u /* with RT_HIT_ORDERED: a box is entered only if Aabb::hit would enter it (aabb.rs:82-107, f64, a box
                                     whose clipped interval is a single point is culled), nearest box first.  A world that is one BVH
                                     over primitives is walked on the 4-wide collapse of the reference's own tree: the f32 slab test
                                     decides alone where it passes with room to spare (more than its rounding error on both ends),
                                     and the f64 test of Aabb::hit decides the rest.  Hits the reference never tests because it culls a
                                     touched box (see RT_HIT_ORDERED) are not reported.  What remains of the visit order: when two
                                     primitives are hit within rounding of each other (a ray through a shared vertex: t equal to
                                     about 1e-13 relative), the box of one can start, in floating point, a hair behind the other's hit,
                                     and which of the two is reported depends on which was reached first; the reported hit is always
                                     a hit the reference's arithmetic produces for that primitive. */

/* Random numbers drawn inside traversal (constant_medium.rs:90) come from the
 * counter-based stream (seed, stream 2, ray index); see rt_rng_uniform.
 * Threads and streams: rt_closest_hit* / rt_render* may be called on one scene from several host threads and on
 * several streams at once, as the reference calls world.hit from every rayon worker (camera.rs:99).  A scene owns
 * one set of work counters, scratch buffers and timing events, so the library puts such calls in a row: a mutex of
 * the scene is held while a call enqueues (to the end of the call for the host-pointer entry points, which wait
 * for their results), and a call's first launch waits on the device for the previous call on that scene when the
 * two use different streams.  Calls on different scenes are independent.  rt_closest_hit_device_noalloc is the
 * exception on the device side (see there).
 * The *_device entry points enqueue and return.  They grow the scene's scratch (4 bytes per ray of transform
 * context; for large batches on some triangle meshes 66 bytes per ray for the binning pass; the path workspace of
 * a render) with hipMalloc, synchronously, when a call needs more than any call before it; rt_scene_reserve sizes
 * all of it ahead of time, after which calls within those sizes allocate nothing. */
int rt_closest_hit(rt_scene* scene, const rt_ray_batch* rays, uint64_t n,
                   rt_hit_batch* hits, uint64_t seed, uint32_t flags);          /* host pointers */
int rt_closest_hit_device(rt_scene* scene, const rt_ray_batch* rays, uint64_t n,
                          rt_hit_batch* hits, uint64_t seed, uint32_t flags,
                          void* hip_stream);                                     /* device pointers */

/* No counterpart in the reference (Rust allocates where it needs to).  Sizes the scene's scratch for closest-hit
 * batches of up to max_rays rays and render batches of up to max_paths paths (0: leave as it is), so that later
 * *_device calls within those sizes only enqueue.  Waits for the device first. */
int rt_scene_reserve(rt_scene* scene, uint64_t max_rays, uint64_t max_paths);

/* No counterpart in the reference.  For callers that cannot allocate device memory while they call (tight
 * serving loops, harnesses that forbid hipMalloc, code that must not block on the allocator): the same query as
 * rt_closest_hit_device, same arguments and results, but the 4 bytes per ray of scratch that the general
 * traversal keeps between its two kernels live in a fixed 64 MiB array inside the library's code object
 * (one copy per GPU, held until the process ends) instead of a buffer grown with hipMalloc.
 *   - n <= RT_NOALLOC_MAX_RAYS, else RT_E_TOO_LARGE;
 *   - calls no hipMalloc / hipFree; the array's address is resolved per device by rt_scene_create;
 *   - every call on one GPU shares that GPU's array: calls that overlap on one GPU (two streams, two
 *     scenes, two threads) would race on it and must be serialised by the caller.  The library checks what it
 *     can see: a call on another stream than the previous one, while that one's work has not finished, returns
 *     RT_E_BUSY and enqueues nothing (calls on one stream follow each other and are always accepted; nothing is
 *     checked, or recorded, on a stream that is being captured into a graph);
 *   - HIP graph capture: the call makes no allocation, synchronisation, event query or timing record, so it can be
 *     captured; REPLAY has only been seen to work for worlds traced by the general kernels (a flat list of objects).
 *     For a world that is one BVH over primitives a replayed call left most rays untraced with the ROCm 7.0 runtime
 *     (the work counters' reset, a memset node, did not take effect on replay), and unwritten outputs then feed the
 *     record completion.  Do not replay captured calls of this library; call it directly. */
#define RT_NOALLOC_MAX_RAYS (1ull << 24)
int rt_closest_hit_device_noalloc(rt_scene* scene, const rt_ray_batch* rays, uint64_t n,
                                  rt_hit_batch* hits, uint64_t seed, uint32_t flags,
                                  void* hip_stream);                             /* device pointers */

/* -------------------------------------------------------------------- render
 * Camera parameters: the public fields of camera.rs:15-27. */
typedef struct rt_camera {
    double  aspect_ratio;
    int32_t image_width;
    int32_t samples_per_pixel;
    int32_t max_depth;
    int32_t russian_roulette; /* 0 (the reference at HEAD): off.  1: the termination rule of the reference's history
                                 (camera.rs:407-416, time_record.txt:23-24): at a bounce sampled through a PDF the path
                                 goes on with probability clamp(max(attenuation), 0.1, 1) and otherwise returns its
                                 emission; as written there, survivors are not re-weighted. */
    double  background[3];
    double  vfov;
    double  lookfrom[3];
    double  lookat[3];
    double  vup[3];
    double  defocus_angle;
    double  focus_dist;
} rt_camera;

/* Derived values of Camera::initialize (camera.rs:161-205). */
typedef struct rt_camera_derived {
    int32_t image_height;
    int32_t sqrt_spp;
    double  pixel_samples_scale;
    double  recip_sqrt_spp;
    double  center[3];
    double  pixel00_loc[3];
    double  pixel_delta_u[3];
    double  pixel_delta_v[3];
    double  defocus_disk_u[3];
    double  defocus_disk_v[3];
} rt_camera_derived;

int rt_camera_initialize(const rt_camera* cam, rt_camera_derived* out);

/* Work unit of the frame partition: pixel rectangle [x0,x1) x [y0,y1) and the
 * strata [sample_begin, sample_end) of the sqrt_spp x sqrt_spp grid, linear index
 * s = s_j * sqrt_spp + s_i (camera.rs:110-111). */
typedef struct rt_work {
    int32_t x0, y0, x1, y1;
    int32_t sample_begin, sample_end;
} rt_work;

/* Replaces the sampling loop of Camera::render (camera.rs:99-131) for one work
 * unit: raygen -> {extend, shade, compact}* as HIP kernels.  `accum` is the full
 * image, row-major, 3 doubles per pixel (sum of ray_color over the unit's strata,
 * NOT yet multiplied by pixel_samples_scale); pixels of the unit are ADDED to.
 * Every random draw comes from the stream (seed, stream 0, pixel, stratum), so
 * the image does not depend on how the frame is partitioned. */
int rt_render(rt_scene* scene, const rt_camera* cam, uint64_t seed,
              const rt_work* work, double* accum);                       /* host accum */
int rt_render_device(rt_scene* scene, const rt_camera* cam, uint64_t seed,
                     const rt_work* work, double* accum, void* hip_stream); /* device accum */

/* Replaces camera.rs:133-135 + write_color_to_string (color.rs:58-80):
 * scale, sqrt gamma, clamp to [0, 0.999], truncate 256*x.  Host buffers. */
int rt_resolve_rgb8(const rt_camera* cam, const double* accum, uint8_t* rgb);
/* Text PPM exactly as camera.rs:86,151-154 print it ("P3\nW H\n255\n" + "r g b\n" per pixel). */
int rt_write_ppm(const char* path, int32_t width, int32_t height, const uint8_t* rgb);

/* ------------------------------------------------------------------ counters */
typedef struct rt_counters {
    uint64_t rays;           /* world.hit queries issued through rt_closest_hit* */
    uint64_t node_visits;    /* Aabb::hit evaluations (only with RT_HIT_COUNTERS) */
    uint64_t prim_tests;     /* primitive hit evaluations (only with RT_HIT_COUNTERS) */
    uint64_t paths;          /* camera samples started by rt_render* */
    uint64_t render_rays;    /* world.hit queries issued by rt_render* (camera.rs:258 calls) */
    uint64_t extend_launches;
    double   ms_last_closest_hit; /* device time of the last rt_closest_hit* call: reorder pass, traversal, record completion (HIP events on the call's stream) */
    double   ms_last_render;      /* device time of the last rt_render* call */
    double   ms_last_traversal;   /* of that closest-hit call, the traversal kernel alone */
    double   ms_last_reorder;     /* of that closest-hit call, the binning pass in front of the traversal (0 when the batch was not reordered) */
    double   ms_sum_closest_hit;  /* the same three, summed over the closest-hit calls since rt_reset_counters (the library keeps the */
    double   ms_sum_traversal;    /* events of the 32 newest calls between two rt_get_counters; older ones are not counted) */
    double   ms_sum_reorder;
    uint64_t timed_calls;         /* number of calls in those sums */
    uint64_t shader_khz;          /* shader clock during the last primitive-BVH launch with RT_HIT_COUNTERS (a wave's cycle counter
                                     against the constant 100 MHz counter); 0 until such a launch has run */
} rt_counters;

int rt_get_counters(rt_scene* scene, rt_counters* out);
int rt_reset_counters(rt_scene* scene);

/* ----------------------------------------------------------------------- rng
 * Replaces rtweekend.rs:15-17 (thread_rng) with Philox4x32-10:
 * key = (seed low, seed high), counter = (index >> 1, stream, a, b); the draw is
 * the high 53 bits of words (0,1) for even index, (2,3) for odd, times 2^-53.
 * stream 0: render paths (a = pixel j*width+i, b = stratum); stream 1: host scene
 * construction; stream 2: closest-hit batches (a,b = low/high ray index). */
double rt_rng_uniform(uint64_t seed, uint32_t stream, uint32_t a, uint32_t b, uint64_t index);

const char* rt_last_error(void);
const char* rt_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MI355RT_H */
