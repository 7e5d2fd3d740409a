//! mi355rt-sys -- raw FFI declarations of libmi355rt.so (include/mi355rt.h): BVH traversal, closest-hit
//! intersection and the wavefront path tracer of this ray tracer, executed by HIP kernels on AMD MI355X (gfx950).
//!
//! One `#[repr(C)]` struct per C struct, one `extern "C"` declaration per entry point, same names and field order
//! as the header (tests/test_host_flatten.py checks this file against the header and the ctypes binding).
//! Safe wrappers live in the host crate (rust/integration/*.rs).
//!
//! The image this library was built in has no Rust toolchain: this crate is source only and has not been compiled.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)] pub struct rt_scene { _private: [u8; 0] }

#[repr(C)] #[derive(Clone, Copy)] pub struct rt_sphere   { pub center: [f64; 3], pub center2: [f64; 3], pub radius: f64, pub mat: i32, pub moving: i32 }
#[repr(C)] #[derive(Clone, Copy)] pub struct rt_quad     { pub q: [f64; 3], pub u: [f64; 3], pub v: [f64; 3], pub mat: i32, pub _pad: i32 }
#[repr(C)] #[derive(Clone, Copy)] pub struct rt_triangle { pub v0: [f64; 3], pub v1: [f64; 3], pub v2: [f64; 3], pub mat: i32, pub _pad: i32 }
#[repr(C)] #[derive(Clone, Copy)] pub struct rt_object   { pub kind: i32, pub a: i32, pub b: i32, pub child_kind: i32, pub mat: i32, pub _pad: i32, pub param: [f64; 3] }
#[repr(C)] #[derive(Clone, Copy)] pub struct rt_material { pub kind: i32, pub tex: i32, pub albedo: [f64; 3], pub fuzz: f64, pub ior: f64 }
#[repr(C)] #[derive(Clone, Copy)] pub struct rt_texture  { pub kind: i32, pub even: i32, pub odd: i32, pub image: i32, pub perlin: i32, pub _pad: i32, pub color: [f64; 3], pub scale: f64 }
#[repr(C)] #[derive(Clone, Copy)] pub struct rt_image    { pub width: i32, pub height: i32, pub offset: u64 }
#[repr(C)] pub struct rt_perlin { pub randvec: [[f64; 3]; 256], pub perm_x: [i32; 256], pub perm_y: [i32; 256], pub perm_z: [i32; 256] }

#[repr(C)] pub struct rt_scene_desc {
    pub spheres: *const rt_sphere, pub n_spheres: u64,
    pub quads: *const rt_quad, pub n_quads: u64,
    pub triangles: *const rt_triangle, pub n_triangles: u64,
    pub objects: *const rt_object, pub n_objects: u64,
    pub children: *const i32, pub n_children: u64,
    pub materials: *const rt_material, pub n_materials: u64,
    pub textures: *const rt_texture, pub n_textures: u64,
    pub images: *const rt_image, pub n_images: u64,
    pub image_bytes: *const u8, pub n_image_bytes: u64,
    pub perlins: *const rt_perlin, pub n_perlins: u64,
    pub world: i32, pub lights: i32,
}
#[repr(C)] pub struct rt_ray_batch { pub ox: *const f64, pub oy: *const f64, pub oz: *const f64, pub dx: *const f64, pub dy: *const f64, pub dz: *const f64,
                                     pub time: *const f64, pub tmin: *const f64, pub tmax: *const f64 }
#[repr(C)] pub struct rt_hit_batch { pub t: *mut f64, pub prim: *mut u32, pub u: *mut f64, pub v: *mut f64, pub front_face: *mut u8, pub mat: *mut i32,
                                     pub nx: *mut f64, pub ny: *mut f64, pub nz: *mut f64, pub px: *mut f64, pub py: *mut f64, pub pz: *mut f64 }
#[repr(C)] pub struct rt_camera { pub aspect_ratio: f64, pub image_width: i32, pub samples_per_pixel: i32, pub max_depth: i32, pub russian_roulette: i32,
                                  pub background: [f64; 3], pub vfov: f64, pub lookfrom: [f64; 3], pub lookat: [f64; 3], pub vup: [f64; 3],
                                  pub defocus_angle: f64, pub focus_dist: f64 }
#[repr(C)] pub struct rt_work { pub x0: i32, pub y0: i32, pub x1: i32, pub y1: i32, pub sample_begin: i32, pub sample_end: i32 }
#[repr(C)] pub struct rt_camera_derived { pub image_height: i32, pub sqrt_spp: i32, pub pixel_samples_scale: f64, pub recip_sqrt_spp: f64,
                                          pub center: [f64; 3], pub pixel00_loc: [f64; 3], pub pixel_delta_u: [f64; 3], pub pixel_delta_v: [f64; 3],
                                          pub defocus_disk_u: [f64; 3], pub defocus_disk_v: [f64; 3] }
#[repr(C)] pub struct rt_bvh_node { pub bounds: [f64; 6], pub left: u32, pub right: u32, pub flags: u32, pub _pad: u32 }
#[repr(C)] pub struct rt_wide_node { pub child: [u32; 4], pub box32: [[f64; 6]; 4], pub box8: [[f64; 6]; 4], pub has_box8: i32, pub _pad: i32 }
#[repr(C)] pub struct rt_scene_info { pub n_nodes: u64, pub n_lists: u64, pub n_xforms: u64, pub n_media: u64, pub n_lights: u64,
                                      pub world_ref: u32, pub max_stack: u32, pub device_bytes: u64, pub world_bounds: [f64; 6] }
#[repr(C)] pub struct rt_counters { pub rays: u64, pub node_visits: u64, pub prim_tests: u64, pub paths: u64, pub render_rays: u64,
                                    pub extend_launches: u64, pub ms_last_closest_hit: f64, pub ms_last_render: f64, pub ms_last_traversal: f64, pub ms_last_reorder: f64,
                                    pub ms_sum_closest_hit: f64, pub ms_sum_traversal: f64, pub ms_sum_reorder: f64, pub timed_calls: u64, pub shader_khz: u64 }

#[link(name = "mi355rt")]
extern "C" {
    pub fn rt_scene_create(desc: *const rt_scene_desc, flags: u32, out: *mut *mut rt_scene) -> c_int;
    pub fn rt_scene_destroy(scene: *mut rt_scene) -> c_int;
    pub fn rt_scene_get_info(scene: *const rt_scene, info: *mut rt_scene_info) -> c_int;
    pub fn rt_scene_get_nodes(scene: *const rt_scene, first: u64, count: u64, out: *mut rt_bvh_node) -> c_int;
    pub fn rt_scene_wide_node_count(scene: *const rt_scene, count: *mut u64) -> c_int;
    pub fn rt_scene_get_wide_nodes(scene: *const rt_scene, first: u64, count: u64, out: *mut rt_wide_node) -> c_int;
    pub fn rt_scene_packed_node_count(scene: *const rt_scene, count: *mut u64, leaf_overlap: *mut f64) -> c_int;
    pub fn rt_scene_fast_tree_info(scene: *const rt_scene, wide_root: *mut u64, references: *mut u64, traversal_only: *mut i32) -> c_int;
    pub fn rt_scene_get_packed_nodes(scene: *const rt_scene, first: u64, count: u64, out: *mut rt_wide_node) -> c_int;
    pub fn rt_scene_object_bounds(scene: *const rt_scene, object: i32, out: *mut f64) -> c_int;
    pub fn rt_obj_load(path: *const c_char, scale_offset: *const f64, mat: i32, triangles: *mut *mut rt_triangle, n_triangles: *mut u64) -> c_int;
    pub fn rt_obj_free(triangles: *mut rt_triangle);
    pub fn rt_closest_hit(scene: *mut rt_scene, rays: *const rt_ray_batch, n: u64, hits: *mut rt_hit_batch, seed: u64, flags: u32) -> c_int;
    pub fn rt_scene_reserve(scene: *mut rt_scene, max_rays: u64, max_paths: u64) -> c_int;
    pub fn rt_closest_hit_device(scene: *mut rt_scene, rays: *const rt_ray_batch, n: u64, hits: *mut rt_hit_batch, seed: u64, flags: u32, stream: *mut c_void) -> c_int;
    pub fn rt_closest_hit_device_noalloc(scene: *mut rt_scene, rays: *const rt_ray_batch, n: u64, hits: *mut rt_hit_batch, seed: u64, flags: u32, stream: *mut c_void) -> c_int;
    pub fn rt_camera_initialize(cam: *const rt_camera, out: *mut rt_camera_derived) -> c_int;
    pub fn rt_render(scene: *mut rt_scene, cam: *const rt_camera, seed: u64, work: *const rt_work, accum: *mut f64) -> c_int;
    pub fn rt_render_device(scene: *mut rt_scene, cam: *const rt_camera, seed: u64, work: *const rt_work, accum: *mut f64, stream: *mut c_void) -> c_int;
    pub fn rt_resolve_rgb8(cam: *const rt_camera, accum: *const f64, rgb: *mut u8) -> c_int;
    pub fn rt_write_ppm(path: *const c_char, width: i32, height: i32, rgb: *const u8) -> c_int;
    pub fn rt_get_counters(scene: *mut rt_scene, out: *mut rt_counters) -> c_int;
    pub fn rt_reset_counters(scene: *mut rt_scene) -> c_int;
    pub fn rt_rng_uniform(seed: u64, stream: u32, a: u32, b: u32, index: u64) -> f64;
    pub fn rt_last_error() -> *const c_char;
    pub fn rt_version() -> *const c_char;
}
