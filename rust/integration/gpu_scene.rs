// gpu_scene.rs -- safe host side over mi355rt-sys: the scene builder every Hittable describes itself into,
// `GpuScene` (the flattened scene resident in HBM) and `impl Hittable for GpuScene`.
//
// Add to the crate as `mod gpu_scene;` (plus `mod vec3;`, which the snapshot lacks).  Not compiled in the image this
// library was built in (no Rust toolchain); the same calls are exercised by the C++ and Python hosts of the library.
use crate::aabb::Aabb;
use crate::hittable::{HitRecord, Hittable};
use crate::interval::Interval;
use crate::ray::Ray;
use crate::vec3::{Point3, Vec3};
use mi355rt_sys as sys;
use std::collections::HashMap;
use std::ffi::CStr;
use std::os::raw::c_void;
use std::ptr;

#[derive(Debug)]
pub struct RtError { pub code: i32, pub message: String }

fn check(rc: i32) -> Result<(), RtError> {
    if rc == 0 { return Ok(()); }
    let message = unsafe { CStr::from_ptr(sys::rt_last_error()) }.to_string_lossy().into_owned();
    Err(RtError { code: rc, message })
}

/// What every implementor of `Hittable` (and of `Material` / `Texture`) appends itself to: the POD arrays of
/// `rt_scene_desc`.  Shared objects (`Arc` clones) are described once: `memo` maps the Arc's address to its id.
#[derive(Default)]
pub struct SceneBuilder {
    pub spheres: Vec<sys::rt_sphere>,
    pub quads: Vec<sys::rt_quad>,
    pub triangles: Vec<sys::rt_triangle>,
    pub objects: Vec<sys::rt_object>,
    pub children: Vec<i32>,
    pub materials: Vec<sys::rt_material>,
    pub textures: Vec<sys::rt_texture>,
    pub images: Vec<sys::rt_image>,
    pub image_bytes: Vec<u8>,
    pub perlins: Vec<sys::rt_perlin>,
    pub memo: HashMap<usize, i32>,
}

pub const RT_OBJ_SPHERE: i32 = 0;
pub const RT_OBJ_QUAD: i32 = 1;
pub const RT_OBJ_TRIANGLE: i32 = 2;
pub const RT_OBJ_LIST: i32 = 3;
pub const RT_OBJ_BVH: i32 = 4;
pub const RT_OBJ_TRANSLATE: i32 = 5;
pub const RT_OBJ_ROTATE_Y: i32 = 6;
pub const RT_OBJ_CONSTANT_MEDIUM: i32 = 7;
pub const RT_OBJ_OBJMODEL: i32 = 8;
pub const RT_CHILD_OBJECTS: i32 = 0;
pub const RT_CHILD_TRIANGLES: i32 = 3;
pub const RT_HIT_REFERENCE_ORDER: u32 = 0;
pub const RT_HIT_ORDERED: u32 = 1;
pub const RT_HIT_EXACT_CULL: u32 = 4;
pub const RT_REF_NONE: u32 = 0xFFFF_FFFF;

impl SceneBuilder {
    pub fn new() -> Self { Self::default() }
    pub fn v(p: Vec3) -> [f64; 3] { [p.x(), p.y(), p.z()] }
    /// Appends one object record and returns its id.
    pub fn emit(&mut self, kind: i32, a: i32, b: i32, child_kind: i32, mat: i32, param: [f64; 3]) -> i32 {
        self.objects.push(sys::rt_object { kind, a, b, child_kind, mat, _pad: 0, param });
        (self.objects.len() - 1) as i32
    }
    /// Children of a list / BVH: their ids go into `children`, the object refers to the range.
    pub fn emit_group(&mut self, kind: i32, ids: &[i32]) -> i32 {
        let first = self.children.len() as i32;
        self.children.extend_from_slice(ids);
        self.emit(kind, first, ids.len() as i32, RT_CHILD_OBJECTS, -1, [0.0; 3])
    }
    fn desc(&self, world: i32, lights: i32) -> sys::rt_scene_desc {
        fn p<T>(v: &Vec<T>) -> *const T { if v.is_empty() { ptr::null() } else { v.as_ptr() } }
        sys::rt_scene_desc {
            spheres: p(&self.spheres), n_spheres: self.spheres.len() as u64,
            quads: p(&self.quads), n_quads: self.quads.len() as u64,
            triangles: p(&self.triangles), n_triangles: self.triangles.len() as u64,
            objects: p(&self.objects), n_objects: self.objects.len() as u64,
            children: p(&self.children), n_children: self.children.len() as u64,
            materials: p(&self.materials), n_materials: self.materials.len() as u64,
            textures: p(&self.textures), n_textures: self.textures.len() as u64,
            images: p(&self.images), n_images: self.images.len() as u64,
            image_bytes: p(&self.image_bytes), n_image_bytes: self.image_bytes.len() as u64,
            perlins: p(&self.perlins), n_perlins: self.perlins.len() as u64,
            world, lights,
        }
    }
}

/// Implemented next to `Hittable` by every scene object (describe_impls.rs holds the impl blocks).
pub trait Describe {
    fn describe(&self, b: &mut SceneBuilder) -> i32;
}

/// The flattened scene on the GPU.  `rt_scene_create` redoes the split of `BvhNode::from_objects` (bvh.rs:20-77) in
/// the same order, flattens the graph and uploads it to the current HIP device.
pub struct GpuScene {
    handle: *mut sys::rt_scene,
    bounds: Aabb,
    pub seed: u64,
}
// The handle may be used from every rayon worker at once, as the reference uses `world` (camera.rs:99): the library
// holds a mutex of the scene while a call enqueues (to the end of the call for the host-pointer entry points, which
// wait for their results) and orders calls on different streams on the device (mi355rt.h, "Threads and streams").
// `rt_last_error` is thread-local.  Nothing behind the handle is reachable from Rust except through those calls.
unsafe impl Send for GpuScene {}
unsafe impl Sync for GpuScene {}

impl GpuScene {
    pub fn new(world: &dyn Describe, lights: Option<&dyn Describe>) -> Result<Self, RtError> {
        let mut b = SceneBuilder::new();
        let w = world.describe(&mut b);
        let l = lights.map_or(-1, |l| l.describe(&mut b));
        let desc = b.desc(w, l);
        let mut handle: *mut sys::rt_scene = ptr::null_mut();
        check(unsafe { sys::rt_scene_create(&desc, 0, &mut handle) })?;
        let mut wb = [0.0f64; 6];
        check(unsafe { sys::rt_scene_object_bounds(handle, w, wb.as_mut_ptr()) })?;
        let bounds = Aabb::from_intervals(Interval::new(wb[0], wb[1]), Interval::new(wb[2], wb[3]), Interval::new(wb[4], wb[5]));
        Ok(Self { handle, bounds, seed: 0 })
    }
    pub fn raw(&self) -> *mut sys::rt_scene { self.handle }

    /// Batched `world.hit(&ray, Interval::new(tmin, tmax), &mut rec)` over host slices (copied to the device and back).
    /// Returns, per ray, `Some(record)` or `None`; `rec.mat` is left `None` (materials live in the library; use the
    /// material id of the hit batch when shading on the host is wanted).
    pub fn hit_batch(&self, rays: &[Ray], ray_t: Interval, flags: u32) -> Result<Vec<Option<HitRecord>>, RtError> {
        let n = rays.len();
        let col = |f: &dyn Fn(&Ray) -> f64| rays.iter().map(|r| f(r)).collect::<Vec<f64>>();
        let (ox, oy, oz) = (col(&|r| r.origin().x()), col(&|r| r.origin().y()), col(&|r| r.origin().z()));
        let (dx, dy, dz) = (col(&|r| r.direction().x()), col(&|r| r.direction().y()), col(&|r| r.direction().z()));
        let tm = col(&|r| r.time());
        let (tmin, tmax) = (vec![ray_t.min; n], vec![ray_t.max; n]);
        let batch = sys::rt_ray_batch { ox: ox.as_ptr(), oy: oy.as_ptr(), oz: oz.as_ptr(), dx: dx.as_ptr(), dy: dy.as_ptr(), dz: dz.as_ptr(),
                                        time: tm.as_ptr(), tmin: tmin.as_ptr(), tmax: tmax.as_ptr() };
        let mut t = vec![0.0f64; n]; let mut prim = vec![0u32; n]; let mut u = vec![0.0f64; n]; let mut v = vec![0.0f64; n];
        let mut ff = vec![0u8; n]; let mut mat = vec![0i32; n];
        let (mut nx, mut ny, mut nz) = (vec![0.0f64; n], vec![0.0f64; n], vec![0.0f64; n]);
        let (mut px, mut py, mut pz) = (vec![0.0f64; n], vec![0.0f64; n], vec![0.0f64; n]);
        let mut hits = sys::rt_hit_batch { t: t.as_mut_ptr(), prim: prim.as_mut_ptr(), u: u.as_mut_ptr(), v: v.as_mut_ptr(),
                                           front_face: ff.as_mut_ptr(), mat: mat.as_mut_ptr(), nx: nx.as_mut_ptr(), ny: ny.as_mut_ptr(),
                                           nz: nz.as_mut_ptr(), px: px.as_mut_ptr(), py: py.as_mut_ptr(), pz: pz.as_mut_ptr() };
        check(unsafe { sys::rt_closest_hit(self.handle, &batch, n as u64, &mut hits, self.seed, flags) })?;
        Ok((0..n).map(|i| {
            if prim[i] == RT_REF_NONE { return None; }
            let mut rec = HitRecord::default();
            rec.t = t[i]; rec.u = u[i]; rec.v = v[i]; rec.front_face = ff[i] != 0;
            rec.p = Point3::new(px[i], py[i], pz[i]);
            rec.normal = Vec3::new(nx[i], ny[i], nz[i]);
            Some(rec)
        }).collect())
    }
}

impl Drop for GpuScene {
    fn drop(&mut self) { unsafe { sys::rt_scene_destroy(self.handle); } }
}

/// Drop-in for the trait (hittable.rs:37-57).  A single ray is a batch of one: correct, and slow; callers that own
/// many rays (a wavefront integrator, light-PDF evaluation) call `hit_batch`.  The traversal is the reference-order
/// one, so the record equals what `BvhNode::hit` / `HittableList::hit` produce for the described graph.
impl Hittable for GpuScene {
    fn hit(&self, r: &Ray, ray_t: Interval, rec: &mut HitRecord) -> bool {
        match self.hit_batch(std::slice::from_ref(r), ray_t, RT_HIT_REFERENCE_ORDER) {
            Ok(mut v) => match v.pop().flatten() {
                Some(h) => { let mat = rec.mat.take(); *rec = h; rec.mat = mat; true }
                None => false,
            },
            Err(e) => panic!("mi355rt: {} ({})", e.message, e.code),
        }
    }
    fn bounding_box(&self) -> Aabb { self.bounds }
    // pdf_value / random keep the trait's defaults (0.0 and (1, 0, 0), hittable.rs:51-56): light sampling happens
    // inside the library's shade kernel, from the light list handed to GpuScene::new.
}

/// Device pointers for callers that keep rays in HBM (a HIP stream as `*mut c_void`, null = the null stream).
pub unsafe fn hit_batch_device(scene: &GpuScene, rays: &sys::rt_ray_batch, n: u64, hits: &mut sys::rt_hit_batch, flags: u32,
                               stream: *mut c_void) -> Result<(), RtError> {
    check(sys::rt_closest_hit_device(scene.raw(), rays, n, hits, scene.seed, flags, stream))
}
