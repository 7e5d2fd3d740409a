// describe_impls.rs -- `impl Describe` for the scene objects of the reference.  Each block belongs in the module of
// its type (the fields it reads are private there): sphere.rs, quad.rs, triangle.rs, hittable_list.rs, bvh.rs,
// hittable.rs, constant_medium.rs, mesh.rs, obj_loader.rs, material.rs, texture.rs.  Two one-line additions to the
// reference make the description exact:
//   * `RotateY::new` keeps its `angle` argument in a new field (the library recomputes sin / cos from the angle in
//     degrees exactly as hittable.rs:106-108 does; sin_theta / cos_theta alone do not give the angle's bits back);
//   * `BvhNode::new` keeps the `HittableList` it was built from (`objects: Vec<Arc<..>>`): the library rebuilds the
//     tree from the objects, in the reference's split order, instead of walking the finished tree.
// Material and Texture get the same treatment through `DescribeMaterial` / `DescribeTexture`.
use crate::gpu_scene::*;
use mi355rt_sys as sys;
use std::sync::Arc;

pub trait DescribeMaterial { fn describe(&self, b: &mut SceneBuilder) -> i32; }
pub trait DescribeTexture { fn describe(&self, b: &mut SceneBuilder) -> i32; }

fn memo<T: ?Sized>(b: &mut SceneBuilder, a: &Arc<T>, make: impl FnOnce(&mut SceneBuilder) -> i32) -> i32 {
    let key = Arc::as_ptr(a) as *const () as usize;
    if let Some(&id) = b.memo.get(&key) { return id; }
    let id = make(b);
    b.memo.insert(key, id);
    id
}

// ---- sphere.rs (sphere.rs:25-45: the centre is a Ray from center1 towards center2; a static sphere has direction 0)
impl Describe for Sphere {
    fn describe(&self, b: &mut SceneBuilder) -> i32 {
        let c1 = self.center.origin();
        let c2 = c1 + self.center.direction();
        let moving = self.center.direction().length_squared() != 0.0;
        let mat = self.mat.describe(b);
        b.spheres.push(sys::rt_sphere { center: SceneBuilder::v(c1), center2: SceneBuilder::v(c2), radius: self.radius, mat, moving: moving as i32 });
        b.emit(RT_OBJ_SPHERE, (b.spheres.len() - 1) as i32, 0, 0, -1, [0.0; 3])
    }
}
// ---- quad.rs (quad.rs:28-49; normal, D, w and area are recomputed by the library as the constructor does)
impl Describe for Quad {
    fn describe(&self, b: &mut SceneBuilder) -> i32 {
        let mat = self.mat.describe(b);
        b.quads.push(sys::rt_quad { q: SceneBuilder::v(self.q), u: SceneBuilder::v(self.u), v: SceneBuilder::v(self.v), mat, _pad: 0 });
        b.emit(RT_OBJ_QUAD, (b.quads.len() - 1) as i32, 0, 0, -1, [0.0; 3])
    }
}
// ---- triangle.rs (triangle.rs:122-149)
impl Describe for Triangle {
    fn describe(&self, b: &mut SceneBuilder) -> i32 {
        let mat = self.material.describe(b);
        b.triangles.push(sys::rt_triangle { v0: SceneBuilder::v(self.v0), v1: SceneBuilder::v(self.v1), v2: SceneBuilder::v(self.v2), mat, _pad: 0 });
        b.emit(RT_OBJ_TRIANGLE, (b.triangles.len() - 1) as i32, 0, 0, -1, [0.0; 3])
    }
}
// ---- hittable_list.rs (hittable_list.rs:37-61: list order is visit order)
impl Describe for HittableList {
    fn describe(&self, b: &mut SceneBuilder) -> i32 {
        let ids: Vec<i32> = self.objects.iter().map(|o| memo(b, o, |b| o.describe(b))).collect();
        b.emit_group(RT_OBJ_LIST, &ids)
    }
}
// ---- bvh.rs (bvh.rs:14-77; `self.objects` is the list BvhNode::new was given, kept for this purpose)
impl Describe for BvhNode {
    fn describe(&self, b: &mut SceneBuilder) -> i32 {
        let ids: Vec<i32> = self.objects.iter().map(|o| memo(b, o, |b| o.describe(b))).collect();
        b.emit_group(RT_OBJ_BVH, &ids)
    }
}
// ---- hittable.rs (hittable.rs:61-94, 97-187)
impl Describe for Translate {
    fn describe(&self, b: &mut SceneBuilder) -> i32 {
        let child = memo(b, &self.object, |b| self.object.describe(b));
        b.emit(RT_OBJ_TRANSLATE, child, 0, 0, -1, SceneBuilder::v(self.offset))
    }
}
impl Describe for RotateY {
    fn describe(&self, b: &mut SceneBuilder) -> i32 {
        let child = memo(b, &self.object, |b| self.object.describe(b));
        b.emit(RT_OBJ_ROTATE_Y, child, 0, 0, -1, [self.angle, 0.0, 0.0]) // degrees, as passed to RotateY::new
    }
}
// ---- constant_medium.rs (constant_medium.rs:21-46: neg_inv_density = -1 / density)
impl Describe for ConstantMedium {
    fn describe(&self, b: &mut SceneBuilder) -> i32 {
        let boundary = memo(b, &self.boundary, |b| self.boundary.describe(b));
        let mat = self.phase_function.describe(b);
        b.emit(RT_OBJ_CONSTANT_MEDIUM, boundary, 0, 0, mat, [-1.0 / self.neg_inv_density, 0.0, 0.0])
    }
}
// ---- mesh.rs (mesh.rs:101-167: a BVH over the file's triangles; the library reads the file itself)
impl Mesh {
    pub fn describe_obj(path: &str, mat: i32, b: &mut SceneBuilder) -> Result<i32, RtError> {
        let first = load_obj_into(path, None, mat, b)?;
        Ok(b.emit(RT_OBJ_BVH, first, (b.triangles.len() as i32) - first, RT_CHILD_TRIANGLES, -1, [0.0; 3]))
    }
}
// ---- obj_loader.rs (obj_loader.rs:21-105: p * scale + offset per component, linear scan)
impl ObjModel {
    pub fn describe_obj(path: &str, scale: f64, offset: crate::vec3::Point3, mat: i32, b: &mut SceneBuilder) -> Result<i32, RtError> {
        let so = [scale, offset.x(), offset.y(), offset.z()];
        let first = load_obj_into(path, Some(&so), mat, b)?;
        Ok(b.emit(RT_OBJ_OBJMODEL, first, (b.triangles.len() as i32) - first, 0, -1, [0.0; 3]))
    }
}
fn load_obj_into(path: &str, scale_offset: Option<&[f64; 4]>, mat: i32, b: &mut SceneBuilder) -> Result<i32, RtError> {
    let c = std::ffi::CString::new(path).map_err(|_| RtError { code: -1, message: "path holds a NUL byte".into() })?;
    let (mut tris, mut n): (*mut sys::rt_triangle, u64) = (std::ptr::null_mut(), 0);
    let rc = unsafe { sys::rt_obj_load(c.as_ptr(), scale_offset.map_or(std::ptr::null(), |s| s.as_ptr()), mat, &mut tris, &mut n) };
    if rc != 0 {
        let m = unsafe { std::ffi::CStr::from_ptr(sys::rt_last_error()) }.to_string_lossy().into_owned();
        return Err(RtError { code: rc, message: m });
    }
    let first = b.triangles.len() as i32;
    b.triangles.extend_from_slice(unsafe { std::slice::from_raw_parts(tris, n as usize) });
    unsafe { sys::rt_obj_free(tris); }
    Ok(first)
}

// ---- material.rs (material.rs:58-301) and texture.rs (texture.rs:11-130): kinds and fields as in mi355rt.h
impl DescribeMaterial for Lambertian {
    fn describe(&self, b: &mut SceneBuilder) -> i32 {
        let tex = self.tex.describe(b);
        b.materials.push(sys::rt_material { kind: 0, tex, albedo: [0.0; 3], fuzz: 0.0, ior: 0.0 });
        (b.materials.len() - 1) as i32
    }
}
impl DescribeMaterial for Metal {
    fn describe(&self, b: &mut SceneBuilder) -> i32 {
        b.materials.push(sys::rt_material { kind: 1, tex: -1, albedo: SceneBuilder::v(self.albedo), fuzz: self.fuzz, ior: 0.0 });
        (b.materials.len() - 1) as i32
    }
}
impl DescribeMaterial for Dielectric {
    fn describe(&self, b: &mut SceneBuilder) -> i32 {
        b.materials.push(sys::rt_material { kind: 2, tex: -1, albedo: [0.0; 3], fuzz: 0.0, ior: self.refraction_index });
        (b.materials.len() - 1) as i32
    }
}
impl DescribeMaterial for DiffuseLight {
    fn describe(&self, b: &mut SceneBuilder) -> i32 {
        let tex = self.tex.describe(b);
        b.materials.push(sys::rt_material { kind: 3, tex, albedo: [0.0; 3], fuzz: 0.0, ior: 0.0 });
        (b.materials.len() - 1) as i32
    }
}
impl DescribeMaterial for Isotropic {
    fn describe(&self, b: &mut SceneBuilder) -> i32 {
        let tex = self.tex.describe(b);
        b.materials.push(sys::rt_material { kind: 4, tex, albedo: [0.0; 3], fuzz: 0.0, ior: 0.0 });
        (b.materials.len() - 1) as i32
    }
}
impl DescribeTexture for SolidColor {
    fn describe(&self, b: &mut SceneBuilder) -> i32 {
        b.textures.push(sys::rt_texture { kind: 0, even: -1, odd: -1, image: -1, perlin: -1, _pad: 0, color: SceneBuilder::v(self.albedo), scale: 0.0 });
        (b.textures.len() - 1) as i32
    }
}
impl DescribeTexture for CheckerTexture {
    fn describe(&self, b: &mut SceneBuilder) -> i32 {
        let (even, odd) = (self.even.describe(b), self.odd.describe(b));
        b.textures.push(sys::rt_texture { kind: 1, even, odd, image: -1, perlin: -1, _pad: 0, color: [0.0; 3], scale: 1.0 / self.inv_scale });
        (b.textures.len() - 1) as i32
    }
}
impl DescribeTexture for ImageTexture {
    fn describe(&self, b: &mut SceneBuilder) -> i32 {
        // RGB8, row-major, 3 * width bytes per row: the layout RtwImage::pixel_data reads (rtw_stb_image.rs:155-171)
        let image = if self.image.height() <= 0 { -1 } else {
            let offset = b.image_bytes.len() as u64;
            b.image_bytes.extend_from_slice(self.image.bytes());
            b.images.push(sys::rt_image { width: self.image.width(), height: self.image.height(), offset });
            (b.images.len() - 1) as i32
        };
        b.textures.push(sys::rt_texture { kind: 2, even: -1, odd: -1, image, perlin: -1, _pad: 0, color: [0.0; 3], scale: 0.0 });
        (b.textures.len() - 1) as i32
    }
}
impl DescribeTexture for NoiseTexture {
    fn describe(&self, b: &mut SceneBuilder) -> i32 {
        b.perlins.push(self.noise.to_rt_perlin()); // the 256 gradient vectors and three permutation tables (perlin.rs:18-53)
        let perlin = (b.perlins.len() - 1) as i32;
        b.textures.push(sys::rt_texture { kind: 3, even: -1, odd: -1, image: -1, perlin, _pad: 0, color: [0.0; 3], scale: self.scale });
        (b.textures.len() - 1) as i32
    }
}
