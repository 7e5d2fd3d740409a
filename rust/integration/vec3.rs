// vec3.rs -- the module every other file of the crate imports and the snapshot of the reference lacks
// (its src/vec3.rs is a byte copy of src/texture.rs).  Reconstructed from the call sites, with the conventions of
// "Ray Tracing in One Weekend": this is the arithmetic the library's oracle and kernels assume, operand order included.
//   new / default / x y z            sphere.rs:26, ray.rs:15-16, camera.rs:212-213
//   Index / IndexMut                 aabb.rs:88,90, hittable.rs:127-128
//   length / length_squared          constant_medium.rs:88, sphere.rs:75,77
//   Neg, Add, Sub, AddAssign, Mul<f64>, f64 * Vec3, Vec3 * Vec3 (component-wise), Div<f64>
//                                    hittable.rs:31, quad.rs:159, ray.rs:55, hittable.rs:86, camera.rs:124-134,327-335
//   min / max (component-wise)       obj_loader.rs:70-71
//   random / random_range            main.rs:134,145,757, perlin.rs:23
//   dot, cross, unit_vector, reflect, refract, random_unit_vector, random_cosine_direction, random_in_unit_disk
//                                    sphere.rs:76, quad.rs:29-33, onb.rs:9-17, material.rs:145-146,209-211, pdf.rs:37,58, camera.rs:238
// One choice the call sites cannot settle: `v / t` is `v * (1.0 / t)` (the book's form), not a per-component divide.
use crate::rtweekend::{random_double, random_double_range, PI};
use std::ops::{Add, AddAssign, Div, Index, IndexMut, Mul, Neg, Sub};

#[derive(Debug, Clone, Copy, Default, PartialEq)]
pub struct Vec3 {
    e: [f64; 3],
}
pub type Point3 = Vec3;

impl Vec3 {
    pub const fn new(x: f64, y: f64, z: f64) -> Self { Self { e: [x, y, z] } }
    pub fn x(&self) -> f64 { self.e[0] }
    pub fn y(&self) -> f64 { self.e[1] }
    pub fn z(&self) -> f64 { self.e[2] }
    pub fn length_squared(&self) -> f64 { self.e[0] * self.e[0] + self.e[1] * self.e[1] + self.e[2] * self.e[2] }
    pub fn length(&self) -> f64 { self.length_squared().sqrt() }
    pub fn near_zero(&self) -> bool { let s = 1e-8; self.e[0].abs() < s && self.e[1].abs() < s && self.e[2].abs() < s }
    pub fn neg(&self) -> Self { Self::new(-self.e[0], -self.e[1], -self.e[2]) }
    pub fn min(&self, o: Vec3) -> Self { Self::new(self.e[0].min(o.e[0]), self.e[1].min(o.e[1]), self.e[2].min(o.e[2])) }
    pub fn max(&self, o: Vec3) -> Self { Self::new(self.e[0].max(o.e[0]), self.e[1].max(o.e[1]), self.e[2].max(o.e[2])) }
    pub fn random() -> Self { Self::new(random_double(), random_double(), random_double()) }
    pub fn random_range(min: f64, max: f64) -> Self {
        Self::new(random_double_range(min, max), random_double_range(min, max), random_double_range(min, max))
    }
}

impl Index<usize> for Vec3 { type Output = f64; fn index(&self, i: usize) -> &f64 { &self.e[i] } }
impl IndexMut<usize> for Vec3 { fn index_mut(&mut self, i: usize) -> &mut f64 { &mut self.e[i] } }
impl Neg for Vec3 { type Output = Vec3; fn neg(self) -> Vec3 { Vec3::new(-self.e[0], -self.e[1], -self.e[2]) } }
impl Add for Vec3 { type Output = Vec3; fn add(self, o: Vec3) -> Vec3 { Vec3::new(self.e[0] + o.e[0], self.e[1] + o.e[1], self.e[2] + o.e[2]) } }
impl Sub for Vec3 { type Output = Vec3; fn sub(self, o: Vec3) -> Vec3 { Vec3::new(self.e[0] - o.e[0], self.e[1] - o.e[1], self.e[2] - o.e[2]) } }
impl AddAssign for Vec3 { fn add_assign(&mut self, o: Vec3) { self.e[0] += o.e[0]; self.e[1] += o.e[1]; self.e[2] += o.e[2]; } }
impl Mul<f64> for Vec3 { type Output = Vec3; fn mul(self, t: f64) -> Vec3 { Vec3::new(self.e[0] * t, self.e[1] * t, self.e[2] * t) } }
impl Mul<Vec3> for f64 { type Output = Vec3; fn mul(self, v: Vec3) -> Vec3 { Vec3::new(self * v.e[0], self * v.e[1], self * v.e[2]) } }
impl Mul<Vec3> for Vec3 { type Output = Vec3; fn mul(self, o: Vec3) -> Vec3 { Vec3::new(self.e[0] * o.e[0], self.e[1] * o.e[1], self.e[2] * o.e[2]) } }
impl Div<f64> for Vec3 { type Output = Vec3; fn div(self, t: f64) -> Vec3 { self * (1.0 / t) } }

pub fn dot(a: &Vec3, b: &Vec3) -> f64 { a.e[0] * b.e[0] + a.e[1] * b.e[1] + a.e[2] * b.e[2] }
pub fn cross(a: &Vec3, b: &Vec3) -> Vec3 {
    Vec3::new(a.e[1] * b.e[2] - a.e[2] * b.e[1], a.e[2] * b.e[0] - a.e[0] * b.e[2], a.e[0] * b.e[1] - a.e[1] * b.e[0])
}
pub fn unit_vector(v: Vec3) -> Vec3 { v / v.length() }
pub fn reflect(v: &Vec3, n: &Vec3) -> Vec3 { *v - 2.0 * dot(v, n) * *n }
pub fn refract(uv: &Vec3, n: &Vec3, etai_over_etat: f64) -> Vec3 {
    let cos_theta = dot(&-*uv, n).min(1.0);
    let r_out_perp = etai_over_etat * (*uv + cos_theta * *n);
    let r_out_parallel = -((1.0 - r_out_perp.length_squared()).abs().sqrt()) * *n;
    r_out_perp + r_out_parallel
}
pub fn random_unit_vector() -> Vec3 {
    loop {
        let p = Vec3::random_range(-1.0, 1.0);
        let lensq = p.length_squared();
        if 1e-160 < lensq && lensq <= 1.0 { return p / lensq.sqrt(); }
    }
}
pub fn random_in_unit_disk() -> Vec3 {
    loop {
        let p = Vec3::new(random_double_range(-1.0, 1.0), random_double_range(-1.0, 1.0), 0.0);
        if p.length_squared() < 1.0 { return p; }
    }
}
pub fn random_cosine_direction() -> Vec3 {
    let r1 = random_double();
    let r2 = random_double();
    let phi = 2.0 * PI * r1;
    Vec3::new(phi.cos() * r2.sqrt(), phi.sin() * r2.sqrt(), (1.0 - r2).sqrt())
}
