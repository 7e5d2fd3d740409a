// camera_gpu.rs -- `Camera::render` (camera.rs:78-158) with its sampling loop on the GPU.
//
// Add to camera.rs (the fields are private to that module).  The rayon block of camera.rs:99-149 -- per row, per
// pixel, per stratum: get_ray, ray_color -- becomes one rt_render call; initialize(), the PPM header and the row
// printing stay as they are.  Every random draw of the library comes from a counter-based stream keyed by
// (seed, pixel, stratum) instead of thread_rng, so the picture no longer depends on the thread schedule.
use crate::gpu_scene::{Describe, GpuScene, RtError};
use mi355rt_sys as sys;

impl Camera {
    fn to_rt_camera(&self, russian_roulette: bool) -> sys::rt_camera {
        let v = |p: crate::vec3::Vec3| [p.x(), p.y(), p.z()];
        sys::rt_camera {
            aspect_ratio: self.aspect_ratio, image_width: self.image_width, samples_per_pixel: self.samples_per_pixel,
            max_depth: self.max_depth, russian_roulette: russian_roulette as i32, background: v(self.background), vfov: self.vfov,
            lookfrom: v(self.lookfrom), lookat: v(self.lookat), vup: v(self.vup), defocus_angle: self.defocus_angle,
            focus_dist: self.focus_dist,
        }
    }

    /// Same output as `render` (a P3 image on stdout, progress on stderr), traced by the library.
    /// `lights = None` renders with the material PDF alone (scenes that predate the light list).
    pub fn render_gpu(&self, world: &dyn Describe, lights: Option<&dyn Describe>, seed: u64) -> Result<(), RtError> {
        let scene = GpuScene::new(world, lights)?;
        let cam = self.to_rt_camera(false);
        let mut d: sys::rt_camera_derived = unsafe { std::mem::zeroed() };
        let rc = unsafe { sys::rt_camera_initialize(&cam, &mut d) };
        if rc != 0 { return Err(RtError { code: rc, message: "rt_camera_initialize".into() }); }
        let (w, h) = (self.image_width as usize, d.image_height as usize);
        let mut accum = vec![0.0f64; w * h * 3]; // sum of ray_color over the strata, per pixel (camera.rs:110-131)
        let rc = unsafe { sys::rt_render(scene.raw(), &cam, seed, std::ptr::null(), accum.as_mut_ptr()) };
        if rc != 0 {
            let m = unsafe { std::ffi::CStr::from_ptr(sys::rt_last_error()) }.to_string_lossy().into_owned();
            return Err(RtError { code: rc, message: m });
        }
        let mut rgb = vec![0u8; w * h * 3]; // camera.rs:133-135 + write_color_to_string (color.rs:58-80)
        unsafe { sys::rt_resolve_rgb8(&cam, accum.as_ptr(), rgb.as_mut_ptr()); }
        println!("P3\n{} {}\n255", w, h); // camera.rs:86
        for p in rgb.chunks(3) { println!("{} {} {}", p[0], p[1], p[2]); } // camera.rs:151-154
        eprintln!("\nDone.");
        Ok(())
    }
}
