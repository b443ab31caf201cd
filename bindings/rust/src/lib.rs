//! Binding of `include/rtw.h` for the reference crate.  SOURCE ONLY (no Rust toolchain in the build
//! image).  Mirrors `Viewport::render` (Rust/src/viewport.rs:430) with the integrator chosen by enum:
//! a host closure cannot run on the GPU.
use std::os::raw::c_void;

#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct RtwCamera { pub origin: [f32; 3], pub u: [f32; 3], pub v: [f32; 3], pub pixel00: [f32; 3],
    pub delta_u: [f32; 3], pub delta_v: [f32; 3], pub lens_radius: f32, pub time0: f32, pub shutter: f32 }

#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct RtwSphere { pub center: [f32; 3], pub radius: f32, pub velocity: [f32; 3], pub col_mod: [f32; 3],
    pub tex_color: [f32; 3], pub metallicness: f32, pub opacity: f32, pub ir: f32, pub emitted: [f32; 3], pub tex: i32 }

#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct RtwTexture { pub row: u32, pub col: u32, pub texel_offset: u32, pub emit_tex: u32 }   // emit_tex: 1 + index of Rust2's emission image, 0 = none

/// `Quad` (objects/quad.rs:8-20) + its Material; normal / d / w of Quad::new are recomputed by the library.
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct RtwQuad { pub origin: [f32; 3], pub u: [f32; 3], pub v: [f32; 3], pub velocity: [f32; 3], pub tex_color: [f32; 3],
    pub metallicness: f32, pub opacity: f32, pub ir: f32, pub emitted: [f32; 3], pub tex: i32 }

/// `Instance` (objects/instance.rs:27-38): member ranges into the scene's instance pools; medium 1 = const_density.
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct RtwInstance { pub first_sphere: u32, pub n_spheres: u32, pub first_quad: u32, pub n_quads: u32,
    pub translation: [f32; 3], pub rotation: [f32; 3], pub density: f32, pub medium: u32 }

#[repr(C)]
pub struct RtwScene { pub spheres: *const RtwSphere, pub textures: *const RtwTexture, pub texels: *const f32,
    pub n_spheres: u32, pub n_textures: u32, pub n_texels: u32, pub background: [f32; 3],
    pub quads: *const RtwQuad, pub instances: *const RtwInstance, pub inst_spheres: *const RtwSphere, pub inst_quads: *const RtwQuad,
    pub n_quads: u32, pub n_instances: u32, pub n_inst_spheres: u32, pub n_inst_quads: u32 }

#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct RtwParams { pub width: u32, pub height: u32, pub samples: u32, pub depth: u32, pub gamma: f32,
    pub mint: f32, pub maxt: f32, pub integrator: u32, pub sampler: u32, pub accel: u32, pub flags: u32,
    pub seed: u64, pub row_block: u32, pub part_index: u32, pub part_count: u32, pub reserved: u32 }

#[repr(C)] #[derive(Clone, Copy, Default, Debug)]
pub struct RtwStats { pub camera_rays: u64, pub segments: u64, pub sphere_tests: u64, pub node_tests: u64,
    pub nan_pixels: u32, pub rows: u32, pub kernel_ms: f32, pub total_ms: f32,
    pub phase_steps: [u64; 6], pub phase_lanes: [u64; 6], pub quad_tests: u64,
    pub enqueue_ms: f32, pub start_ms: f32 }      // ABI v4: the call's timeline (rtw.h)

#[repr(C)] pub struct RtwCtx { _private: [u8; 0] }
#[repr(C)] pub struct RtwMgpu { _private: [u8; 0] }

#[repr(u32)] #[derive(Clone, Copy)]
pub enum Integrator { Gradient = 0, BgColor = 1, Normal = 2, Flag = 3, Rust2 = 4 }   // ray_color.rs:12-92; Rust2/src/viewport/ray_color.rs:12-37
#[repr(u32)] #[derive(Clone, Copy)]
pub enum Sampler { Row = 0, Stratified = 1, Centres = 2, NoRand = 3 }      // viewport.rs:270-305, 430-516

extern "C" {
    fn rtw_ctx_create(device: i32, out: *mut *mut RtwCtx) -> i32;
    fn rtw_ctx_destroy(ctx: *mut RtwCtx);
    fn rtw_ctx_set_scene(ctx: *mut RtwCtx, scene: *const RtwScene, t_begin: f32, t_end: f32) -> i32;
    fn rtw_ctx_render(ctx: *mut RtwCtx, cam: *const RtwCamera, p: *const RtwParams, out_rgb: *mut c_void, st: *mut RtwStats) -> i32;
    fn rtw_ctx_render_multi(ctx: *mut RtwCtx, cam: *const RtwCamera, p: *const RtwParams, fps: f32, start_frame: u32, n_frames: u32,
                            out_rgb: *mut c_void, st: *mut RtwStats) -> i32;
    fn rtw_strerror(status: i32) -> *const std::os::raw::c_char;
    fn rtw_part_rows(height: u32, row_block: u32, part_index: u32, part_count: u32) -> u32;
    fn rtw_ctx_set_option(ctx: *mut RtwCtx, key: u32, value: f64) -> i32;
    fn rtw_mgpu_create(devices: *const i32, n: u32, out: *mut *mut RtwMgpu) -> i32;
    fn rtw_mgpu_destroy(m: *mut RtwMgpu);
    fn rtw_mgpu_set_scene(m: *mut RtwMgpu, scene: *const RtwScene, t_begin: f32, t_end: f32) -> i32;
    fn rtw_mgpu_render(m: *mut RtwMgpu, cam: *const RtwCamera, p: *const RtwParams, out_rgb: *mut c_void,
                       per_device: *mut RtwStats, total: *mut RtwStats) -> i32;
}

#[derive(Debug)]
pub struct RtwError(pub i32);
impl std::fmt::Display for RtwError {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        let s = unsafe { std::ffi::CStr::from_ptr(rtw_strerror(self.0)) };
        write!(f, "rtw: {}", s.to_string_lossy())
    }
}
fn check(rc: i32) -> Result<(), RtwError> { if rc == 0 { Ok(()) } else { Err(RtwError(rc)) } }

/// One GPU.  `Renderer::new(0)?.set_scene(..)?.render(..)` replaces `viewport.render(&ray_color, &scene)`.
pub struct Renderer { ctx: *mut RtwCtx }
impl Renderer {
    pub fn new(device: i32) -> Result<Self, RtwError> {
        let mut ctx = std::ptr::null_mut();
        check(unsafe { rtw_ctx_create(device, &mut ctx) })?;
        Ok(Self { ctx })
    }
    /// == Scene::new_sphere(spheres) (viewport.rs:90-105); `texels` is the concatenation of every
    /// ImageTexture.img (texture.rs:21-27), `textures[i]` = {row, col, offset}.
    pub fn set_scene(&mut self, spheres: &[RtwSphere], textures: &[RtwTexture], texels: &[[f32; 3]],
                     background: [f32; 3], t_begin: f32, t_end: f32) -> Result<(), RtwError> {
        let sc = RtwScene { spheres: spheres.as_ptr(), textures: textures.as_ptr(), texels: texels.as_ptr() as *const f32,
            n_spheres: spheres.len() as u32, n_textures: textures.len() as u32, n_texels: texels.len() as u32, background,
            quads: std::ptr::null(), instances: std::ptr::null(), inst_spheres: std::ptr::null(), inst_quads: std::ptr::null(),
            n_quads: 0, n_instances: 0, n_inst_spheres: 0, n_inst_quads: 0 };
        check(unsafe { rtw_ctx_set_scene(self.ctx, &sc, t_begin, t_end) })
    }
    /// == Scene::new(spheres, quads, instances) (viewport.rs:122-135).  `inst_spheres` / `inst_quads` are the member pools
    /// the instances' ranges index (Instance{spheres, quads} flattened in instance order).
    pub fn set_scene_full(&mut self, spheres: &[RtwSphere], quads: &[RtwQuad], instances: &[RtwInstance],
                          inst_spheres: &[RtwSphere], inst_quads: &[RtwQuad], textures: &[RtwTexture], texels: &[[f32; 3]],
                          background: [f32; 3], t_begin: f32, t_end: f32) -> Result<(), RtwError> {
        let sc = RtwScene { spheres: spheres.as_ptr(), textures: textures.as_ptr(), texels: texels.as_ptr() as *const f32,
            n_spheres: spheres.len() as u32, n_textures: textures.len() as u32, n_texels: texels.len() as u32, background,
            quads: quads.as_ptr(), instances: instances.as_ptr(), inst_spheres: inst_spheres.as_ptr(), inst_quads: inst_quads.as_ptr(),
            n_quads: quads.len() as u32, n_instances: instances.len() as u32,
            n_inst_spheres: inst_spheres.len() as u32, n_inst_quads: inst_quads.len() as u32 };
        check(unsafe { rtw_ctx_set_scene(self.ctx, &sc, t_begin, t_end) })
    }
    /// Tuning knobs (`RTW_OPT_*` of rtw.h: 1 chunk length, 2 sample bank GiB, 3 LDS geometry, 4 workgroups per CU, 5 list-walk
    /// threshold); none of them changes the image.
    pub fn set_option(&mut self, key: u32, value: f64) -> Result<(), RtwError> { check(unsafe { rtw_ctx_set_option(self.ctx, key, value) }) }
    /// -> `Img`-shaped rows ([height][width] of Rgb<f32>), gamma-corrected, unclamped (viewport.rs:301).
    pub fn render(&mut self, cam: &RtwCamera, p: &RtwParams) -> Result<(Vec<Vec<[f32; 3]>>, RtwStats), RtwError> {
        // a row partition (RtwParams.part_count > 1) returns only the rows it owns, compactly
        let rows = unsafe { rtw_part_rows(p.height, p.row_block, p.part_index, p.part_count) } as usize;
        let mut flat = vec![[0f32; 3]; (p.width as usize) * rows];
        let mut st = RtwStats::default();
        check(unsafe { rtw_ctx_render(self.ctx, cam, p, flat.as_mut_ptr() as *mut c_void, &mut st) })?;
        let rows = flat.chunks(p.width as usize).take(st.rows as usize).map(|r| r.to_vec()).collect();
        Ok((rows, st))
    }
}
impl Renderer {
    /// == `render_multi` (viewport.rs:249-269): frames start_frame .. start_frame + n_frames at time frame / fps.
    pub fn render_multi(&mut self, cam: &RtwCamera, p: &RtwParams, fps: f32, start_frame: u32, n_frames: u32)
        -> Result<Vec<Vec<Vec<[f32; 3]>>>, RtwError> {
        let per = (p.width as usize) * (p.height as usize);
        let mut flat = vec![[0f32; 3]; per * n_frames as usize];
        check(unsafe { rtw_ctx_render_multi(self.ctx, cam, p, fps, start_frame, n_frames, flat.as_mut_ptr() as *mut c_void, std::ptr::null_mut()) })?;
        Ok(flat.chunks(per).map(|f| f.chunks(p.width as usize).map(|r| r.to_vec()).collect()).collect())
    }
}
impl Drop for Renderer { fn drop(&mut self) { unsafe { rtw_ctx_destroy(self.ctx) } } }

/// All GPUs of a node from one host thread: the fork / ordered join of `async_render`'s row tasks
/// (viewport.rs:236-244; rayon: Rust2/src/viewport.rs:119-122) with GPUs in place of worker threads.  The host owns the
/// frame (`Img`); every GPU copies its interleaved 8-row blocks straight into their image rows.
pub struct MultiRenderer { m: *mut RtwMgpu, n: usize }
impl MultiRenderer {
    pub fn new(devices: &[i32]) -> Result<Self, RtwError> {
        let mut m = std::ptr::null_mut();
        check(unsafe { rtw_mgpu_create(devices.as_ptr(), devices.len() as u32, &mut m) })?;
        Ok(Self { m, n: devices.len() })
    }
    pub fn set_scene(&mut self, sc: &RtwScene, t_begin: f32, t_end: f32) -> Result<(), RtwError> {
        check(unsafe { rtw_mgpu_set_scene(self.m, sc, t_begin, t_end) })
    }
    pub fn render(&mut self, cam: &RtwCamera, p: &RtwParams) -> Result<(Vec<Vec<[f32; 3]>>, Vec<RtwStats>), RtwError> {
        let mut flat = vec![[0f32; 3]; (p.width as usize) * (p.height as usize)];
        let mut per = vec![RtwStats::default(); self.n];
        check(unsafe { rtw_mgpu_render(self.m, cam, p, flat.as_mut_ptr() as *mut c_void, per.as_mut_ptr(), std::ptr::null_mut()) })?;
        Ok((flat.chunks(p.width as usize).map(|r| r.to_vec()).collect(), per))
    }
}
impl Drop for MultiRenderer { fn drop(&mut self) { unsafe { rtw_mgpu_destroy(self.m) } } }

// In the reference crate, next to `impl Viewport` (viewport.rs:307):
//
// impl Viewport {
//     pub fn render_gpu(&self, ray_color: Integrator, scene: &Scene) -> Result<Img, RtwError> {
//         let cam = RtwCamera { origin: self.origin.into(), u: self.u.into(), v: self.v.into(),
//             pixel00: self.upper_left_corner.into(), delta_u: self.p_delta_u.into(), delta_v: self.p_delta_v.into(),
//             lens_radius: self.lens_radius, time0: self.frame as f32 / self.fps, shutter: self.shutter_speed };
//         let spheres: Vec<RtwSphere> = scene.spheres.iter().map(|s| RtwSphere {
//             center: s.origin.into(), radius: s.radius, velocity: s.velocity.into(), col_mod: s.col_mod.into(),
//             tex_color: /* 1x1 texel, or [1.;3] with tex = index */, metallicness: s.mat.metallicness,
//             opacity: s.mat.opacity, ir: s.mat.ir, emitted: s.mat.emmited.into(), tex: -1 }).collect();
//         let p = RtwParams { width: self.width as u32, height: self.height as u32, samples: self.samples as u32,
//             depth: self.depth as u32, gamma: self.gamma, mint: 0.001, maxt: 100000.0, integrator: ray_color as u32,
//             sampler: Sampler::Stratified as u32, accel: 1, flags: 0, seed: 1, row_block: 8, part_index: 0, part_count: 1, reserved: 0 };
//         let mut r = Renderer::new(0)?;
//         r.set_scene(&spheres, &[], &[], scene.background_color.into(), cam.time0, cam.time0 + cam.shutter)?;
//         Ok(r.render(&cam, &p)?.0.into_iter().map(|row| row.into_iter().map(Rgb).collect()).collect())
//     }
// }
