// ref_driver.cpp -- thin C-ABI driver over the REFERENCE's own C++ objects (oracle/_ref/librtw_ref.so).
//
// TEST INFRASTRUCTURE, NOT PRODUCT.  Built only where /root/reference exists (this container);
// the .so travels to the GPU box, the reference sources do not.
//
// What is the reference's and what is ours:
//   reference, compiled where it lies (oracle/Makefile REF_SRCS): Sphere::collisionNormal
//     (C++/src/sphere.cpp:12-36) -> Material::onHit (C++/headers/materials.h:85-122,
//     C++/src/materials.cpp:4-18), vec3 (C++/src/vec3.cpp), Ray, Hit, RGB, rand()-based
//     random_double (C++/headers/defines.h:22-25), Camera (C++/headers/viewport.h:13-70, header-only).
//   ours (this file): the pixel loops and the integrators, restated from
//     C++/src/viewport.cpp:4-105 (Render / Render_no_rand), C++/src/tests.cpp:150-175 (ray_colorD),
//     :233-272 (ray_color_small), :76-97 (ray_colorSc), :275-294 (s_test) and the P3 writer
//     C++/src/ppm_writer.cpp:12-27 + C++/src/RGB.cpp:16-20 -- because viewport.cpp includes the
//     empty lib/indicators submodule and is therefore unbuildable here (no stand-in is written).
//   The restated loops are pinned by the md5 of the two s_test images the real viewport.cpp
//   produced (SURVEY.md 8c / BASELINE.md 3): tests/test_oracle_golden.py::test_ref_s_test_md5.
#include "viewport.h"   // reference Camera/Viewport declarations (header-only parts used)
#include "sphere.h"
#include "scene.h"
#include "hit.h"
#include "rtw.h"

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <sstream>

namespace {

using color = RGB_float;

// C++/src/tests.cpp:150-175
RGB_float ray_colorD(const Ray &r, const Scene &scene, uint depth, uint max_depth, uint64_t *segments) {
    if (depth >= max_depth) return RGB_float(0, 0, 0);
    float mint = 0.001, maxt = 1000;
    Hit min_hit = NO_HIT;
    ++*segments;
    for (const auto &s : scene.spheres) {
        Hit hit = s.collisionNormal(r, mint, maxt);
        if (hit.isHit() && (!min_hit.isHit() || hit.t < min_hit.t)) min_hit = hit;
    }
    if (min_hit.isHit()) return ray_colorD(min_hit.next, scene, depth + 1, max_depth, segments) * min_hit.col_mod;
    auto t = 0.5 * (r.direction.unit_vector().y + 1.0);
    return (1.0 - t) * color(1.0, 1.0, 1.0) + t * color(0.5, 0.7, 1.0);
}

// C++/src/tests.cpp:233-272
RGB_float ray_color_small(const Ray &r, const Scene &scene, uint depth, uint max_depth) {
    if (depth >= max_depth) return RGB_float(0, 0, 0);
    float mint = 0.001, maxt = 1000.0;
    Hit min_hit = NO_HIT;
    bool is_rand = false;
    for (const auto &s : scene.spheres) {
        Hit hit = s.collisionNormal(r, mint, maxt);
        if (hit.isHit() && (!min_hit.isHit() || hit.t < min_hit.t)) {
            min_hit = hit;
            is_rand = s.material.metallicness != 1.0;
        }
    }
    if (is_rand) return color(1.0, 1.0, 0.0);
    if (min_hit.isHit()) return ray_color_small(min_hit.next, scene, depth + 1, max_depth) * min_hit.col_mod;
    return color(0.0, 0.0, 1.0);
}

// C++/src/tests.cpp:76-97
RGB_float ray_colorSc(const Ray &r, const Scene &scene) {
    float mint = 0, maxt = 1000;
    bool any = false; Hit best;
    for (const auto &s : scene.spheres) {
        Hit hit = s.collisionNormal(r, mint, maxt);
        if (hit.isHit() && (!any || hit.t < best.t)) { best = hit; any = true; }
    }
    if (any) return RGB_float(best.normal.x + 1, best.normal.y + 1, best.normal.z + 1) * 0.5;
    auto t = 0.5 * (r.direction.unit_vector().y + 1.0);
    return (1.0 - t) * color(1.0, 1.0, 1.0) + t * color(0.5, 0.7, 1.0);
}

Scene make_scene(const RtwScene *sc) {
    Scene scene;
    for (uint32_t i = 0; i < sc->n_spheres; i++) {
        const RtwSphere &s = sc->spheres[i];
        // C++ has one col_mod and no texture: use the albedo the Rust hit would report (texel * col_mod)
        vec3 cm(s.tex_color[0] * s.col_mod[0], s.tex_color[1] * s.col_mod[1], s.tex_color[2] * s.col_mod[2]);
        scene.spheres.push_back(Sphere(vec3(s.center[0], s.center[1], s.center[2]), s.radius,
                                       materials::Material(s.metallicness, s.opacity, s.ir), cm));
    }
    return scene;
}

// C++/src/ppm_writer.cpp:12-27 + C++/src/RGB.cpp:16-20 (truncating int(255*c))
std::string ppm_p3(const std::vector<std::vector<RGB_float>> &vec) {
    std::ostringstream os;
    os << "P3\n" << vec[0].size() << ' ' << vec.size() << "\n255\n";
    for (const auto &row : vec) {
        for (const auto &c : row) os << RGB_int(c) << "  ";
        os << '\n';
    }
    return os.str();
}

} // namespace

extern "C" {

// One Sphere::collisionNormal call.  out = {hit?, t, normal xyz, point xyz, next.origin xyz, next.direction xyz}
int rtw_ref_sphere_hit(const float center[3], float radius, const float mat3[3], const float ray_o[3],
                       const float ray_d[3], float mint, float maxt, unsigned rand_seed, double out[14]) {
    srand(rand_seed);
    Sphere s(vec3(center[0], center[1], center[2]), radius, materials::Material(mat3[0], mat3[1], mat3[2]), vec3(1, 1, 1));
    Ray r(vec3(ray_o[0], ray_o[1], ray_o[2]), vec3(ray_d[0], ray_d[1], ray_d[2]));
    Hit h = s.collisionNormal(r, mint, maxt);
    out[0] = h.isHit() ? 1.0 : 0.0; out[1] = h.t;
    out[2] = h.normal.x; out[3] = h.normal.y; out[4] = h.normal.z;
    out[5] = h.point.x; out[6] = h.point.y; out[7] = h.point.z;
    out[8] = h.next.origin.x; out[9] = h.next.origin.y; out[10] = h.next.origin.z;
    out[11] = h.next.direction.x; out[12] = h.next.direction.y; out[13] = h.next.direction.z;
    return 0;
}

// The reference Camera (header-only ctor, C++/headers/viewport.h:24-46) -> RtwCamera fields.
int rtw_ref_camera(uint32_t width, float aspect, float vfov, const float origin[3], const float vup[3],
                   const float direction[3], float lens_radius, RtwCamera *cam, uint32_t *height) {
    Camera c(width, aspect, vfov, vec3(origin[0], origin[1], origin[2]), vec3(vup[0], vup[1], vup[2]),
             vec3(direction[0], direction[1], direction[2]), lens_radius);
    std::memset(cam, 0, sizeof *cam);
    const vec3 *src[6] = { &c.origin, &c.u, &c.v, &c.pixel00_loc, &c.pixel_delta_u, &c.pixel_delta_v };
    float *dst[6] = { cam->origin, cam->u, cam->v, cam->pixel00, cam->delta_u, cam->delta_v };
    for (int k = 0; k < 6; k++) { dst[k][0] = src[k]->x; dst[k][1] = src[k]->y; dst[k][2] = src[k]->z; }
    cam->lens_radius = c.lens_radius;
    *height = c.height;
    return 0;
}

// s_test (C++/src/tests.cpp:275-294): default Viewport() camera 300 x 200, Render_no_rand with
// ray_color_small; writes the two P3 texts (caller frees with rtw_ref_free).
int rtw_ref_s_test(char **ppm_control, size_t *len_control, char **ppm_glass, size_t *len_glass) {
    srand(1);
    Scene scene;
    scene.spheres = { Sphere(vec3(0, 0, -1), 0.5, materials::metallicM, vec3(1, 1, 1)),
                      Sphere(vec3(0, -100.5, -1), 100, materials::scatterM, vec3(0.8, 0.5, 1.0)) };
    Viewport viewport = Viewport();
    for (int pass = 0; pass < 2; pass++) {
        if (pass == 1) scene.spheres[0].material = materials::glass;
        const Camera &cam = viewport.cam;
        std::vector<std::vector<RGB_float>> img(cam.height);
        float inv_g = 1 / viewport.gamma;
        for (int j = 0; j < (int)cam.height; j++) {                 // viewport.cpp:82-101
            std::vector<RGB_float> row(cam.width);
            for (int i = 0; i < (int)cam.width; ++i) {
                RGB_float pixel_color = RGB_float(0, 0, 0);
                Ray r(cam.origin, cam.pixel00_loc + i * cam.pixel_delta_u + j * cam.pixel_delta_v);
                pixel_color += ray_color_small(r, scene, 0, viewport.max_reflections);
                row[i] = (pixel_color / viewport.samples_per_pixel).gamma(inv_g);
            }
            img[j] = row;
        }
        std::string s = ppm_p3(img);
        char *buf = (char *)malloc(s.size());
        std::memcpy(buf, s.data(), s.size());
        if (pass == 0) { *ppm_control = buf; *len_control = s.size(); } else { *ppm_glass = buf; *len_glass = s.size(); }
    }
    return 0;
}
void rtw_ref_free(void *p) { free(p); }

// Viewport::Render restated over the reference objects (C++/src/viewport.cpp:28-53), serial, rand().
// integrator: 0 ray_colorD, 2 ray_colorSc (normal), 3 ray_color_small (flag).  sampler 3 = Render_no_rand.
// out: [height][width][3] double, gamma-corrected like the reference.
int rtw_ref_render(const RtwCamera *cam, const RtwScene *sc, const RtwParams *p, unsigned rand_seed,
                   double *out, uint64_t *segments_out, double *seconds_out) {
    srand(rand_seed);
    Scene scene = make_scene(sc);
    vec3 origin(cam->origin[0], cam->origin[1], cam->origin[2]);
    vec3 u(cam->u[0], cam->u[1], cam->u[2]), v(cam->v[0], cam->v[1], cam->v[2]);
    vec3 p00(cam->pixel00[0], cam->pixel00[1], cam->pixel00[2]);
    vec3 du(cam->delta_u[0], cam->delta_u[1], cam->delta_u[2]), dv(cam->delta_v[0], cam->delta_v[1], cam->delta_v[2]);
    float inv_g = 1 / p->gamma;
    uint64_t segments = 0;
    auto t0 = std::chrono::steady_clock::now();
    for (int j = 0; j < (int)p->height; j++) {
        for (int i = 0; i < (int)p->width; ++i) {
            RGB_float pixel_color = RGB_float(0, 0, 0);
            uint n = p->sampler == RTW_SAMPLER_NO_RAND ? 1 : p->samples;
            for (uint s = 0; s < n; s++) {
                Ray r;
                if (p->sampler == RTW_SAMPLER_NO_RAND) {
                    r = Ray(origin, p00 + i * du + j * dv);
                } else {
                    auto random_point = vec3::random_in_unit_disk();
                    r = Ray(origin + (random_point.x * u + random_point.y * v) * cam->lens_radius,
                            p00 + (i + random_double()) * du + (j + random_double()) * dv);
                }
                if (p->integrator == RTW_INTEGRATOR_NORMAL) pixel_color += ray_colorSc(r, scene);
                else if (p->integrator == RTW_INTEGRATOR_FLAG) pixel_color += ray_color_small(r, scene, 0, p->depth);
                else pixel_color += ray_colorD(r, scene, 0, p->depth, &segments);
            }
            RGB_float c = (pixel_color / n).gamma(inv_g);
            double *o = out + 3 * ((size_t)j * p->width + i);
            o[0] = c.R; o[1] = c.G; o[2] = c.B;
        }
    }
    auto t1 = std::chrono::steady_clock::now();
    if (segments_out) *segments_out = segments;
    if (seconds_out) *seconds_out = std::chrono::duration<double>(t1 - t0).count();
    return 0;
}

} // extern "C"
