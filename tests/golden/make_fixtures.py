"""Regenerates tests/golden/*.json|*.npy.  Runs ONLY in the authoring container (needs /root/reference
and the oracle/_ref build); the fixtures it writes are data -- inputs and expected outputs -- and are
committed so that the GPU box and later rounds never need the reference tree.

  cerr_trace.json     Rust/cerr (stderr of an older C++ build of the 10x10 glass scene,
                      Rust/src/viewport/glass_tests.rs:102-142) parsed into per-pixel bounce records.
  s_test.npz          the two RNG-free 300x200 images of s_test (C++/src/tests.cpp:275-294) as produced
                      by oracle/_ref (md5 of the P3 text == the md5 SURVEY.md 8c recorded from the real
                      viewport.cpp), stored as one bit per pixel (yellow=1 / blue=0).
  ref_sphere_hits.json  Sphere::collisionNormal outputs of the reference's own objects (oracle/_ref)
                      for seeded random rays: t, normal, point, next direction (RNG-free materials only).
  ref_camera.json     the reference C++ Camera for four configurations.
  ref_images.npz      two images the reference itself rendered and checked in, reduced to 16x16-block means of
                      the 8-bit values (25x25x3 each) + the count of pure-black pixels:
                        Rust/Presentation.png  presentation_image (Rust/src/main.rs:89-419): 2500 spp, quads,
                                               a light, a smoke box, a glass pane, ray_color_bg_color
                        Rust/First frame.png   main()'s seven spheres (main.rs:427-545): 100 spp, depth 100
                      Both were rendered with the reference's unseeded ThreadRng, so they pin the restatement
                      statistically (block means), not bit for bit.  Plus, verbatim, Rust/test.png (256x256), the
                      image the reference's writer test saves (write_img.rs:33-58): an exact golden for the 8-bit
                      quantiser.
"""
import ctypes as C
import hashlib
import json
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"


def parse_cerr():
    lines = open(os.path.join(REF, "Rust", "cerr")).read().splitlines()
    pixels, cur, ev = [], None, None
    fl = lambda s: [float(x) for x in s.split()]
    for ln in lines:
        ln = ln.strip()
        if not ln:
            continue
        m = re.match(r"x: (\d+) y: (\d+)", ln)
        if m:
            cur = {"x": int(m.group(1)), "y": int(m.group(2)), "bounces": []}
            pixels.append(cur)
            continue
        m = re.match(r"u: (\S+) v: (\S+)", ln)
        if m:
            cur["u"], cur["v"] = float(m.group(1)), float(m.group(2))
            continue
        m = re.match(r"D: (\d+)", ln)
        if m:
            ev = {"depth": int(m.group(1))}
            cur["bounces"].append(ev)
            continue
        m = re.match(r"ff: (\d) can refract: (\d) ref_ratio: (\S+)", ln)
        if m:
            ev.update(kind="dielectric", front_face=int(m.group(1)), can_refract=int(m.group(2)), ratio=float(m.group(3)))
            continue
        if ln.startswith("ud "):
            a = fl(ln.replace("ud", "").replace("hn", ""))
            ev["unit_dir"], ev["facing_normal"] = a[:3], a[3:]
            continue
        if ln == "reflect":
            ev.setdefault("kind", "scatter_hit")      # non-dielectric onHit branch: the ground sphere
            continue
        if ln == "Sky":
            ev["kind"] = "sky"
            continue
        if ln == "Hit":
            continue
        a = fl(ln)
        if len(a) == 3:
            ev["facing_normal"] = a                   # first record of the file: `hn` printed without label
        elif len(a) == 6:
            ev["next_origin"], ev["next_dir"] = a[:3], a[3:]
    return pixels


def ref_images():
    from PIL import Image
    out = {}
    for key, rel in (("presentation", "Rust/Presentation.png"), ("first_frame", "Rust/First frame.png")):
        img = np.asarray(Image.open(os.path.join(REF, rel)).convert("RGB")).astype(np.float64)
        h, w, _ = img.shape
        assert (h, w) == (400, 400)
        out[key + "_blocks16"] = img.reshape(h // 16, 16, w // 16, 16, 3).mean(axis=(1, 3)).astype(np.float32)
        out[key + "_black_pixels"] = np.array([(img.sum(axis=2) == 0).sum()])
        print(rel, "black pixels", int(out[key + "_black_pixels"][0]), "mean", img.mean(axis=(0, 1)))
    # Rust/test.png: the output of the reference's own writer test (write_img.rs:33-58): pixel (i, j) = round(255 * (i/255, j/255, 0.25))
    wt = np.asarray(Image.open(os.path.join(REF, "Rust/test.png")).convert("RGB")).astype(np.uint8)
    assert wt.shape == (256, 256, 3)
    # separable by construction (R depends on the column, G on the row, B is constant): store the three profiles
    assert (wt[:, :, 0] == wt[0:1, :, 0]).all() and (wt[:, :, 1] == wt[:, 0:1, 1]).all() and (wt[:, :, 2] == wt[0, 0, 2]).all()
    out["write_test_r_of_column"], out["write_test_g_of_row"], out["write_test_b"] = wt[0, :, 0].copy(), wt[:, 0, 1].copy(), wt[0, 0, 2:3].copy()
    # Rust/assets/squares.png (128 x 64, five flat colours): one of the image textures the reference's texture tests load
    # (viewport/texture_test.rs); stored as the texel array ImageTexture::from_path would build (u8 / 255 per channel).
    out["squares_png_rgb"] = np.asarray(Image.open(os.path.join(REF, "Rust/assets/squares.png")).convert("RGB")).astype(np.uint8)
    np.savez_compressed(os.path.join(HERE, "ref_images.npz"), **out)


def main():
    if sys.argv[1:] == ["images"]:
        return ref_images()
    ref_images()
    from tests import oracle_binding as O
    import rtw_amd as R
    assert O.have_ref(), "build oracle/_ref first (make -C oracle)"
    L = O.ref()

    # 1. cerr
    px = parse_cerr()
    json.dump({"source": "Rust/cerr", "camera": "origin 0, dir = (-1+2u, -1+2v, -1), u=x/9, v=(9-y)/9",
               "scene": "glass (0,0,-1) r0.5 ir1.5 + ground (0,-100.5,-1) r100 scatter", "pixels": px},
              open(os.path.join(HERE, "cerr_trace.json"), "w"), indent=0)
    print("cerr:", len(px), "pixels,", sum(len(p["bounces"]) for p in px), "bounce records")

    # 2. s_test
    pa, pb = C.POINTER(C.c_char)(), C.POINTER(C.c_char)()
    la, lb = C.c_size_t(), C.c_size_t()
    L.rtw_ref_s_test(C.byref(pa), C.byref(la), C.byref(pb), C.byref(lb))
    out = {}
    for name, p, n in (("control", pa, la), ("glass", pb, lb)):
        txt = C.string_at(p, n.value)
        md5 = hashlib.md5(txt).hexdigest()
        tok = txt.split()
        assert tok[0] == b"P3"
        w, h = int(tok[1]), int(tok[2])
        rgb = np.array(tok[4:], dtype=np.int32).reshape(h, w, 3)
        yellow = (rgb == np.array([255, 255, 0])).all(axis=2)
        blue = (rgb == np.array([0, 0, 255])).all(axis=2)
        assert (yellow | blue).all()
        out[name + "_bits"] = np.packbits(yellow)
        out[name + "_md5"] = np.frombuffer(md5.encode(), dtype=np.uint8)
        out[name + "_shape"] = np.array([h, w])
        print("s_test", name, md5, "yellow", int(yellow.sum()), "blue", int(blue.sum()))
        L.rtw_ref_free(p)
    np.savez_compressed(os.path.join(HERE, "s_test.npz"), **out)

    # 3. per-hit vectors from the reference objects
    rng = np.random.default_rng(20241223)
    cases = []
    mats = {"metallic": (1.0, 0.0, 1.0), "glass": (1.0, 1.0, 1.5), "glassR": (1.0, 1.0, float(np.float32(1 / 1.5)))}
    while len(cases) < 240:
        centre = rng.uniform(-2, 2, 3).astype(np.float32)
        radius = np.float32(rng.uniform(0.2, 1.5))
        o = rng.uniform(-4, 4, 3).astype(np.float32)
        target = centre + rng.uniform(-1, 1, 3).astype(np.float32) * radius
        d = (target - o).astype(np.float32) * np.float32(rng.uniform(0.3, 3.0))
        mname = list(mats)[len(cases) % 3]
        m = mats[mname]
        res = (C.c_double * 14)()
        L.rtw_ref_sphere_hit((C.c_float * 3)(*centre), radius, (C.c_float * 3)(*m), (C.c_float * 3)(*o), (C.c_float * 3)(*d),
                             0.001, 1000.0, 1, res)
        r = list(res)
        cases.append({"centre": [float(x) for x in centre], "radius": float(radius), "material": mname, "mat3": list(m),
                      "origin": [float(x) for x in o], "dir": [float(x) for x in d], "hit": int(r[0]), "t": r[1],
                      "normal": r[2:5], "point": r[5:8], "next_dir": r[11:14]})
    json.dump({"source": "oracle/_ref: Sphere::collisionNormal (C++/src/sphere.cpp:12-36) -> Material::onHit "
                         "(C++/headers/materials.h:85-122); mint 0.001 maxt 1000; dielectric is the C++ twin's deterministic one",
               "cases": cases}, open(os.path.join(HERE, "ref_sphere_hits.json"), "w"))
    print("sphere hits:", len(cases), "cases,", sum(c["hit"] for c in cases), "hits")

    # 4. reference Camera
    cams = []
    for (w, aspect, vfov, org, vup, direction, lens) in (
            (300, 1.5, 90.0, (0, 0, 0), (0, 1, 0), (0, 0, -1), 0.0),
            (400, 400 / 225, 90.0, (0, 0, 0), (0, 1, 0), (0, 0, -1), 0.0),
            (400, 4 / 3, 120.0, (0, 0, 0), (0, -1, 0), (0, 0, -1), 0.01),
            (1920, 1920 / 1080, 20.0, (13, 2, 3), (0, 1, 0), tuple(-np.array([13, 2, 3]) / np.linalg.norm([13, 2, 3])), 0.05)):
        cam, h = R.RtwCamera(), C.c_uint32()
        L.rtw_ref_camera(w, aspect, vfov, (C.c_float * 3)(*org), (C.c_float * 3)(*vup), (C.c_float * 3)(*[float(x) for x in direction]),
                         lens, C.byref(cam), C.byref(h))
        cams.append({"width": w, "aspect": aspect, "vfov": vfov, "origin": list(map(float, org)), "vup": list(map(float, vup)),
                     "direction": [float(x) for x in direction], "lens_radius": lens, "height": h.value,
                     **{f: [float(x) for x in getattr(cam, f)] for f in ("origin", "u", "v", "pixel00", "delta_u", "delta_v")}})
    json.dump({"source": "oracle/_ref: Camera (C++/headers/viewport.h:24-46); unit direction, so focal_length == 1",
               "cameras": cams}, open(os.path.join(HERE, "ref_camera.json"), "w"), indent=0)
    print("cameras:", len(cams))


if __name__ == "__main__":
    main()
