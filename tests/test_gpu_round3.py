"""Round-3 additions on the HIP path (through the C ABI): the asynchronous multi-GPU fork with its per-device timeline, Rust2's image
texture rule, guided unit lengths."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import rtw_amd as R
from tests import oracle_binding as O
from tests.test_oracle_golden import small_view

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- rtw_mgpu_render: the fork must not wait for any GPU (Rust/src/viewport.rs:236-244 spawns every row task before it awaits one) ----
def test_mgpu_fork_is_asynchronous_for_a_pageable_host_frame(gpu):
    """Three contexts and a frame in ORDINARY host memory (a numpy array: what the reference's Img, the Rust binding's Vec and
    examples/render_scene pass).  A device-to-host copy into pageable memory returns only when it is done, so a fork that enqueues
    kernels + copy per device would sit in device 0's enqueue until device 0 has rendered.  The fork is three passes now
    (prepare all, launch all, copy all, pageable frames staged through pinned memory): every device's launches must have been issued
    long before any device can have finished, and the frame is bit-equal to the one-context render."""
    scene = R.Scene.generate(R.SCENE_C2)
    cam, p = R.default_view(R.SCENE_C5)
    cam.shutter, p.samples = 0.0, 60                       # ~9 ms of kernel time per frame: far above any host-side enqueue cost
    gpu.set_scene(scene)
    full, st_full = gpu.render(cam, p)
    with R.MultiRenderer([0, 0, 0]) as m:
        m.set_scene(scene)
        m.render(cam, p)                                   # first call: allocations, occupancy queries
        frame = np.full((p.height, p.width, 3), -1.0, np.float32)          # pageable
        img, tot, per = m.render(cam, p, out=frame)
        assert np.array_equal(img, full) and tot.segments == st_full.segments
        kernel = [s.kernel_ms for s in per]
        assert min(kernel) > 1.0, kernel
        for k, s in enumerate(per):
            # issued within a fraction of the time one device needs for its share -- with the per-device copy inside the fork, device k's
            # enqueue returned only after devices 0 .. k-1 had rendered: >= k x kernel_ms
            assert 0.0 < s.enqueue_ms < 0.25 * min(kernel), (k, s.enqueue_ms, kernel)
            assert s.start_ms >= 0.0
        assert per[2].enqueue_ms >= per[1].enqueue_ms >= per[0].enqueue_ms          # one host thread, in order
        assert tot.enqueue_ms == per[2].enqueue_ms
        # pinned host memory and device memory take the direct strided copies: same frame
        import torch
        pinned = torch.full((p.height, p.width, 3), -1.0, dtype=torch.float32).pin_memory()
        _, _, per2 = m.render(cam, p, out=pinned.data_ptr())
        assert np.array_equal(pinned.numpy(), full)
        assert all(s.enqueue_ms < 0.25 * min(kernel) for s in per2)
        # ... and a ragged, non-default block height through the staging path
        q = R.RtwParams.from_buffer_copy(p)
        q.samples, q.row_block = 4, 7
        gpu_img, _ = gpu.render(cam, q)
        img3, _, _ = m.render(cam, q, out=np.empty_like(frame))
        assert np.array_equal(img3, gpu_img)


def test_single_context_timeline_fields(gpu):
    scene, cam, p = small_view(R.SCENE_C2, 96, 54, 4)
    gpu.set_scene(scene)
    _, st = gpu.render(cam, p)
    assert st.enqueue_ms > 0.0 and st.start_ms >= 0.0 and st.total_ms >= st.enqueue_ms


def test_example_prints_the_per_device_timeline(gpu):
    """examples/render_scene --devices 0,0,0: the C ABI from compiled code, frame in a std::vector (pageable)."""
    exe = os.path.join(ROOT, "examples", "render_scene")
    assert os.path.exists(exe), "make -C raytracing-in-a-weekend_amd/csrc example"
    out = subprocess.run([exe, "--devices", "0,0,0", "--spp", "8"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("device ")]
    assert len(lines) == 3 and all("enqueue" in l and "start" in l for l in lines), out.stdout


# ---- a10: Rust2's ImageTexture::color_at on the device (Rust2/src/objects/texture.rs:94-105) ---------------------------------------------
@pytest.mark.parametrize("shape", [(4, 4), (4, 8), (6, 3)])          # (height, width): square, wide, and a wide one with odd sizes
def test_rust2_image_texture_on_the_gpu(gpu, shape):
    """A textured sphere under RTW_INTEGRATOR_RUST2 takes its ColorResult by Rust2's rule -- (u * width) as usize, index x * width + y
    (transposed), emission image added -- on every kernel, bit for bit with the oracle (texel edges may fall differently between glibc's
    and ocml's atan2f / acosf: a handful of pixels), and the image is visibly NOT what the Rust/ rule gives for the same scene."""
    from tests.test_round3_cpu import rust2_texture_scene
    from tests.test_gpu_parity import render_both, ulp_diff
    from tests.test_oracle_golden import flag_params
    h, w = shape
    rng = np.random.default_rng(11)
    img = rng.uniform(0.1, 0.9, size=(h, w, 3)).astype(np.float32)
    emit = rng.uniform(0.0, 0.05, size=(5, 7, 3)).astype(np.float32)
    big = R.Sphere.new_with_texture((0.0, 0.0, -2.0), 1.0, None, R.SCATTER_M, 0)
    for k in range(3):
        big.pod.col_mod[k] = 1.0
    spheres = [big, R.Sphere.with_albedo((1.6, 0.0, -1.6), 0.4, (0.8, 0.8, 0.8), R.METALLIC_M),
               R.Sphere.with_albedo((0.0, -101.0, -2.0), 100.0, (0.5, 0.5, 0.5), R.SCATTER_M)]
    scene = R.Scene(spheres, textures=[img, emit], background=(0.7, 0.8, 1.0), emission_images={0: 1})
    cam = R.camera2_new(np.float32(96) / np.float32(54), (0, 0, 0.5), (0, 1, 0), (0, 0, -1), 80.0, 0.01)
    p = flag_params(depth=5)
    p.width, p.height, p.samples, p.integrator, p.sampler, p.gamma, p.maxt = 96, 54, 9, R.INTEGRATOR_RUST2, R.SAMPLER_CENTRES, 1.0, 1000.0
    ref, st_ref, out = render_both(gpu, scene, cam, p)
    for accel, (im, st) in out.items():
        assert st.segments == st_ref.segments, accel
        same = (ulp_diff(im, ref) <= 2).all(axis=2)
        assert same.mean() > 0.995, (accel, same.mean())
    assert np.array_equal(out[R.ACCEL_BRUTE][0], out[R.ACCEL_BVH][0]) and np.array_equal(out[R.ACCEL_BVH][0], out["tree forced"][0])
    # the same scene without the emission image and under the Rust/ lookup (gradient integrator shares nothing else: compare the DIRECT view
    # of the sphere at depth 1 with a white background, where a pixel is emmited + texel)
    p.depth, p.samples = 1, 1                                           # (one sample through the pixel centre: Rust2's fixed-centre sampler)
    scene.pod.background[0] = scene.pod.background[1] = scene.pod.background[2] = 1.0
    gpu.set_scene(scene)
    direct, _ = gpu.render(cam, p)
    centre = direct[20:34, 40:56].reshape(-1, 3)
    flat = img.reshape(-1, 3)
    # every pixel of the sphere's middle is one texel of the image plus one of the emission image
    for px in centre[::7]:
        assert any(np.abs(px - t - e).max() < 1e-6 for t in flat for e in emit.reshape(-1, 3)), px
    O_ref, _ = O.render(cam, scene, p, threads=4)
    assert (ulp_diff(direct, O_ref) <= 2).all(axis=2).mean() > 0.995


def test_second_hip_runtime_is_refused_loudly(gpu):
    """PyTorch imported AFTER the first rtw_* call brings a second libamdhip64 into the process and is blind from then on ("No HIP GPUs are
    available", gpurun_out/r02_debug_torch.log).  The library cannot stop torch from loading, but a context created in such a process fails
    with RTW_E_RUNTIME_CONFLICT and a message that names the cause, instead of rendering next to a broken torch."""
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import rtw_amd as R\n"
            "assert R.lib().rtw_hip_runtime_count() == 1\n"
            "r = R.Renderer(0); r.close()\n"                       # fine: one runtime so far
            "import torch\n"
            "print('runtimes', R.lib().rtw_hip_runtime_count())\n"
            "try:\n"
            "    R.Renderer(0)\n"
            "    print('created')\n"
            "except R.RtwError as e:\n"
            "    print('refused', e.status, e)\n") % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "runtimes 2" in out.stdout and "refused -8" in out.stdout and "import torch before" in out.stdout, out.stdout


# ---- the unit length may vary within a launch: the image may not ---------------------------------------------------------------------------
@pytest.mark.parametrize("k", [0.0, 0.5, 2.0, 64.0])
def test_guided_unit_lengths_never_change_the_image(rtw, k):
    """RTW_OPT_TAIL_UNITS cuts the last tiles of the queue into units of one sample and the ones before into shorter units than the rest; the
    bank is indexed by (tile, sample, pixel) and the resolve adds in sample order, so full frames, banded renders, row partitions, a permuted
    tile order and both kernels give the same bits, and every sample is traced exactly once."""
    scene, cam, p = small_view(R.SCENE_C2, 320, 184, 40)
    p.gamma, p.depth = 1.0, 6
    with rtw.Renderer(0) as r:
        r.set_scene(scene)
        p.accel = R.ACCEL_BVH
        ref, st_ref = r.render(cam, p)                                # one unit length (the default)
        r.set_option(R.OPT_TAIL_UNITS, k)
        for chunk in (0, 12, 5):
            r.set_option(R.OPT_CHUNK_LEN, chunk)
            for accel in (R.ACCEL_BVH, R.ACCEL_BRUTE):
                p.accel = accel
                img, st = r.render(cam, p)
                assert st.camera_rays == 320 * 184 * 40 and st.segments == st_ref.segments and np.array_equal(img, ref), (k, chunk, accel)
        p.accel = R.ACCEL_BVH
        for order in (2, 5):
            r.set_option(R.OPT_TILE_ORDER, order)
            img, _ = r.render(cam, p)
            assert np.array_equal(img, ref), order
        r.set_option(R.OPT_TILE_ORDER, 0)
        r.set_option(R.OPT_SAMPLE_BANK_GB, 0.01)                      # several bands of tile rows, each with its own regions
        img, _ = r.render(cam, p)
        assert np.array_equal(img, ref)
        r.set_option(R.OPT_SAMPLE_BANK_GB, 48)
        p.row_block, p.part_index, p.part_count = 8, 1, 3
        part, _ = r.render(cam, p)
        rows = [j for j in range(184) if (j // 8) % 3 == 1]
        assert np.array_equal(part, ref[rows])


def test_a_ray_from_infinity_hits_what_the_reference_says(gpu):
    """presentation_image at 64 spp, sample 53 of pixel (88, 19): the smoke box draws xi == 0 for its free path, ln(0) / -density is
    +inf, the scatter point -- the next ray's origin -- is not finite, and the reference's sphere test accepts such a ray (its NaN root
    passes both range tests, sphere.rs:118-121): the path goes on from the sphere and ends NaN.  A tree prunes that ray at its root; the
    GEOM builds of the traversal kernel therefore walk the list for rays that are not finite.  Until round 3 the tree rendered this
    pixel finite.  The whole frame: list walk == tree; the row of the event against the oracle."""
    scene = R.Scene.generate_geom(R.SCENE_PRESENTATION)
    cam, p = R.default_view(R.SCENE_PRESENTATION)
    p.samples, p.gamma = 64, 1.0
    gpu.set_scene(scene)
    imgs = {}
    try:
        for walk_max in (48, 0):                                 # the default (this scene walks the list), and 0: the tree forced
            gpu.set_option(R.OPT_LIST_WALK_MAX, walk_max)
            imgs[walk_max] = gpu.render(cam, p)
    finally:
        gpu.set_option(R.OPT_LIST_WALK_MAX, 48)
    (a, sa), (b, sb) = imgs[48], imgs[0]
    assert np.isnan(a[19, 88]).all() and sa.nan_pixels == sb.nan_pixels
    assert np.array_equal(a, b, equal_nan=True)
    assert sa.segments == sb.segments and sa.sphere_tests == sb.sphere_tests and sa.quad_tests == sb.quad_tests
    q = R.RtwParams.from_buffer_copy(p)
    q.row_block, q.part_index, q.part_count = 1, 19, 400       # row 19 alone
    ref, _ = O.render(cam, scene, q, 16)
    assert np.array_equal(ref.reshape(-1, 400, 3)[0], b[19], equal_nan=True)
    # ... and the frame as the reference renders it (2500 spp: ~60 such paths): one image from both kernels
    cam, p = R.default_view(R.SCENE_PRESENTATION)
    try:
        full = {}
        for walk_max in (48, 0):
            gpu.set_option(R.OPT_LIST_WALK_MAX, walk_max)
            full[walk_max] = gpu.render(cam, p)
    finally:
        gpu.set_option(R.OPT_LIST_WALK_MAX, 48)
    assert full[48][1].nan_pixels == full[0][1].nan_pixels and full[48][1].segments == full[0][1].segments
    assert np.array_equal(full[48][0], full[0][0], equal_nan=True)


@pytest.mark.parametrize("integrator", ["gradient", "bg_color"])
def test_book1_with_quads_and_a_medium_through_the_tree(gpu, integrator):
    """The scene that runs the GEOM builds of the traversal kernel as shipped: the Book-1 final scene's 485 spheres through the tree, three quads (a light,
    a mirror, a glass pane) and a rotated smoke box walked in the SHADE step.  `gradient` with the render_row sampler is the common configuration, which since
    round 3 has GEOM builds of its own (integrator / sampler folded in at compile time, SPEC == 2); `bg_color` runs the generic GEOM build.  Bit-exact against the
    oracle, tree and list walk alike."""
    base = R.Scene.generate(R.SCENE_C2, 42)
    spheres = [R.RtwSphere.from_buffer_copy(base._spheres[i]) for i in range(base.n_spheres)]
    quads = [R.Quad.new((-2.0, 6.0, -2.0), (4, 0, 0), (0, 0, 4), (0.0, 0.0, 1.0), (1, 1, 1), emitted=(7, 7, 7)),
             R.Quad.new((-8.0, 0.0, -9.0), (16, 0, 0), (0, 5, 0), R.METALLIC_M, (0.8, 0.85, 0.88)),
             R.Quad.new((2.0, 0.0, 2.5), (1.5, 0, -1.0), (0, 1.5, 0), R.GLASS_M, (1, 1, 1))]
    box = R.Instance.new_box((-1.0, 0.0, -1.0), (1.0, 1.6, 1.0), (0.9, 0.9, 0.9), R.SCATTER_M)
    box.rotate((0.0, 0.5, 0.0)); box.translate((-3.0, 0.0, 3.0)); box.const_density(0.8)
    scene = R.Scene(spheres, background=(0.5, 0.7, 1.0), quads=quads, instances=[box])
    cam, p = R.default_view(R.SCENE_C2)
    p.samples, p.gamma = 4, 1.0                                          # the frame's own camera (1200 x 675); 4 spp keeps the oracle at seconds
    p.integrator = R.INTEGRATOR_GRADIENT if integrator == "gradient" else R.INTEGRATOR_BG_COLOR
    q = R.RtwParams.from_buffer_copy(p)
    q.row_block, q.part_index, q.part_count = 8, 3, 12                   # every twelfth block of 8 rows: 56 rows spread over the frame
    ref, st_ref = O.render(cam, scene, q, 16)
    gpu.set_scene(scene)
    for accel in (R.ACCEL_BVH, R.ACCEL_BRUTE):
        q.accel = accel
        img, st = gpu.render(cam, q)
        assert st.segments == st_ref.segments and st.quad_tests == st_ref.quad_tests, accel
        assert np.array_equal(img, ref, equal_nan=True), accel


@pytest.mark.parametrize("case", ["far 1e17", "far 4e17", "far 6e17", "far 1e19", "far 1e30", "centre -1e25", "centre inf", "centre nan", "radius nan", "radius inf",
                                  "radius 1e20", "radius 0", "radius negative", "far 1e17 radius 1e17", "velocity nan"])
def test_spheres_at_extreme_or_non_finite_coordinates(gpu, case):
    """The reference's quadratic degenerates for a sphere whose centre is at 1e19 or beyond, or not a number: |oc|^2 overflows, the discriminant is NaN,
    and a NaN root passes both range tests of sphere.rs:118-121 -- such a sphere is hit by every ray that reaches it in list order with nothing accepted
    before, and real hits never replace it.  Only the list walk gives that order-dependent answer, so a BVH request on such a scene walks the list (the
    shim bounds |d| x the farthest centre by 1e18; up to round 3 the tree pruned the sphere and rendered another image).  Scenes just below the bound go
    through the tree and must match too; huge and infinite radii are fine either way (exact test for every query, no NaN).  With the odd sphere first, in the
    middle and last in list order."""
    inf, nan = float("inf"), float("nan")
    odd = {"far 1e17": ((1e17, 0, -8), 1.0), "far 4e17": ((4e17, 3e17, -8), 1.0), "far 6e17": ((6e17, 0, -8), 1.0), "far 1e19": ((1e19, 0, -8), 1.0),
           "far 1e30": ((1e30, 0, -8), 1.0), "centre -1e25": ((0, 0, 1e25), 3.0), "centre inf": ((inf, 0, -8), 1.0), "centre nan": ((nan, 0, -8), 1.0),
           "radius nan": ((0, 1, -8), nan), "radius inf": ((0, 1, -8), inf), "radius 1e20": ((0, 0, -8), 1e20), "radius 0": ((0, 1, -6), 0.0),
           "radius negative": ((0.5, 0.5, -6), -0.7), "far 1e17 radius 1e17": ((1e17, 0, -8), 1e17), "velocity nan": ((0, 1, -6), 0.5)}[case]
    rng = np.random.default_rng(5)
    mats = [R.SCATTER_M, R.METALLIC_M, R.GLASS_M]
    vp = R.Viewport.new_from_res(96, 54, 4, 8, 1.0, vfov=70.0, lens_radius=0.0)
    cam = vp.camera(); p = vp.params(R.INTEGRATOR_GRADIENT, R.SAMPLER_ROW)
    for pos in (0, 30, 60):
        sp = [R.Sphere.with_albedo(rng.uniform(-4, 4, 3) + [0, 0, -8], float(rng.uniform(0.2, 0.6)), rng.uniform(0.2, 0.95, 3), mats[i % 3]) for i in range(60)]
        vel = (nan, 0.0, 0.0) if case == "velocity nan" else (0.0, 0.0, 0.0)
        extra = R.Sphere.with_albedo(odd[0], odd[1], (0.9, 0.2, 0.2), R.SCATTER_M, velocity=vel)
        sp.insert(pos, extra)
        scene = R.Scene(sp)
        ref, st_ref, out = render_both_r3(gpu, scene, cam, p)
        for accel, (img, st) in out.items():
            assert st.segments == st_ref.segments, (case, pos, accel)
            assert np.array_equal(img, ref, equal_nan=True), (case, pos, accel)


def render_both_r3(gpu, scene, cam, p):
    ref, st_ref = O.render(cam, scene, p, 16)
    gpu.set_scene(scene, cam.time0, cam.time0 + cam.shutter)
    out = {}
    try:
        for name, walk_max, accel in (("list walk", 48, R.ACCEL_BRUTE), ("bvh request", 48, R.ACCEL_BVH), ("tree forced", 0, R.ACCEL_BVH)):
            gpu.set_option(R.OPT_LIST_WALK_MAX, walk_max); p.accel = accel
            out[name] = gpu.render(cam, p)
    finally:
        gpu.set_option(R.OPT_LIST_WALK_MAX, 48)
    return ref, st_ref, out


@pytest.mark.parametrize("case", ["maxt nan", "maxt inf", "mint nan", "mint inf", "mint -1", "ir nan", "ir 0", "ir inf", "ir 1e30", "ir -1.5", "metallicness nan", "metallicness inf",
                                  "metallicness 1e30", "metallicness 2", "metallicness -1", "opacity nan", "albedo nan", "albedo inf", "depth 200"])
def test_odd_ranges_and_material_scalars(gpu, case):
    """More inputs that take the reference's arithmetic out of the ordinary.  A NaN `maxt` makes `x > maxt` reject nothing; a NaN (or absurd) metallicness or
    refraction index makes the NEXT ray NaN or its |d|^2 overflow, and such a ray "hits" the first sphere of the list (NaN roots pass, sphere.rs:118-121).  A BVH
    request on such inputs walks the list (rtw_ctx_set_scene / rtw_ctx_render record and test it); everything else here goes through the tree.  Found with
    scripts/gpu_extreme2.py (profiles/r03_extreme3.log: the tree rendered maxt = NaN, ir = NaN and metallicness = NaN differently up to round 3)."""
    inf, nan = float("inf"), float("nan")
    rng = np.random.default_rng(7)
    mats = [R.SCATTER_M, R.METALLIC_M, R.GLASS_M, R.FUZZY3_M]
    sp = [R.Sphere.with_albedo((0, -100.5, -8), 100.0, (0.5, 0.5, 0.5), R.SCATTER_M)]
    sp += [R.Sphere.with_albedo(rng.uniform(-4, 4, 3) + [0, 0, -8], float(rng.uniform(0.2, 0.6)), rng.uniform(0.2, 0.95, 3), mats[i % 4]) for i in range(70)]
    odd = {"ir nan": ((1.0, 1.0, nan), (1, 1, 1)), "ir 0": ((1.0, 1.0, 0.0), (1, 1, 1)), "ir inf": ((1.0, 1.0, inf), (1, 1, 1)), "ir 1e30": ((1.0, 1.0, 1e30), (1, 1, 1)),
           "ir -1.5": ((1.0, 1.0, -1.5), (1, 1, 1)), "metallicness nan": ((nan, 0.0, 1.0), (0.8, 0.8, 0.8)), "metallicness inf": ((inf, 0.0, 1.0), (0.8, 0.8, 0.8)),
           "metallicness 1e30": ((1e30, 0.0, 1.0), (0.8, 0.8, 0.8)), "metallicness 2": ((2.0, 0.0, 1.0), (0.8, 0.8, 0.8)), "metallicness -1": ((-1.0, 0.0, 1.0), (0.8, 0.8, 0.8)),
           "opacity nan": ((0.5, nan, 1.5), (0.8, 0.8, 0.8)), "albedo nan": (R.SCATTER_M, (nan, 0.5, 0.5)), "albedo inf": (R.SCATTER_M, (inf, 0.5, 0.5))}.get(case)
    if odd is not None:
        for i in range(1, 70, 3): sp[i] = R.Sphere.with_albedo(tuple(sp[i].pod.center), sp[i].pod.radius, odd[1], odd[0])
    vp = R.Viewport.new_from_res(96, 54, 4, 12, 1.0, vfov=70.0, lens_radius=0.02)
    cam = vp.camera(); p = vp.params(R.INTEGRATOR_GRADIENT, R.SAMPLER_ROW)
    if case == "maxt nan": p.maxt = nan
    if case == "maxt inf": p.maxt = inf
    if case == "mint nan": p.mint = nan
    if case == "mint inf": p.mint = inf
    if case == "mint -1": p.mint = -1.0
    if case == "depth 200": p.depth = 200
    ref, st_ref, out = render_both_r3(gpu, R.Scene(sp), cam, p)
    for accel, (img, st) in out.items():
        assert st.segments == st_ref.segments, (case, accel)
        assert np.array_equal(img, ref, equal_nan=True), (case, accel)


def test_two_hundred_thousand_spheres_tree_equals_list(gpu):
    """A scene far beyond the reference's sizes (200 000 spheres: nodes in global memory, a tree ~18 levels deep): the traversal kernel returns the list walk's
    image bit for bit (no oracle here: the CPU would need minutes).  scripts/gpu_many_spheres.py goes to a million."""
    n = 200000
    rng = np.random.default_rng(n)
    pods = (R.RtwSphere * n)()
    base = R.Sphere.with_albedo((0, 0, 0), 1.0, (0.7, 0.6, 0.5), R.SCATTER_M).pod
    np.frombuffer(pods, dtype=np.uint8).reshape(n, -1)[:] = np.frombuffer(bytes(base), dtype=np.uint8)
    f = np.frombuffer(pods, dtype=np.float32).reshape(n, -1)
    f[:, 0:3] = rng.uniform(-30, 30, (n, 3)).astype(np.float32) + np.float32([0, 0, -40])
    f[:, 3] = rng.uniform(0.02, 0.25, n).astype(np.float32)
    scene = R.Scene([R.Sphere.with_albedo((0, 0, 0), 1.0, (0.7, 0.6, 0.5), R.SCATTER_M)])
    scene._spheres = pods; scene.n_spheres = n                    # (the ctypes array stands in for the list of wrappers: 200 000 Python objects are slow)
    scene.pod.spheres = pods; scene.pod.n_spheres = n
    vp = R.Viewport.new_from_res(96, 54, 2, 8, 1.0, vfov=70.0, lens_radius=0.01)
    cam = vp.camera(); p = vp.params(R.INTEGRATOR_GRADIENT, R.SAMPLER_ROW)
    gpu.set_scene(scene)
    p.accel = R.ACCEL_BVH
    a, sa = gpu.render(cam, p)
    p.accel = R.ACCEL_BRUTE
    b, sb = gpu.render(cam, p)
    assert sa.node_tests > 0 and sb.node_tests == 0
    assert sa.segments == sb.segments and np.array_equal(a, b, equal_nan=True)


@pytest.mark.parametrize("integrator", ["gradient", "bg_color"])
def test_odd_quads_instances_and_media_next_to_a_field_of_spheres(gpu, integrator):
    """Degenerate and non-finite quads, instances rotated by NaN or moved to infinity, smokes of density 0 / negative / NaN / inf / 1e-30 / 1e30, each next to 70
    spheres that go through the tree: the GEOM builds (which test every new ray and walk the list for one that is not finite) and the list walk render the oracle's
    image.  scripts/gpu_extreme3.py; profiles/r03_extreme4.log."""
    inf, nan = float("inf"), float("nan")
    rng = np.random.default_rng(9)
    mats = [R.SCATTER_M, R.METALLIC_M, R.GLASS_M]
    def field():
        return [R.Sphere.with_albedo(rng.uniform(-4, 4, 3) + [0, 0, -8], float(rng.uniform(0.2, 0.6)), rng.uniform(0.2, 0.95, 3), mats[i % 3]) for i in range(70)]
    def box(rot=(0.0, 0.5, 0.0), tr=(1.5, 0.0, -6.0), density=None):
        b = R.Instance.new_box((-1.0, -1.0, -1.0), (1.0, 1.0, 1.0), (0.8, 0.8, 0.8), R.SCATTER_M)
        b.rotate(rot); b.translate(tr)
        if density is not None: b.const_density(density)
        return b
    bg = (0.4, 0.5, 0.7)
    scenes = [R.Scene(field(), background=bg, quads=[R.Quad.new((-3, -1, -9), (6, 0, 0), (3, 0, 0), R.SCATTER_M, (0.8, 0.8, 0.8))]),            # u x v = 0
              R.Scene(field(), background=bg, quads=[R.Quad.new((nan, -1, -9), (6, 0, 0), (0, 4, 0), R.SCATTER_M, (0.8, 0.8, 0.8))]),
              R.Scene(field(), background=bg, quads=[R.Quad.new((1e20, -1, -9), (6, 0, 0), (0, 4, 0), R.SCATTER_M, (0.8, 0.8, 0.8))]),
              R.Scene(field(), background=bg, quads=[R.Quad.new((-5e19, -5e19, -12), (1e20, 0, 0), (0, 1e20, 0), R.SCATTER_M, (0.8, 0.8, 0.8))]),
              R.Scene(field(), background=bg, instances=[box(rot=(0.0, nan, 0.0))]),
              R.Scene(field(), background=bg, instances=[box(tr=(inf, 0.0, -6.0))]),
              R.Scene(field(), background=bg, instances=[box(tr=(1e25, 0.0, -6.0))])]
    scenes += [R.Scene(field(), background=bg, instances=[box(density=d)]) for d in (0.0, -1.0, nan, inf, 1e-30, 1e30)]
    vp = R.Viewport.new_from_res(96, 54, 4, 10, 1.0, vfov=70.0, lens_radius=0.0)
    cam = vp.camera()
    p = vp.params(R.INTEGRATOR_GRADIENT if integrator == "gradient" else R.INTEGRATOR_BG_COLOR, R.SAMPLER_ROW)
    for k, scene in enumerate(scenes):
        ref, st_ref, out = render_both_r3(gpu, scene, cam, p)
        for accel, (img, st) in out.items():
            assert st.segments == st_ref.segments, (k, accel)
            assert np.array_equal(img, ref, equal_nan=True), (k, accel)


@pytest.mark.parametrize("case", ["shutter nan", "time0 nan", "time0 inf", "shutter negative", "shutter 1e30", "time0 -1e10"])
def test_odd_times_with_moving_spheres(gpu, case):
    """Ray times out of the ordinary with a third of the spheres moving: a NaN shutter makes every moving centre NaN (centre = origin + velocity * time,
    sphere.rs:100) and with it the reference's NaN-root hits; up to round 3 a NaN slipped through the shim's fmin / fmax range check and the tree
    rendered another image.  scripts/gpu_extreme4.py, profiles/r03_extreme5.log."""
    inf, nan = float("inf"), float("nan")
    t0, sh = {"shutter nan": (0.0, nan), "time0 nan": (nan, 0.02), "time0 inf": (inf, 0.0), "shutter negative": (0.5, -0.03), "shutter 1e30": (0.0, 1e30), "time0 -1e10": (-1e10, 0.0)}[case]
    rng = np.random.default_rng(11)
    mats = [R.SCATTER_M, R.METALLIC_M, R.GLASS_M]
    sp = [R.Sphere.with_albedo(rng.uniform(-4, 4, 3) + [0, 0, -8], float(rng.uniform(0.2, 0.6)), rng.uniform(0.2, 0.95, 3), mats[i % 3],
                               velocity=tuple(rng.uniform(-3, 3, 3)) if i % 3 == 0 else (0, 0, 0)) for i in range(70)]
    vp = R.Viewport.new_from_res(96, 54, 4, 10, 1.0, vfov=70.0, lens_radius=0.01)
    cam = vp.camera(); cam.time0, cam.shutter = t0, sh
    p = vp.params(R.INTEGRATOR_GRADIENT, R.SAMPLER_ROW)
    ref, st_ref, out = render_both_r3(gpu, R.Scene(sp), cam, p)
    for accel, (img, st) in out.items():
        assert st.segments == st_ref.segments, (case, accel)
        assert np.array_equal(img, ref, equal_nan=True), (case, accel)
