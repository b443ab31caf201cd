"""Time every BASELINE config at full size with both closest-hit strategies (one GPU); prints a markdown table.
Columns: BVH request as shipped (small scenes are routed to the list walk, RTW_OPT_LIST_WALK_MAX), BVH forced (option 0), list walk."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rtw_amd as R
r = R.Renderer(0)
rows = []
cfgs = [("C1 3 spheres", R.SCENE_C1, R.SCENE_C1, None), ("C2 Book-1 final", R.SCENE_C2, R.SCENE_C2, None),
        ("C3 Book-1 final (bench)", R.SCENE_C2, R.SCENE_C5, 0.0), ("C4 dielectric-heavy", R.SCENE_C4, R.SCENE_C4, None),
        ("C5 motion blur + texture", R.SCENE_C5, R.SCENE_C5, None)]
cfgs += [("quad_test (quad.rs:152)", R.SCENE_QUAD_TEST, R.SCENE_QUAD_TEST, None), ("presentation_image (main.rs:89)", R.SCENE_PRESENTATION, R.SCENE_PRESENTATION, None),
         ("First frame (main.rs:427)", R.SCENE_FIRST_FRAME, R.SCENE_FIRST_FRAME, None)]
def book1_with_geometry():
    """The Book-1 final scene + a light quad, a mirror quad, a glass pane and a rotated smoke box: what runs the GEOM builds of the traversal kernel as shipped."""
    base = R.Scene.generate(R.SCENE_C2, 42)
    spheres = [R.RtwSphere.from_buffer_copy(base._spheres[i]) for i in range(base.n_spheres)]
    quads = [R.Quad.new((-2.0, 6.0, -2.0), (4, 0, 0), (0, 0, 4), (0.0, 0.0, 1.0), (1, 1, 1), emitted=(7, 7, 7)),
             R.Quad.new((-8.0, 0.0, -9.0), (16, 0, 0), (0, 5, 0), R.METALLIC_M, (0.8, 0.85, 0.88)),
             R.Quad.new((2.0, 0.0, 2.5), (1.5, 0, -1.0), (0, 1.5, 0), R.GLASS_M, (1, 1, 1))]
    box = R.Instance.new_box((-1.0, 0.0, -1.0), (1.0, 1.6, 1.0), (0.9, 0.9, 0.9), R.SCATTER_M)
    box.rotate((0.0, 0.5, 0.0)); box.translate((-3.0, 0.0, 3.0)); box.const_density(0.8)
    return R.Scene(spheres, background=(0.5, 0.7, 1.0), quads=quads, instances=[box])
cfgs += [("Book-1 + 3 quads + smoke box (scripts/gpu_geom_mixed.py)", "book1+geom", R.SCENE_C2, None)]
for name, scene_id, view_id, shutter in cfgs:
    sc = book1_with_geometry() if scene_id == "book1+geom" else R.Scene.generate_geom(scene_id) if scene_id in (R.SCENE_QUAD_TEST, R.SCENE_PRESENTATION) else R.Scene.generate(scene_id)
    cam, p = R.default_view(view_id)
    if shutter is not None: cam.shutter = shutter
    out = torch.zeros((p.height, p.width, 3), dtype=torch.float32, device="cuda:0")
    r.set_scene(sc, cam.time0, cam.time0 + cam.shutter)
    res = {}
    for accel in (R.ACCEL_BVH, "tree", R.ACCEL_BRUTE):
        r.set_option(R.OPT_LIST_WALK_MAX, 0 if accel == "tree" else 48)
        p.accel = R.ACCEL_BVH if accel == "tree" else accel
        r.render(cam, p, out=out.data_ptr())
        best = None
        for _ in range(3):
            _, st = r.render(cam, p, out=out.data_ptr())
            if best is None or st.kernel_ms < best.kernel_ms: best = st
        res[accel] = best
    b, f, t = res[R.ACCEL_BVH], res[R.ACCEL_BRUTE], res["tree"]
    print(f"| {name} | {sc.n_spheres}+{sc.n_quads}q+{sc.n_instances}i | {p.width}x{p.height}x{b.camera_rays // (p.width * p.height)} | {p.depth} | {b.segments / b.camera_rays:.2f} | "
          f"{b.kernel_ms:.2f} | {b.segments / b.kernel_ms / 1e6:.2f} | {b.camera_rays / b.kernel_ms / 1e6:.2f} | {t.kernel_ms:.2f} | {t.segments / t.kernel_ms / 1e6:.2f} | {f.kernel_ms:.2f} | {f.segments / f.kernel_ms / 1e6:.2f} |", flush=True)
