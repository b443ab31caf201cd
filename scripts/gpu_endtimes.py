"""Wave lifetimes of the bench frame whole, split 8 and 64 ways (-DRTW_ENDTIMES build: RTW_HIP_LIB=.../lib_endtimes.so RTW_ENDTIMES_DUMP=1).
Run twice: the second time with RTW_ENDTIMES_REF=<longest lifetime of the case> for the histogram of the waves' end times."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import rtw_amd as R
scene = R.Scene.generate(R.SCENE_C2, 42)
cam, p = R.default_view(R.SCENE_C5); cam.shutter = 0.0
r = R.Renderer(0); r.set_scene(scene)
if os.environ.get("TAIL"): r.set_option(R.OPT_TAIL_UNITS, float(os.environ["TAIL"]))
if os.environ.get("ORDER"): r.set_option(R.OPT_TILE_ORDER, float(os.environ["ORDER"]))
if os.environ.get("WGS"): r.set_option(R.OPT_BLOCKS_PER_CU, float(os.environ["WGS"]))
if os.environ.get("CHUNK"): r.set_option(R.OPT_CHUNK_LEN, float(os.environ["CHUNK"]))
out = torch.zeros((1080, 1920, 3), dtype=torch.float32, device="cuda:0")
which = [int(x) for x in os.environ.get("PARTS", "1,8,64").split(",")]
for n in which:
    p.row_block, p.part_index, p.part_count = 8, n // 2, n
    r.render(cam, p, out=out.data_ptr())
    sys.stderr.write(f"--- 1/{n} of the rows\n"); sys.stderr.flush()
    _, st = r.render(cam, p, out=out.data_ptr())
    sys.stderr.write(f"    kernel {st.kernel_ms:.3f} ms\n")
