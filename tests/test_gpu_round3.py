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
