"""The N>1 path on CPU: world_size-2 (and 3) `gloo` process groups run the same partition + single
gather + de-interleave code as the GPU job (raytracing-in-a-weekend_amd/parallel.py).  The oracle stands
in for the renderer here only because this container has no GPU -- it is the checker in this test too:
the gathered frame must equal the unsplit oracle frame bit for bit."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, height, width, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import rtw_amd as R
    from tests import oracle_binding as O
    from tests.test_oracle_golden import small_view
    par = importlib.import_module("raytracing-in-a-weekend_amd.parallel")
    scene, cam, p = small_view(R.SCENE_C2, width, height, 2)

    def render_rows(row_block, idx, cnt, out):
        pp = R.RtwParams.from_buffer_copy(p)
        pp.row_block, pp.part_index, pp.part_count = row_block, idx, cnt
        img, st = O.render(cam, scene, pp, threads=2)
        out[: img.shape[0]] = torch.from_numpy(img)
        return st

    frame, st = par.render_frame(render_rows, height, width, rank, world, "cpu")
    seg, = par.reduce_counters([st.segments], world, "cpu")
    tmax = par.max_over_ranks(float(rank), world, "cpu")
    if rank == 0:
        full, st_full = O.render(cam, scene, p, threads=2)
        q.put((bool(np.array_equal(frame.numpy(), full)), seg == st_full.segments, tmax == world - 1))
    else:
        assert frame is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,height", [(2, 36), (3, 45)])
def test_gather_reassembles_the_frame(world, height):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, height, 64, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res == (True, True, True)


def test_row_ownership_rule_matches_the_c_abi():
    sys.path.insert(0, ROOT)
    import rtw_amd as R
    par = importlib.import_module("raytracing-in-a-weekend_amd.parallel")
    for h, world in ((1080, 8), (1080, 4), (675, 2), (45, 3), (7, 8)):
        seen = []
        for r in range(world):
            rows = par.rows_of(h, r, world)
            assert len(rows) == R.lib().rtw_part_rows(h, 8, r, world)
            seen += rows
        assert sorted(seen) == list(range(h))
        assert par.max_rows(h, world) == max(R.lib().rtw_part_rows(h, 8, r, world) for r in range(world))
