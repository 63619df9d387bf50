"""N > 1 path on CPU: two processes over gloo run the same tile split + gather + de-interleave that
bench.py runs over RCCL, with the per-rank tiles rendered by the CPU oracle (no GPU here)."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = Path(__file__).resolve().parent.parent


def _worker(rank, world, port, w, h, strip, out_path):
    sys.path.insert(0, str(REPO))
    sys.path.insert(0, str(REPO / "sycl-ray-tracer_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from rtamd import scenes
    from rtamd.dist import FrameGatherer
    sd = scenes.cube_scene()
    sc = O.OracleScene(sd)
    cam = O.camera(w, h, sd.camera.position, sd.camera.direction, sd.camera.focal_length)
    f, _, rays = sc.render(cam, 1, 6, 2, nthreads=2, rank=rank, world=world, strip_rows=strip)
    fg = FrameGatherer(rank, world, w, h, strip, torch.device("cpu"))
    assert fg.rows == f.shape[0]
    fg.local[: fg.rows] = torch.from_numpy(f)
    frame = fg.gather()
    tot = torch.tensor([float(rays)], dtype=torch.float64)
    dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    if rank == 0:
        np.savez(out_path, frame=frame.numpy(), rays=tot.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,h,strip", [(2, 45, 8), (3, 64, 8)])
def test_two_process_tile_gather_equals_single_frame(tmp_path, oracle, world, h, strip):
    from rtamd import scenes
    w = 48
    out = tmp_path / "frame.npz"
    port = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, w, h, strip, str(out)), nprocs=world, join=True)
    got = np.load(out)
    sd = scenes.cube_scene()
    sc = oracle.OracleScene(sd)
    cam = oracle.camera(w, h, sd.camera.position, sd.camera.direction, sd.camera.focal_length)
    full, _, rays = sc.render(cam, 1, 6, 2)
    np.testing.assert_array_equal(got["frame"], full)
    assert int(got["rays"][0]) == rays


def test_strip_ownership_partitions_the_frame():
    from rtamd.dist import max_local_rows, strip_rows_of
    for h in (1, 7, 8, 9, 1080, 2160, 93):
        for world in (1, 2, 3, 4, 8):
            rows = [strip_rows_of(r, world, h, 8) for r in range(world)]
            assert sorted(sum(rows, [])) == list(range(h))
            assert max(len(r) for r in rows) == max_local_rows(world, h, 8)
    # 1080p over 8 GPUs: 135 strips -> 17 or 16 strips per rank
    assert sorted({len(strip_rows_of(r, 8, 1080)) for r in range(8)}) == [128, 136]
