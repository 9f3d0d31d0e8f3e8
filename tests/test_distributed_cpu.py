"""CPU, world_size 2 over gloo: the row-cyclic partition + single gather of drt_dist reassembles the frame.
The per-rank renderer here is the CPU oracle standing in for the HIP renderer (the tiling code takes a callable)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cases


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, height, width, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    for p in (os.path.join(cases.REPO, "daily-ray-trace_amd"), os.path.join(cases.REPO, "oracle"), os.path.join(cases.REPO, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import drt_dist
    import oracle_py as O
    import pydrt
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bundle = pydrt.load_scene(cases.scene_path("cornell_plane_light.scn"), width, height)
    S = bundle.S

    def render_tile(y0, tile_h, stride):
        p = pydrt.make_params(width, height, spp=2, max_depth=4, seed=5, y0=y0, tile_h=tile_h, row_stride=stride)
        px, av, va, _, _ = O.oracle_render_tile(bundle, p, math_mode=O.MATH_DEVICE)
        return [torch.from_numpy(px), torch.from_numpy(av), torch.from_numpy(va)]

    full = drt_dist.render_distributed(render_tile, height, width, rank, world, channels=(S + 1, S, S))
    # the single-collective form bench.py uses: one contiguous film buffer per rank, one gather
    fg = drt_dist.FilmGather(height, width, S, rank, world, torch.device("cpu"))
    y0, tile_h, stride = drt_dist.rank_rows(height, rank, world)
    for i, t in enumerate(render_tile(y0, tile_h, stride)):
        fg.region(i).copy_(t)
    flat_full = fg.gather()
    if rank == 0:
        for a, b in zip(full, flat_full):
            assert torch.equal(a, b)
    else:
        assert flat_full is None
    # the pipelined form: the rank's rows in 3 blocks, each gathered asynchronously while the next one "renders"
    blocks = drt_dist.film_blocks(height, width, S, rank, world, torch.device("cpu"), 3)
    for fb in blocks:
        by0, brows, bstride = fb.tile()
        if brows:
            for i, t in enumerate(render_tile(by0, brows, bstride)):
                fb.region(i).copy_(t)
        fb.gather_async()
    piped = None
    for fb in blocks:
        piped = fb.finish()
    if rank == 0:
        for a, b in zip(full, piped):
            assert torch.equal(a, b)
    # one buffer of 8 doubles per pixel (the XYZ film's layout) through the same blocks
    blocks8 = drt_dist.film_blocks(height, width, S, rank, world, torch.device("cpu"), 2, channels=(8,))
    for fb in blocks8:
        by0, brows, bstride = fb.tile()
        if brows:
            fb.region(0).copy_(render_tile(by0, brows, bstride)[0][:, :8])
        fb.gather_async()
    img8 = None
    for fb in blocks8:
        img8 = fb.finish()
    if rank == 0:
        assert len(img8) == 1 and torch.equal(img8[0], full[0][:, :, :8])
    dist.barrier()
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)  # the max-over-ranks timing reduction bench.py uses
    assert t.item() == world
    if rank == 0:
        np.savez(out_path, px=full[0].numpy(), av=full[1].numpy(), va=full[2].numpy())
    else:
        assert full is None
    dist.destroy_process_group()


@pytest.mark.parametrize("world,height", [(2, 20), (2, 21), (3, 20)])
def test_row_cyclic_tiles_gather_to_the_full_frame(tmp_path, world, height):
    import oracle_py as O
    import pydrt
    width = 24
    out = str(tmp_path / "full.npz")
    mp.spawn(_worker, args=(world, _free_port(), height, width, out), nprocs=world, join=True)
    got = np.load(out)
    bundle = pydrt.load_scene(cases.scene_path("cornell_plane_light.scn"), width, height)
    p = pydrt.make_params(width, height, spp=2, max_depth=4, seed=5)
    px, av, va, _, _ = O.oracle_render_tile(bundle, p, math_mode=O.MATH_DEVICE)
    S = bundle.S
    assert np.array_equal(got["px"].reshape(-1, S + 1), px)
    assert np.array_equal(got["av"].reshape(-1, S), av)
    assert np.array_equal(got["va"].reshape(-1, S), va)


def _solo_worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    sys.path.insert(0, os.path.join(cases.REPO, "daily-ray-trace_amd"))
    import drt_dist
    dist.init_process_group("gloo", rank=0, world_size=1)
    S, H, W = 5, 7, 4
    fg = drt_dist.FilmGather(H, W, S, 0, 1, torch.device("cpu"), always_collective=True)
    assert fg.collective and fg.recv is not None
    want = []
    for i in range(3):
        t = torch.arange(fg.region(i).numel(), dtype=torch.float64).reshape(fg.region(i).shape) + 100.0 * i
        fg.region(i).copy_(t)
        want.append(t.reshape(H, W, -1))
    fg.gather_async()
    img = fg.finish()
    assert all(torch.equal(a, b) for a, b in zip(img, want))
    plain = drt_dist.FilmGather(H, W, S, 0, 1, torch.device("cpu"))
    assert not plain.collective and plain.recv is None
    dist.destroy_process_group()


def test_world_of_one_can_still_go_through_the_collective():
    """FilmGather(always_collective=True): a single rank runs the very gather the N > 1 path runs (what the 1-rank RCCL test on
    the GPU box uses); by default a world of one just copies its rows into the frame."""
    mp.spawn(_solo_worker, args=(1, _free_port()), nprocs=1, join=True)


def test_rank_rows_cover_every_row_once():
    import drt_dist
    for height in (1, 7, 8, 1024, 2047):
        for world in (1, 2, 3, 4, 8):
            seen = np.zeros(height, dtype=int)
            for r in range(world):
                y0, th, st = drt_dist.rank_rows(height, r, world)
                ys = y0 + st * np.arange(th)
                assert (ys < height).all()
                seen[ys] += 1
            assert (seen == 1).all()


def test_bench_self_launch_reports_a_failing_rank():
    """`python bench.py --gpus 2` from a plain shell starts its two ranks itself. Without a GPU every rank refuses to run
    (no CPU fallback); the launcher must pass that on as a non-zero exit code and print no JSON line."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is present: the ranks would run")
    except ImportError:
        pytest.skip("torch not importable")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-device",
                          "--size", "16", "--spp", "1", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode != 0
    assert "needs a GPU" in out.stderr and "2-rank launch failed" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
