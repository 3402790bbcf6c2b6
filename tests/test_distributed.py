"""N > 1 path on CPU: two gloo ranks, each owning its interleaved 16-row bands (rs_ray_toy_amd/partition.py),
one sum-reduce to rank 0 — the same partition + collective bench.py runs over RCCL. The per-rank executor here
is the oracle (no GPU in this container); the partition/collective logic is what is under test."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, workdir, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import oracle_lib as O
    from rs_ray_toy_amd import Scene, scenes
    from rs_ray_toy_amd.partition import band_rects, gather_film, reduce_film
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg, root = scenes.cfg2(os.path.join(workdir, f"r{rank}"), xres=64, yres=72, nsamp=4)
    sc = Scene.loads(cfg, root)
    W, H = sc.resolution
    film = np.zeros((H, W, 4))
    rects = band_rects(W, H, rank, world)
    for rect in rects:
        film += O.render(sc, rect, n_threads=2)
    t = torch.from_numpy(film.copy())
    reduce_film(t, world)
    # the gather form of the same exchange (what rrt_film_gather does over RCCL for the box filter): only band rows travel
    g = torch.from_numpy(film.copy())
    gather_film(g, world)
    if rank == 0:
        full = O.render(sc, n_threads=2)
        np.save(out, np.stack([t.numpy(), full, g.numpy()]))
    dist.barrier()
    dist.destroy_process_group()


def test_band_partition_covers_the_film_once():
    from rs_ray_toy_amd.partition import band_rects
    for W, H, world in ((64, 72, 2), (1024, 1024, 8), (33, 50, 3)):
        cover = np.zeros((H, W), int)
        sizes = []
        for r in range(world):
            rects = band_rects(W, H, r, world)
            sizes.append(sum((y1 - y0) * (x1 - x0) for x0, y0, x1, y1 in rects))
            for x0, y0, x1, y1 in rects:
                cover[y0:y1, x0:x1] += 1
        assert (cover == 1).all()
        assert max(sizes) - min(sizes) <= 16 * W


def test_two_rank_gloo_reduce_reassembles_the_frame(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "films.npy")
    mp.spawn(_worker, args=(2, port, str(tmp_path), out), nprocs=2, join=True)
    got, full, gathered = np.load(out)
    assert full[..., :3].max() > 0
    assert np.array_equal(got, full)      # bands are disjoint: the reduce is exact
    assert np.array_equal(gathered, full)  # and so is the gather of the band rows alone
