"""Multi-rank path on the real device code, the C-ABI collective, and the native command line (`-m gpu`).

A one-GPU box cannot run RCCL between two ranks (one rank per device), so the two-rank test exchanges the films over gloo while both
ranks render their bands with the HIP kernels on GPU 0 - each in its own fresh process with its own handle - and the RCCL entry
points are driven with a one-rank communicator (ncclCommInitRank / ncclReduce / group of zero sends really execute)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _worker(rank, world, port, workdir, out, filt):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from rs_ray_toy_amd import RRT_F32, Renderer, Scene, scenes
    from rs_ray_toy_amd.partition import gather_film, reduce_film
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg, root = scenes.cfg2(os.path.join(workdir, f"r{rank}"), xres=96, yres=80, nsamp=9, max_depth=3)
    if filt:
        cfg["Film"]["Filter"] = filt
    sc = Scene.loads(cfg, root)
    W, H = sc.resolution
    r = Renderer(sc, 0, RRT_F32)            # created after the spawn: this process's own HIP context, streams and pools
    film = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
    r.render_bands_device(rank, world, film.data_ptr(), stats=False)
    torch.cuda.synchronize()
    mine = film.clone()
    reduce_film(film, world)                # sum on rank 0 (any filter)
    res = [film.cpu().numpy()]
    if not filt:                            # box filter: the gather form (what rrt_film_gather sends over RCCL)
        g = gather_film(mine, world)
        res.append(g.cpu().numpy())
    if rank == 0:
        full = r.render()                   # the single-process frame on the same device code
        np.save(out, np.stack(res + [full]))
    r.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("filt", [None, {"filter_type": "TriangleFilter", "radius": [2.0, 2.0]}], ids=["box", "triangle"])
def test_two_ranks_render_their_bands_on_the_device_and_reassemble(tmp_path, filt):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "films.npy")
    mp.spawn(_worker, args=(2, port, str(tmp_path), out, filt), nprocs=2, join=True)
    films = np.load(out)
    full = films[-1]
    assert full[..., :3].max() > 0
    if filt is None:
        assert np.array_equal(films[0], full) and np.array_equal(films[1], full)   # disjoint bands: bit for bit
    else:
        # splats cross band borders: a border pixel's sum is split over two ranks, i.e. added in a different order
        assert np.array_equal(films[0][..., 3], full[..., 3])
        np.testing.assert_allclose(films[0], full, rtol=2e-6, atol=1e-7 * full.max())


def test_rccl_collective_through_the_c_abi(workdir):
    """rrt_comm_* / rrt_film_gather / rrt_film_gather_all with the one rank a one-GPU box allows: the communicator is a real RCCL one
    and the wide-filter form runs a real ncclReduce on the handle's stream; the film must come back unchanged."""
    from rs_ray_toy_amd import RRT_F32, Renderer, Scene, scenes
    from rs_ray_toy_amd.api import Comm
    import ctypes as C
    from rs_ray_toy_amd import _abi as A
    for filt in (None, {"filter_type": "GaussianFilter", "radius": [1.5, 1.5], "alpha": 2.0}):
        cfg, root = scenes.cfg2(workdir, xres=64, yres=48, nsamp=5, max_depth=2)
        if filt:
            cfg["Film"]["Filter"] = filt
        sc = Scene.loads(cfg, root)
        W, H = sc.resolution
        r = Renderer(sc, 0, RRT_F32)
        film = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
        torch.cuda.synchronize()
        comm = Comm(Comm.new_id(), 0, 1, 0)
        r.set_option("frame_stats", 1)
        r.render_bands_begin(0, 1, film.data_ptr())
        comm.gather(r, film.data_ptr(), 0)
        st = r.render_end(stats=True)
        # rrt_render_stats::ms_gather: HIP events on the handle's stream around the collective, apart from the render (ms_total)
        assert st.ms_total > 0 and st.ms_gather >= 0
        if filt:
            assert st.ms_gather > 0          # a real ncclReduce ran between the two events
        r.render_bands_begin(0, 1, film.data_ptr())
        assert r.render_end(stats=True).ms_gather == 0    # a frame no collective followed reports none
        film.zero_(); torch.cuda.synchronize()
        r.render_bands_begin(0, 1, film.data_ptr())
        comm.gather(r, film.data_ptr(), 0)
        r.render_end()
        ref = r.render()
        assert np.array_equal(film.cpu().numpy(), ref)
        handles = (C.c_void_p * 1)(r._h)
        films = (C.c_void_p * 1)(film.data_ptr())
        assert A.lib().rrt_film_gather_all(handles, films, 1, 0) == A.RRT_OK, A.lib().rrt_last_error()
        torch.cuda.synchronize()
        assert np.array_equal(film.cpu().numpy(), ref)
        comm.close()
        r.close()


def test_a_failure_inside_the_rccl_group_closes_it_before_the_communicator_is_aborted(workdir):
    """an enqueue error inside an open ncclGroup. RRT_TEST_FAIL_GATHER makes this rank's transfer name a peer that does not exist, so
    RCCL refuses the call inside the group; rrt_film_gather must end the group first (RCCL discards a group that holds a failed call), then abort
    the communicator, report RRT_EDEVICE, refuse the rrt_comm from then on - and the process, the handle and a fresh communicator must go on working."""
    from rs_ray_toy_amd import RRT_F32, Renderer, RrtError, Scene, scenes
    from rs_ray_toy_amd.api import Comm
    cfg, root = scenes.cfg2(workdir, xres=64, yres=48, nsamp=5, max_depth=2)
    cfg["Film"]["Filter"] = {"filter_type": "GaussianFilter", "radius": [1.5, 1.5], "alpha": 2.0}   # the reduce form: one rank has something to issue
    sc = Scene.loads(cfg, root)
    W, H = sc.resolution
    r = Renderer(sc, 0, RRT_F32)
    film = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    ref = r.render()
    comm = Comm(Comm.new_id(), 0, 1, 0)
    os.environ["RRT_TEST_FAIL_GATHER"] = "0"
    try:
        r.render_bands_begin(0, 1, film.data_ptr())
        with pytest.raises(RrtError) as e:
            comm.gather(r, film.data_ptr(), 0)
        assert "RCCL error" in str(e.value)
        r.render_end()
        assert np.array_equal(film.cpu().numpy(), ref)      # the frame itself is intact: nothing of the failed group was launched
    finally:
        del os.environ["RRT_TEST_FAIL_GATHER"]
    with pytest.raises(RrtError) as e:                      # the aborted communicator is refused, not reused
        comm.gather(r, film.data_ptr(), 0)
    assert "aborted" in str(e.value)
    comm.close()
    comm2 = Comm(Comm.new_id(), 0, 1, 0)                    # and RCCL is still usable in this process
    film.zero_(); torch.cuda.synchronize()
    r.render_bands_begin(0, 1, film.data_ptr())
    comm2.gather(r, film.data_ptr(), 0)
    r.render_end()
    assert np.array_equal(film.cpu().numpy(), ref)
    comm2.close()
    r.close()


@pytest.mark.parametrize("filt", [None, {"filter_type": "TriangleFilter", "radius": [2.0, 2.0]}], ids=["box", "triangle"])
def test_film_gather_all_over_every_device_of_the_box(workdir, filt):
    """rrt_film_gather_all over rrt_device_count() devices - 1 on this pool's boxes (nothing to send, RCCL not needed), N on a node, where
    this is the test that runs the grouped ncclSend / ncclRecv gather (box) and the ncclReduce (wide filter) between real ranks: one handle
    per device renders its bands, the collective reassembles the frame on device 0, and it must equal the single-handle frame. The cached
    communicators go away with the handles (rrt_destroy)."""
    import ctypes as C
    from rs_ray_toy_amd import RRT_F32, Renderer, Scene, scenes
    from rs_ray_toy_amd import _abi as A
    n = A.lib().rrt_device_count()
    assert n >= 1
    cfg, root = scenes.cfg2(workdir, xres=96, yres=80, nsamp=9, max_depth=3)
    if filt:
        cfg["Film"]["Filter"] = filt
    sc = Scene.loads(cfg, root)
    W, H = sc.resolution
    rs = [Renderer(sc, i, RRT_F32) for i in range(n)]
    films = [torch.zeros((H, W, 4), dtype=torch.float32, device=f"cuda:{i}") for i in range(n)]
    torch.cuda.synchronize()
    for i in range(n):
        rs[i].render_bands_begin(i, n, films[i].data_ptr())
    handles = (C.c_void_p * n)(*[r._h for r in rs])
    ptrs = (C.c_void_p * n)(*[f.data_ptr() for f in films])
    assert A.lib().rrt_film_gather_all(handles, ptrs, n, 0) == A.RRT_OK, A.lib().rrt_last_error()
    for i in reversed(range(n)):
        rs[i].render_end()
    got = films[0].cpu().numpy()
    full = rs[0].render()
    if filt is None or n == 1:
        assert np.array_equal(got, full)
    else:
        assert np.array_equal(got[..., 3], full[..., 3])
        np.testing.assert_allclose(got, full, rtol=2e-6, atol=1e-7 * full.max())
    for r in rs:
        r.close()


def test_native_command_line_matches_the_python_host(tmp_path):
    """rrt_render <scene.json> <out.png> (main.rs:55-61 / deploy_render renderprocess.rs:92-105 in C++ over the C ABI) writes the very
    PNG the Python mirror writes, for the reference's own samples/scene.json (StratifiedSampler, Debug integrator, 640x360)."""
    from rs_ray_toy_amd import deploy_render
    exe = os.path.join(ROOT, "rs_ray_toy_amd", "csrc", "rrt_render")
    scene = os.path.join(ROOT, "tests", "golden", "scene.json")
    a, b = str(tmp_path / "cli.png"), str(tmp_path / "py.png")
    p = subprocess.run([exe, scene, a], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr
    _, st = deploy_render(scene, b)
    assert p.stdout.strip() == f"{st.camera_rays} rays generated"      # integrator/mod.rs:137
    assert open(a, "rb").read() == open(b, "rb").read()
    assert subprocess.run([exe, scene], capture_output=True).returncode == 2          # usage
    q = subprocess.run([exe, str(tmp_path / "missing.json"), a], capture_output=True, text=True)
    assert q.returncode == 1 and "cannot open" in q.stderr
