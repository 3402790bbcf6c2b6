"""Film partition across ranks: the reference parallelises over independent 16x16 tiles
(integrator/mod.rs:55-74); here rank r owns the interleaved 16-row bands b with b % world == r, which keeps
each band contiguous in the film and balances path length across the image. With the box filter (radius 0.5)
no sample splats outside its own pixel, so bands are disjoint and only they need to travel.

The collective itself lives behind the C ABI (rrt_film_gather: RCCL send / recv of the band rows, include/rrt.h).
This module is the torch.distributed form of the same exchange, used where RCCL cannot run: the CPU tests (gloo) and
rehearsals with more ranks than GPUs. Both take their band arithmetic from rrt_band_rows."""
from .api import band_rows

BAND = 16


def band_rects(W, H, rank, world, band=BAND):
    assert band == BAND
    return [(0, y0, W, y1) for y0, y1 in band_rows(H, rank, world)]


def reduce_film(film, world, dst=0):
    """Sum of the ranks' films on `dst` (any filter): RCCL (backend "nccl") on GPUs, gloo in the CPU tests."""
    if world > 1:
        import torch.distributed as dist
        if film.is_cuda and dist.get_backend() == "gloo":
            # rehearsal of the multi-rank path on a box with fewer GPUs than ranks (bench.py --dist-backend gloo): gloo reduces host tensors
            tmp = film.cpu()
            dist.reduce(tmp, dst=dst, op=dist.ReduceOp.SUM)
            if dist.get_rank() == dst:
                film.copy_(tmp)
        else:
            dist.reduce(film, dst=dst, op=dist.ReduceOp.SUM)
    return film


def gather_film(film, world, dst=0):
    """What rrt_film_gather does for the box filter, over torch.distributed point-to-point: every rank sends the rows of its own
    bands to `dst`, which receives them in place (1 / world of the film per rank instead of the whole film). film: (H, W, 4)."""
    if world == 1:
        return film
    import torch.distributed as dist
    rank = dist.get_rank()
    H = film.shape[0]
    host = film.is_cuda and dist.get_backend() == "gloo"
    buf = film.cpu() if host else film
    ops = []
    for r in range(world):
        if r == dst or (rank != dst and rank != r):
            continue
        for y0, y1 in band_rows(H, r, world):
            ops.append(dist.P2POp(dist.irecv if rank == dst else dist.isend, buf[y0:y1], r if rank == dst else dst))
    for req in (dist.batch_isend_irecv(ops) if ops else []):
        req.wait()
    if host and rank == dst:
        film.copy_(buf)
    return film
