"""Film partition across ranks: the reference parallelises over independent 16x16 tiles
(integrator/mod.rs:55-74); here rank r owns the interleaved 16-row bands b with b % world == r, which keeps
each band contiguous in the film and balances path length across the image. With the box filter (radius 0.5)
no sample splats outside its own pixel, so bands are disjoint and a sum-reduce reassembles the frame."""

BAND = 16


def band_rects(W, H, rank, world, band=BAND):
    rects = []
    for b, y0 in enumerate(range(0, H, band)):
        if b % world == rank:
            rects.append((0, y0, W, min(H, y0 + band)))
    return rects


def reduce_film(film, world, dst=0):
    """One collective per frame: RCCL (backend "nccl") on GPUs, gloo in the CPU tests."""
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
