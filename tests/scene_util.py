"""Scene generators shared by the CPU and the GPU tests."""
import os

import numpy as np

from rs_ray_toy_amd import scenes


def rough_terrain(workdir, seed, n=48, amp=2.5):
    """A steep, noisy heightfield (slopes beyond 60 degrees, narrow valleys): the horizon tables' hard case."""
    rng = np.random.default_rng(seed)
    cfg, root = scenes.cfg4(workdir, xres=96, yres=96, nsamp=9, max_depth=6, n=n)
    g = n + 1
    xs = np.linspace(-10.0, 10.0, g)
    X, Z = np.meshgrid(xs, xs, indexing="ij")
    Y = amp * np.sin(X * 1.3 + rng.random()) * np.cos(Z * 1.1) + amp * 0.6 * (rng.random((g, g)) - 0.5) - 2.0
    P = np.stack([X + 35.0, Y, Z], -1).reshape(-1, 3)
    idx = np.arange(g * g).reshape(g, g)
    a, b, cc, d = idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel(), idx[:-1, 1:].ravel()
    F = np.concatenate([np.stack([a, cc, b], 1), np.stack([a, d, cc], 1)]) + 1
    with open(os.path.join(workdir, "heightfield.obj"), "w") as f:
        np.savetxt(f, P, fmt="v %.6f %.6f %.6f"); np.savetxt(f, F, fmt="f %d %d %d")
    return cfg, root
