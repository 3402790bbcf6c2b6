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


def boxes_on_a_plane(workdir, seed, n_boxes=7):
    """A flat ground of 8 x 8 quads with boxes that rest ON it (bottom faces coplanar with ground triangles, side faces rising from lines inside them), float above
    it (one of them by a hair: 1e-6), sink into it (triangles that cross), lean on an edge, and touch each other face to face: everything the horizon builder's special cases exist for, as plain
    world-space triangles of one mesh (the file replaces config 4's heightfield)."""
    rng = np.random.default_rng(seed)
    cfg, root = scenes.cfg4(workdir, xres=64, yres=64, nsamp=5, max_depth=5, n=8)
    V, F = [], []

    def add(verts, faces):
        base = len(V)
        V.extend(verts)
        F.extend([(a + base + 1, b + base + 1, c + base + 1) for a, b, c in faces])

    g = 9
    xs = np.linspace(-10.0, 10.0, g)
    ground = [(35.0 + x, -2.0, z) for x in xs for z in xs]
    idx = np.arange(g * g).reshape(g, g)
    a, b, cc, d = idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel(), idx[:-1, 1:].ravel()
    add(ground, [tuple(t) for t in np.concatenate([np.stack([a, cc, b], 1), np.stack([a, d, cc], 1)])])
    corners = np.array([(sx, sy, sz) for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], float)
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    tri = [(q[0], q[1], q[2]) for q in quads] + [(q[0], q[2], q[3]) for q in quads]

    def box(center, half, rot_y=0.0, rot_x=0.0):
        c, s = np.cos(rot_y), np.sin(rot_y)
        ry = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])
        c, s = np.cos(rot_x), np.sin(rot_x)
        rx = np.array([[1, 0, 0], [0, c, -s], [0, s, c]])
        pts = (corners * np.asarray(half)) @ rx.T @ ry.T + np.asarray(center)
        add([tuple(p) for p in pts], tri)

    kinds = ["rest", "float", "sunk", "lean", "pair", "hair"]
    for i in range(n_boxes):
        kind = kinds[i % len(kinds)]
        x, z = 35.0 + rng.uniform(-7, 7), rng.uniform(-7, 7)
        h = rng.uniform(0.4, 1.6, 3)
        if kind == "rest": box((x, -2.0 + h[1], z), h, rot_y=rng.uniform(0, 3))
        elif kind == "float": box((x, -2.0 + h[1] + rng.uniform(0.2, 2.5), z), h, rot_y=rng.uniform(0, 3), rot_x=rng.uniform(0, 0.5))
        elif kind == "sunk": box((x, -2.0 + 0.3 * h[1], z), h, rot_y=rng.uniform(0, 3), rot_x=rng.uniform(0, 0.4))
        elif kind == "hair": box((x, -2.0 + h[1] + 1e-6, z), h, rot_y=rng.uniform(0, 3))      # a hair above the ground: not coplanar, closer than an origin's rounding
        elif kind == "lean": box((x, -2.0 + h[1] * np.sqrt(2.0), z), (h[0], h[1], h[1]), rot_x=np.pi / 4)      # an edge (about) on the ground
        else:
            box((x, -2.0 + h[1], z), h)
            box((x + h[0] + 0.5, -2.0 + 0.7 * h[1], z), (0.5, 0.7 * h[1], h[2]))      # face to face with the first one
    with open(os.path.join(workdir, "heightfield.obj"), "w") as f:
        np.savetxt(f, np.array(V), fmt="v %.9g %.9g %.9g"); np.savetxt(f, np.array(F), fmt="f %d %d %d")
    return cfg, root
