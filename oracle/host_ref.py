"""ORACLE (test infrastructure) — plain-Python / numpy restatement of the *host-side* parts of the path:
BVHAccel::new (bvh.rs:307-751, HLBVH with Q26/Q27), the Halton index helpers (samplers/halton.rs:23-61,131-150)
and RealisticCamera::new's focusing + exit-pupil bound (camera.rs:66-135,332-358,421-480).

Used only by tests/ to check the C++ host scene builder in rs_ray_toy_amd/csrc/host on small inputs.
Parity status: pinned by samples/cube.obj + samples/scene.json (36 instanced triangles) and the Halton
constants of SURVEY §8c; partition_in_place's swap order follows the documented nightly-std algorithm
("parity unpinned" at that one boundary)."""
import math

import numpy as np

F64_MAX = np.finfo(np.float64).max


def _left_shift3(x):
    assert x <= (1 << 10)
    if x == (1 << 10):
        x -= 1
    x = (x | (x << 16)) & 0b00000011000000000000000011111111
    x = (x | (x << 8)) & 0b00000011000000001111000000001111
    x = (x | (x << 4)) & 0b00000011000011000011000011000011
    x = (x | (x << 2)) & 0b00001001001001001001001001001001
    return x


def _as_u32(v):
    if v != v or v <= 0:
        return 0
    return min(int(v), 0xFFFFFFFF)


def build_bvh(bounds, max_prims_in_node=4, fix_slice=False, fix_sah=False):
    """bounds: (n, 6) world bounds of the primitives in aggregate order. Returns (nodes, order) where nodes is a
    list of (bmin3+bmax3, offset, n_primitives, axis) in flattened pre-order and order the ordered_prims."""
    bounds = np.asarray(bounds, np.float64)
    n = len(bounds)
    cent = (bounds[:, :3] + bounds[:, 3:]) * 0.5
    cmin, cmax = cent.min(0), cent.max(0)
    codes = []
    for i in range(n):
        o = cent[i] - cmin
        for k in range(3):
            if cmax[k] > cmin[k]:
                o[k] /= cmax[k] - cmin[k]
        v = o * 1024.0
        codes.append((_left_shift3(_as_u32(v[2])) << 2) | (_left_shift3(_as_u32(v[1])) << 1) | _left_shift3(_as_u32(v[0])))
    mp = list(zip(range(n), codes))
    for p in range(5):                      # LSD radix sort, 6 bits per pass, stable
        buckets = [[] for _ in range(64)]
        for e in mp:
            buckets[(e[1] >> (6 * p)) & 63].append(e)
        mp = [e for b in buckets for e in b]
    build = []                              # (bounds6, child0, child1, axis, first, nprims)
    order = []
    total = [0]

    def union(a, b):
        return np.concatenate([np.minimum(a[:3], b[:3]), np.maximum(a[3:], b[3:])])

    def emit(sl, bit):
        nn = len(sl)
        if bit == -1 or nn < max_prims_in_node:
            total[0] += 1
            b = np.array([F64_MAX] * 3 + [-F64_MAX] * 3)
            first = len(order)
            for pi, _ in sl:
                order.append(pi)
                b = union(b, bounds[pi])
            build.append((b, -1, -1, 0, first, nn))
            return len(build) - 1
        mask = 1 << bit
        if (sl[0][1] & mask) == (sl[-1][1] & mask):
            return emit(sl, bit - 1)
        s, e = 0, nn - 1
        while s + 1 != e:
            mid = (s + e) // 2
            if (sl[s][1] & mask) == (sl[mid][1] & mask):
                s = mid
            else:
                e = mid
        split = e
        total[0] += 1
        c0 = emit(sl[:split], bit - 1)
        c1 = emit(sl[split:] if fix_slice else sl[:nn - split], bit - 1)   # Q26
        build.append((union(build[c0][0], build[c1][0]), c0, c1, bit % 3, 0, 0))
        return len(build) - 1

    roots = []
    start = 0
    for end in range(1, n + 1):
        if end == n or (mp[start][1] & 0x3FFC0000) != (mp[end][1] & 0x3FFC0000):
            roots.append(emit(mp[start:end], 17))
            start = end

    def sa(b):
        with np.errstate(over="ignore", invalid="ignore"):
            d = b[3:] - b[:3]
            r = d[0] * d[1] + d[0] * d[2] + d[1] * d[2]
            return r + r

    def upper(lo, hi):
        if hi - lo == 1:
            return roots[lo]
        total[0] += 1
        bb = np.array([F64_MAX] * 3 + [-F64_MAX] * 3)
        cb = np.array([F64_MAX] * 3 + [-F64_MAX] * 3)
        for i in range(lo, hi):
            b = build[roots[i]][0]
            bb = union(bb, b)
            c = (b[:3] + b[3:]) * 0.5
            cb = union(cb, np.concatenate([c, c]))
        d = cb[3:] - cb[:3]
        dim = 0 if (d[0] > d[1] and d[0] > d[2]) else (1 if d[1] > d[2] else 2)
        assert cb[3 + dim] != cb[dim]

        def bucket(b):
            c = (b[dim] + b[3 + dim]) * 0.5
            k = int(12 * ((c - cb[dim]) / (cb[3 + dim] - cb[dim])))
            return 11 if k == 12 else k
        cnt = [0] * 12
        bnd = [np.array([F64_MAX] * 3 + [-F64_MAX] * 3) for _ in range(12)]
        for i in range(lo, hi):
            k = bucket(build[roots[i]][0])
            cnt[k] += 1
            bnd[k] = union(bnd[k], build[roots[i]][0])
        costs = []
        for i in range(11):
            b0 = np.array([F64_MAX] * 3 + [-F64_MAX] * 3)
            b1 = b0.copy()
            c0 = c1 = 0
            for j in range(i + 1 if fix_sah else i):          # Q27: bucket i in neither side
                b0 = union(b0, bnd[j]); c0 += cnt[j]
            for j in range(i + 1, 12):
                b1 = union(b1, bnd[j]); c1 += cnt[j]
            with np.errstate(over="ignore", invalid="ignore"):
                costs.append(0.125 + (c0 * sa(b0) + c1 * sa(b1)) / sa(bb))
        mc, mb = costs[0], 0
        for i in range(1, 11):
            if costs[i] < mc:
                mc, mb = costs[i], i
        first, last = lo, hi                                   # partition_in_place
        while True:
            while first != last and bucket(build[roots[first]][0]) <= mb:
                first += 1
            if first == last:
                break
            last -= 1
            while first != last and not (bucket(build[roots[last]][0]) <= mb):
                last -= 1
            if first == last:
                break
            roots[first], roots[last] = roots[last], roots[first]
            first += 1
        mid = first
        assert lo < mid < hi
        c0 = upper(lo, mid)
        c1 = upper(mid, hi)
        build.append((union(build[c0][0], build[c1][0]), c0, c1, dim, 0, 0))
        return len(build) - 1

    root = upper(0, len(roots))
    nodes = [None] * total[0]
    off = [0]

    def flatten(i):
        b, c0, c1, axis, first, nn = build[i]
        my = off[0]
        off[0] += 1
        if nn > 0:
            nodes[my] = (b, first, nn, 0)
        else:
            flatten(c0)
            second = flatten(c1)
            nodes[my] = (b, second, 0, axis)
        return my
    flatten(root)
    assert off[0] == total[0]
    return nodes, order


def transform_bounds(m, b):
    """Transformable for Bounds3f (transform.rs:539-612): union of the 8 transformed corners."""
    m = np.asarray(m, np.float64).reshape(4, 4)
    pts = []
    for cx in (b[0], b[3]):
        for cy in (b[1], b[4]):
            for cz in (b[2], b[5]):
                pts.append(m[:3, :3] @ np.array([cx, cy, cz]) + m[:3, 3])
    pts = np.array(pts)
    return np.concatenate([pts.min(0), pts.max(0)])


# ---- Halton helpers ----------------------------------------------------------------------------------------
def halton_params(xres, yres):
    scales, exps = [], []
    for i, res in enumerate((xres, yres)):
        base = 2 if i == 0 else 3
        scale, e = 1, 0
        while scale < min(res, 128):
            scale *= base
            e += 1
        scales.append(scale); exps.append(e)

    def egcd(a, b):
        if b == 0:
            return 1, 1                 # the reference's base case (Q24)
        d = a // b
        xp, yp = egcd(b, a % b)
        return yp, xp - d * yp

    def minv(a, n):
        x, _ = egcd(a, n)
        return (x % (1 << 64)) % n
    stride = scales[0] * scales[1]
    return scales, exps, stride, [minv(scales[1], scales[0]), minv(scales[0], scales[1])]


# ---- RealisticCamera init, vectorised over rays ---------------------------------------------------------------
def _quadratic(a, b, c):
    disc = b * b - 4.0 * a * c
    ok = disc >= 0
    root = np.sqrt(np.where(ok, disc, 0.0))
    q = np.where(b < 0, -0.5 * (b - root), -0.5 * (b + root))
    with np.errstate(divide="ignore", invalid="ignore"):
        t0, t1 = q / a, c / q
    lo, hi = np.minimum(t0, t1), np.maximum(t0, t1)
    return ok, lo, hi


def _norm(v):
    l = np.sqrt((v * v).sum(-1, keepdims=True))
    return np.where(l == 0, v, v / np.where(l == 0, 1, l))


def trace_from_film(elems, o, d):
    """elems: (n,4) [curvature_radius, thickness, eta, aperture_radius] in metres. o, d: (N,3) camera-space rays.
    Returns alive mask (trace_lenses_from_film succeeded)."""
    o = o * np.array([1.0, 1.0, -1.0]); d = _norm(_norm(d * np.array([1.0, 1.0, -1.0])))
    alive = np.ones(len(o), bool)
    z = 0.0
    for i in range(len(elems) - 1, -1, -1):
        R, th, eta, ap = elems[i]
        z -= th
        if R == 0.0:
            alive &= ~(d[:, 2] >= 0.0)
            with np.errstate(divide="ignore", invalid="ignore"):
                t = (z - o[:, 2]) / d[:, 2]
            n = None
        else:
            zc = z + R
            oo = o - np.array([0.0, 0.0, zc])
            a = (d * d).sum(-1); b = 2.0 * (d * oo).sum(-1); c = (oo * oo).sum(-1) - R * R
            ok, lo, hi = _quadratic(a, b, c)
            closer = (d[:, 2] > 0.0) ^ (R < 0.0)
            t = np.where(closer, lo, hi)
            alive &= ok & ~(t < 0.0)
            n = oo + d * t[:, None]
            n = n / np.sqrt((n * n).sum(-1, keepdims=True))
            flip = (n * -d).sum(-1) < 0.0
            n = np.where(flip[:, None], -n, n)
        t = np.where(alive, t, 0.0)
        p = o + d * t[:, None]
        alive &= ~((p[:, 0] ** 2 + p[:, 1] ** 2) >= ap * ap)
        o = p
        if R != 0.0:
            eta_t = elems[i - 1][2] if (i > 0 and elems[i - 1][2] != 0.0) else 1.0
            e = eta / eta_t
            wi = _norm(-d)
            ci = (n * wi).sum(-1)
            s2i = np.maximum(0.0, 1.0 - ci * ci)
            s2t = e * e * s2i
            alive &= ~(s2t >= 1.0)
            ct = np.sqrt(np.maximum(0.0, 1.0 - s2t))
            d = -wi * e + n * (e * ci - ct)[:, None]
    return alive


def radical_inverse_2(i):
    i = np.asarray(i, np.uint64)
    r = np.zeros_like(i)
    v = i.copy()
    for _ in range(64):
        r = (r << np.uint64(1)) | (v & np.uint64(1))
        v >>= np.uint64(1)
    return r.astype(np.float64) * 2.0 ** -64


def radical_inverse_3(i):
    i = np.asarray(i, np.uint64).copy()
    rev = np.zeros_like(i)
    inv_n = np.ones(len(i))
    inv = 1.0 / 3.0
    while (i != 0).any():
        nz = i != 0
        nxt = i // np.uint64(3)
        dig = i - nxt * np.uint64(3)
        rev = np.where(nz, rev * np.uint64(3) + dig, rev)
        inv_n = np.where(nz, inv_n * inv, inv_n)
        i = nxt
    return np.minimum(rev.astype(np.float64) * inv_n, 1.0 - 2.0 ** -53)


def bound_exit_pupil(elems, x0, x1, n_samples=1024 * 1024):
    """camera.rs:421-480 incl. the zero-box default and the shifting expand() (Q7)."""
    rear_r = elems[-1][3]
    lo, hi = -1.5 * rear_r, 1.5 * rear_r
    i = np.arange(n_samples, dtype=np.uint64)
    tt = (i.astype(np.float64) + 0.5) / n_samples
    pf = np.stack([x0 * (1.0 - tt) + x1 * tt, np.zeros(n_samples), np.zeros(n_samples)], -1)
    u0, u1 = radical_inverse_2(i), radical_inverse_3(i)
    pr = np.stack([lo * (1.0 - u0) + hi * u0, lo * (1.0 - u1) + hi * u1, np.full(n_samples, elems[-1][1])], -1)
    ok = trace_from_film(elems, pf, _norm(pr - pf))
    if not ok.any():
        return [lo, lo, hi, hi]
    px, py = pr[ok, 0], pr[ok, 1]
    mnx, mny, mxx, mxy = min(0.0, px.min()), min(0.0, py.min()), max(0.0, px.max()), max(0.0, py.max())
    delta = 2.0 * math.sqrt(2.0 * (hi - lo) ** 2) / math.sqrt(n_samples)
    a, b = (mnx - delta, mny - delta), (mxx - delta, mxy - delta)
    return [min(a[0], b[0]), min(a[1], b[1]), max(a[0], b[0]), max(a[1], b[1])]


# ---- ImageTexture on the host: load_image (renderprocess.rs:532-561) + MIPMap::create (mipmap.rs:270-382) over the
# ---- BlockedArray of memory.rs:24-98, restated with numpy for tests/test_host.py ------------------------------------------
def ba_index(u_blocks, u, v):
    """BlockedArray's index expression memory.rs:76-85 (block and offset swapped relative to pbrt: texels alias)."""
    return 16 * (u_blocks * (v & 3) + (u & 3)) + 4 * (v >> 2) + (u >> 2)


class BlockedLevel:
    def __init__(self, u_res, v_res):
        self.u_res, self.v_res = u_res, v_res
        self.u_blocks = ((u_res + 3) & ~3) >> 2
        self.data = np.zeros((((u_res + 3) & ~3) * ((v_res + 3) & ~3), 3))

    def set(self, u, v, rgb):
        self.data[ba_index(self.u_blocks, u, v)] = rgb

    def get(self, u, v):
        return self.data[ba_index(self.u_blocks, u, v)]


def _usize(v):
    return 0 if not (v > 0.0) else int(v)


def _lanczos(x, tau):
    x = abs(x)
    if x < 1e-5:
        return 1.0
    if x > 1.0:
        return 0.0
    x *= math.pi
    return (math.sin(x * tau) / (x * tau)) * (math.sin(x) / x)


def resample_weights(old_res, new_res):
    out = []
    for i in range(new_res):
        center = (i + 0.5) * old_res / new_res
        first = _usize(math.floor(center - 2.0 + 0.5))
        w = [_lanczos(((first + j) + 0.5 - center) / 2.0, 2.0) for j in range(4)]
        inv = 1.0 / (w[0] + w[1] + w[2] + w[3])
        out.append((first, [x * inv for x in w]))
    return out


def mip_texel(level, wrap, s, t):
    ts = tt = 0
    if wrap == 0:
        ts, tt = s % level.u_res, t % level.v_res
    elif wrap == 1:
        if s >= level.u_res or t >= level.v_res:
            return np.zeros(3)
    else:
        ts, tt = min(s, level.u_res), min(t, level.v_res)
    return level.get(ts, tt)


def build_mipmap(rgb8, wrap=0):
    """rgb8: (h, w, 3) uint8 as decoded (top row first). Returns the pyramid as a list of BlockedLevel."""
    img = rgb8.astype(np.float64) / 255.0
    h, w = img.shape[:2]
    img = img.copy()
    for y in range(h // 2):
        tmp = img[y].copy(); img[y] = img[h - 1 - y]; img[h - 1 - y] = tmp
    pow2 = lambda v: v != 0 and (v & (v - 1)) == 0
    rup = lambda v: 1 << (v - 1).bit_length()
    if not pow2(w) or not pow2(h):
        pw, ph = rup(w), rup(h)
        sw = resample_weights(w, pw)
        res = np.zeros((ph, pw, 3))
        for t in range(h):
            for x in range(pw):
                acc = np.zeros(3)
                for j in range(4):
                    o = sw[x][0] + j
                    if wrap == 0: o = o % w
                    elif wrap == 2: o = min(max(o, 0), w - 1)
                    if o < w: acc = acc + img[t, o] * sw[x][1][j]
                res[t, x] = acc
        tw = resample_weights(h, ph)
        for x in range(pw):
            work = np.zeros((ph, 3))
            for t in range(ph):
                for j in range(4):
                    o = tw[t][0] + j
                    if wrap == 0: o = o % h
                    elif wrap == 2: o = min(max(o, 0), h - 1)
                    if o < h: work[t] = work[t] + res[o, x] * tw[t][1][j]
            for t in range(ph):
                res[t, x] = np.maximum(work[t], 0.0)
        img, w, h = res, pw, ph
    n_levels = 1 + int(math.log2(max(w, h)))
    lv = BlockedLevel(w, h)
    for u in range(w):
        for v in range(h):
            lv.set(u, v, img[v, u])
    pyr = [lv]
    for i in range(1, n_levels):
        sr, tr = max(pyr[-1].u_res // 2, 1), max(pyr[-1].v_res // 2, 1)
        if min(sr, tr) < 64:
            break
        nl = BlockedLevel(sr, tr)
        p = pyr[-1]
        for t in range(tr):
            for s in range(sr):
                nl.set(s, t, (((mip_texel(p, wrap, 2 * s, 2 * t) + mip_texel(p, wrap, 2 * s + 1, 2 * t)) + mip_texel(p, wrap, 2 * s, 2 * t + 1)) + mip_texel(p, wrap, 2 * s + 1, 2 * t + 1)) * 0.25)
        pyr.append(nl)
    return pyr
