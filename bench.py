#!/usr/bin/env python3
"""bench.py — BASELINE.json headline: Mrays/s + achieved GB/s, 100k-triangle scene @ 1024^2, 256 spp, depth 8.

One "step" = one full frame of BASELINE config 4 (procedural 100 352-triangle heightfield, Path integrator
max_depth 8, HaltonSampler nsamp 257 = 256 effective spp, RealisticCamera) rendered by the HIP wavefront path.
With N > 1 ranks (torchrun, one process per GPU) the film is partitioned into interleaved 16-row bands, every
rank renders its bands into a device film, and ONE collective per frame reassembles the image on rank 0: the
product's own rrt_film_gather (C ABI, rs_ray_toy_amd/csrc/device/rrt_comm.hip: grouped ncclSend / ncclRecv of the
band rows over xGMI, enqueued on the frame's stream) on an rrt_comm whose id rank 0 broadcasts through
torch.distributed; `--dist-backend gloo` (rehearsal with more ranks than GPUs) keeps torch.distributed.reduce.
The collective is inside the timed region. Total work is fixed: strong scaling.

Inputs are resident in HBM before the timed region (scene upload + pool allocation happen in setup/warm-up).
The JSON line carries `roofline` for the dominant kernel (closest-hit BVH traversal + triangle tests) and
`cpu_baseline` (the f64 oracle on the host cores, bounded sample, rank 0 at N=1 only).
"""
import argparse
import ctypes as C
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.3 TB/s measured streaming copy)
# MI355X_MICROARCH.md "Indexed rows: gather": rows served from the XCD's L2 16.8-18.8 TB/s chip-wide, from the Infinity Cache 8.6 TB/s.
# The 11 MB BVH (pair nodes + triangles) does not fit one XCD's 4 MiB L2, so its gathered 64-B lines come from a mix of both.
GATHER_L2_GBS, GATHER_IC_GBS = 17800.0, 8600.0
# vector L1 (TCP) of a CU: 39 cycles per 16-byte load whose 64 lanes read distinct lines, L1 hits - 0.61 cycles per lane-level access - but
# 1.0 per access when the active lanes have no active neighbour (32 alternate lanes: 32 cycles; 16 of 64: 16) and 0.63 for adjacent pairs
# (tools/micro/tcp_gather2.hip, measured on MI355X). For a wave whose lanes are active with probability p: 1 - 0.39 p cycles per access.
TCP_CYCLES_PER_ACCESS, TCP_CYCLES_PER_ACCESS_SPARSE, N_CUS, CLOCK_HZ = 39.0 / 64.0, 1.0, 256, 2.4e9


def l1_gather_bound(accesses, load_insts, launch_s, alone_s):
    """How busy the CUs' vector L1 (TCP) is during a closest-hit launch: lane-level accesses (PMC TCP_TOTAL_CACHE_ACCESSES) x the cycles each
    occupies the TCP, over 256 CUs x duration x 2.4 GHz. The cost per access depends on how full the loading waves are (see the constants):
    `frac_dense` prices every access as in a full wave (a lower bound), `frac` at the measured density p = accesses / (64 x wave-level vector
    loads, PMC SQ_INSTS_VMEM_RD) with 1 - 0.39 p cycles per access; L1 misses (13 % of the accesses) are not priced."""
    p = None if not load_insts else min(1.0, accesses / (64.0 * load_insts))
    c = TCP_CYCLES_PER_ACCESS if p is None else 1.0 - (1.0 - TCP_CYCLES_PER_ACCESS) * p
    cap = lambda t: N_CUS * t * CLOCK_HZ
    return {"tcp_accesses_per_launch": round(accesses, 1), "lanes_per_load": None if p is None else round(64.0 * p, 1), "cycles_per_access": round(c, 3),
            "cycles_per_access_dense": round(TCP_CYCLES_PER_ACCESS, 3), "frac": round(accesses * c / cap(launch_s), 4), "frac_alone": round(accesses * c / cap(alone_s), 4),
            "frac_dense": round(accesses * TCP_CYCLES_PER_ACCESS / cap(launch_s), 4), "frac_alone_dense": round(accesses * TCP_CYCLES_PER_ACCESS / cap(alone_s), 4)}


def kernel_source_hash():
    """Identifies the device code a PMC file was measured on (tools/pmc_traffic.py stores it): a stale file must not feed `traffic`."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "rs_ray_toy_amd", "csrc", "device", "*"))):
        if f.endswith((".hpp", ".hip")):
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def cpu_baseline(scene, target_seconds):
    """The f64 oracle (a port: the reference itself is a nightly-only Rust crate that cannot be built here) on
    this box's host cores, OpenMP over 16x16 tiles like the reference's rayon loop, sampler tables built once
    ("fair" variant of BASELINE.md §2). Bounded sample: whole 16-row tile bands from the middle of the same
    frame, at the full 256 spp, sized from a one-band calibration to take about `target_seconds`."""
    import oracle_lib as O
    W, H = scene.resolution
    cores = os.cpu_count() or 1
    mid = (H // 2) // 16 * 16
    t0 = time.perf_counter()
    _, st = O.render(scene, (0, mid, W, mid + 16), stats=True)
    dt = time.perf_counter() - t0
    bands = int(max(1, min((H - mid) // 16 - 1, round(target_seconds / max(dt, 1e-3)) - 1)))
    q, rays, secs, rows = st.closest_queries + st.any_queries, st.camera_rays, dt, 16
    if bands >= 1:
        t0 = time.perf_counter()
        _, st2 = O.render(scene, (0, mid + 16, W, mid + 16 + 16 * bands), stats=True)
        dt2 = time.perf_counter() - t0
        q += st2.closest_queries + st2.any_queries
        rays += st2.camera_rays
        secs += dt2
        rows += 16 * bands
    # "reference-faithful" variant: the reference rebuilds the 3.67 M-entry Halton permutation table
    # for every 16x16 tile (integrator/mod.rs:73); one band with that cost model, reported next to the fair number
    t0 = time.perf_counter()
    _, stf = O.render(scene, (0, mid, W, mid + 16), stats=True, faithful_sampler_rebuild=True)
    dtf = time.perf_counter() - t0
    faithful = (stf.closest_queries + stf.any_queries) / dtf / 1e6
    return {"value": round(q / secs / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "faithful_value": round(faithful, 4),
            "sample": "rows %d..%d of the same %dx%d frame at full spp/depth (%d ray queries, %.1f s, f64 oracle incl. the "
                      "reference's BSDF-sampled MIS ray and final dead bounce)" % (mid, mid + rows, W, H, q, secs),
            "camera_mrays_per_s": round(rays / secs / 1e6, 4), "est_full_frame_s": round(secs * H / rows, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="cfg4", choices=("cfg4", "cfg5"),
                    help="cfg4 (default): BASELINE.json's headline configuration (configs[3]); cfg5: configs[4] - the same mesh with Plastic / Metal "
                         "microfacet materials and 4 sphere area lights, 2048^2, 1024 spp, depth 16 (one step = one such frame, ~0.6 s)")
    ap.add_argument("--res", type=int, default=0, help="default: the configuration's own (1024 / 2048)")
    ap.add_argument("--spp", type=int, default=0, help="default: the configuration's own (256 / 1024)")
    ap.add_argument("--depth", type=int, default=0, help="default: the configuration's own (8 / 16)")
    ap.add_argument("--grid", type=int, default=224, help="heightfield cells per side (224 -> 100 352 triangles)")
    ap.add_argument("--compat-bvh", action="store_true", help="reference-exact builder incl. Q26/Q27 (default: fixed-bvh)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target wall time of the cpu_baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--max-paths", type=int, default=0)
    ap.add_argument("--opt", action="append", default=[], help="handle option key=value (rrt_set_option), e.g. pt_split_any=1e9")
    ap.add_argument("--dist-backend", default="nccl", choices=("nccl", "gloo"),
                    help="gloo: rehearse the multi-rank path with more ranks than GPUs (ranks share devices, films reduced on the host)")
    ap.add_argument("--frames-in-flight", type=int, default=0, choices=(0, 1, 2, 3, 4),
                    help="0 (default): 2 on one GPU, 4 on several (a rank's pools shrink with its share of the film, and its shorter "
                         "launches leave more of the chip to fill: tools/band_pipeline.py); n >= 2: alternate n handles (rrt_render_bands_begin / _end): a frame's latency-bound last bounces drain "
                         "while the next frame's camera rays fill the chip; 1: one synchronous frame at a time")
    args = ap.parse_args()
    full_size = {"cfg4": (1024, 256, 8), "cfg5": (2048, 1024, 16)}[args.config]
    args.res, args.spp, args.depth = args.res or full_size[0], args.spp or full_size[1], args.depth or full_size[2]
    at_full_size = (args.res, args.spp, args.depth) == full_size and args.grid == 224

    import numpy as np
    import torch
    import torch.distributed as dist

    from rs_ray_toy_amd import RRT_F32, RRT_FIXED_BVH, Renderer, RrtError, Scene, scenes
    from rs_ray_toy_amd.api import Comm
    from rs_ray_toy_amd.partition import reduce_film

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists for the product path)")
    if args.dist_backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()   # rehearsal: ranks may share a device
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    # ---- setup (untimed): scene build on the host, upload, pools ------------------------------------------------
    wd = tempfile.mkdtemp(prefix=f"rrt_bench_r{rank}_")
    make = scenes.cfg4 if args.config == "cfg4" else scenes.cfg5
    cfg, root = make(wd, xres=args.res, yres=args.res, nsamp=args.spp + 1, max_depth=args.depth, n=args.grid)
    flags = 0 if args.compat_bvh else RRT_FIXED_BVH
    t0 = time.time()
    scene = Scene.loads(cfg, root, flags=flags)
    t_build = time.time() - t0
    W, H = scene.resolution
    nfl = args.frames_in_flight or (2 if world == 1 else 4)
    t0 = time.time()
    handles = [Renderer(scene, local_rank, RRT_F32) for _ in range(nfl)]
    t_handles = time.time() - t0      # upload + the device-side tables (pair nodes, tile trees on first use, shadow lists, horizon tables)
    for h in handles:
        if args.max_paths:
            h.set_option("max_paths", args.max_paths)
        for kv in args.opt:
            k, v = kv.split("=")
            h.set_option(k, float(v))
        h.set_option("frame_stats", 1)               # kernel timings of the frames in flight (HIP events on the handle's own streams)
        if nfl > 1:
            h.set_option("nonblocking_streams", 1)   # frames of the two handles may overlap; ordering with torch's stream is explicit below
    r = handles[0]
    films = [torch.zeros((H, W, 4), dtype=torch.float32, device=f"cuda:{local_rank}") for _ in range(nfl)]
    film = films[0]
    # The frame's one collective. RCCL ranks: the product's rrt_film_gather on an rrt_comm (ncclCommInitRank behind the C ABI; the id travels
    # from rank 0 through torch.distributed). gloo rehearsal (ranks may share a GPU, which RCCL does not allow): torch.distributed.reduce.
    comm = None
    collective = "none (one rank)"
    if world > 1 and args.dist_backend == "nccl":
        ident = torch.zeros(128, dtype=torch.uint8, device=f"cuda:{local_rank}")
        if rank == 0:
            ident = torch.tensor(list(Comm.new_id()), dtype=torch.uint8, device=f"cuda:{local_rank}")
        dist.broadcast(ident, 0)
        comm = Comm(bytes(ident.cpu().tolist()), rank, world, local_rank)
        collective = "rrt_film_gather (C ABI: grouped ncclSend/ncclRecv of the band rows to rank 0 over RCCL, on the frame's stream)"
    elif world > 1:
        collective = "torch.distributed.reduce over gloo (rehearsal: ranks may share a device)"

    def frame_begin(k):
        """Enqueue one frame on handle k: this rank's bands into films[k], then (RCCL ranks) the gather to rank 0 on the same stream."""
        films[k].zero_()                              # (ordered after that film's previous use on torch's stream)
        torch.cuda.current_stream().synchronize()     # the handle's streams do not wait for torch's stream
        handles[k].render_bands_begin(rank, world, films[k].data_ptr())
        if comm is not None:
            comm.gather(handles[k], films[k].data_ptr(), 0)

    def frame_end(k):
        st = handles[k].render_end(stats=True)        # waits for the frame and, behind it on the stream, the gather
        if world > 1 and comm is None:
            reduce_film(films[k], world)              # gloo: disjoint bands, the sum reassembles the frame on rank 0
        return {kk: getattr(st, kk) for kk, _ in st._fields_}

    def step(collect=False):
        frame_begin(0)
        agg = frame_end(0)
        return agg if collect else None

    frame_log = []   # per-frame statistics of the frames run_frames() completed (kernel timings: HIP events on the handle's streams)

    def run_frames(n):
        """n frames (steps), at most `nfl` in flight: frame i renders on handle i % nfl while frame i - 1 finishes on the other one;
        each frame ends with its film reduced to rank 0, all inside the caller's timed region."""
        for i in range(n):
            k = i % nfl
            if i >= nfl:
                frame_log.append(frame_end(k))
            frame_begin(k)
        for i in range(max(0, n - nfl), n):
            frame_log.append(frame_end(i % nfl))

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # one counting frame (untimed): exact per-kernel ray / node / triangle-test counts for the byte model
    r.set_option("count_traversal", 1)
    counted = step(collect=True)
    r.set_option("count_traversal", 0)
    # Warm-up = pool allocation: every handle renders one frame of its own, WITHOUT the collective, so that a rank that cannot hold `nfl`
    # sets of wavefront pools (free HBM) finds out before any reduce is issued; the ranks then agree (MAX over ranks) on one frame at a
    # time or `nfl`, and only then run the warm-up frames - every rank issues the same sequence of collectives whatever happened.
    failed = 0
    if nfl > 1:
        try:
            for k in range(nfl):
                films[k].zero_()
                torch.cuda.current_stream().synchronize()
                handles[k].render_bands_begin(rank, world, films[k].data_ptr())   # (no collective here: see above)
                handles[k].render_end()
        except RrtError as e:
            print(f"[rank {rank}] {nfl} frames in flight not possible here ({e})", file=sys.stderr)
            failed = 1
    if world > 1:
        flag = torch.tensor([failed], dtype=torch.int32, device="cpu" if args.dist_backend == "gloo" else f"cuda:{local_rank}")
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        failed = int(flag.item())
    if failed:
        for h in handles:
            try:
                h.render_end()
            except RrtError:
                pass
        for h in handles[1:]:
            h.close()
        nfl = 1
        del handles[1:], films[1:]
    run_frames(args.warmup)
    sync()
    del frame_log[:]
    t0 = time.perf_counter()
    run_frames(args.steps)
    sync()
    elapsed = time.perf_counter() - t0
    # per-kernel durations: HIP events on the handles' own streams around every launch of every frame of the timed region, averaged per frame
    assert len(frame_log) == args.steps
    timed = {k: sum(f[k] for f in frame_log) / len(frame_log) for k in frame_log[0]}
    # Kernel durations for the roofline (after the timed region, not part of `value`): the same frames ONE AT A TIME on the first handle. With
    # several frames in flight an event pair around a launch also holds the time the stream waits for compute units behind the other frame's
    # kernels, and rocprofv3's per-dispatch duration of a persistent launch holds the time it runs on the few CUs it got first - neither is the
    # kernel's duration. One frame at a time both views coincide (profiles/: kernel_stats_1flight.csv of `bench.py --frames-in-flight 1`).
    step()
    single = [step(collect=True) for _ in range(max(2, args.steps // 2))]
    single = {k: sum(f[k] for f in single) / len(single) for k in single[0]}
    # ... and with the shadow launches back on the main stream: the closest-hit kernel alone on the chip
    r.set_option("overlap_shadow", 0)
    step()
    isolated = step(collect=True)
    r.set_option("overlap_shadow", 1)
    sync()

    stat_dev = "cpu" if args.dist_backend == "gloo" else f"cuda:{local_rank}"
    tt = torch.tensor([elapsed], dtype=torch.float64, device=stat_dev)
    keys = ["camera_samples", "camera_rays", "closest_queries", "any_queries", "closest_nodes", "closest_prims", "any_nodes", "any_prims", "closest_launches", "root_culled", "sky_culled"]
    counted["closest_launches"] = timed["closest_launches"]
    counted["root_culled"] = timed["root_culled"]   # (the counting frame keeps every camera ray in the queue; the horizon cull - geometry, not a kernel's test - applies to it too)
    cnt = torch.tensor([float(counted[k]) for k in keys] + [timed["ms_closest"], timed["ms_any"], timed["ms_shade"], timed["ms_raygen"], timed["ms_film"], timed["ms_total"], isolated["ms_closest"],
                                                            single["ms_closest"], single["ms_any"], single["ms_shade"], single["ms_raygen"], single["ms_film"], single["ms_total"],
                                                            timed["ms_gather"]],
                       dtype=torch.float64, device=stat_dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        mx, mn = cnt.clone(), cnt.clone()
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(mn, op=dist.ReduceOp.MIN)
    else:
        mx = mn = cnt
    elapsed = float(tt.item())
    names = keys + ["ms_closest", "ms_any", "ms_shade", "ms_raygen", "ms_film", "ms_total", "ms_closest_isolated",
                    "ms1_closest", "ms1_any", "ms1_shade", "ms1_raygen", "ms1_film", "ms1_total", "ms_gather"]
    tot = dict(zip(names, cnt.tolist()))
    mx_tot = dict(zip(names, mx.tolist()))
    mn_tot = dict(zip(names, mn.tolist()))

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        queries = tot["closest_queries"] + tot["any_queries"]
        value = queries / (ms_per_step * 1e-3) / 1e6
        # Roofline of the dominant kernel: closest-hit BVH traversal. ALGORITHMIC bytes per closest query: 28 B ray in +
        # 16 B hit out + 32 B per BVH node visited + 48 B per triangle tested; node / triangle counts are exact device counters of a
        # counting frame of this very workload; duration = the kernel's launches, HIP events on the stream they were launched on, in the
        # one-frame-at-a-time pass above (shadow launches beside them on the second stream, as in the product): what rocprofv3's kernel trace
        # of `bench.py --frames-in-flight 1` shows per dispatch (profiles/). `as_ran` = the same events inside the timed region.
        n_launch = max(1.0, tot["closest_launches"])
        # camera rays that miss the root box are answered by the camera kernel (option "root_cull"), bounce rays that provably leave the scene by the shading kernel's
        # horizon tables (option "horizon_cull"): queries, but no records through the traversal launches
        in_queue = tot["closest_queries"] - tot["root_culled"] - tot["sky_culled"]
        bytes_closest = in_queue * 44.0 + 32.0 * tot["closest_nodes"] + 48.0 * tot["closest_prims"]
        ms_closest = mx_tot["ms1_closest"]   # per frame; ranks run concurrently: the slowest rank's sum
        launch_s = ms_closest * 1e-3 / (n_launch / world) if ms_closest > 0 else float("inf")
        achieved = (bytes_closest / n_launch) / launch_s / 1e9
        # `traffic`: fabric-side bytes per launch from the committed PMC passes of this same workload and this same device code
        # (tools/profile_round.sh: separate --pmc FETCH_SIZE / WRITE_SIZE runs, x1024; reads x the factor calibrated on a known-byte
        # 64-B-node gather, tools/micro/fetch_calib.hip -> profiles/r3_fetch_calibration.txt; Infinity-Cache hits are included). A file
        # measured on other kernel sources is ignored (null) rather than quoted stale.
        traffic = lane_util = tcp_acc = tcp_insts = l2_req = tcc_hit = tcc_miss = None
        pmc_fetch_factor = (None, None)
        pmc_file = pmc_stale = None
        if at_full_size and world == 1:
            import glob
            src = kernel_source_hash()
            tag = "" if args.config == "cfg4" else "_" + args.config
            cands = [f for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*%s_pmc_traffic.json" % tag)), reverse=True)
                     if args.config != "cfg4" or "_cfg" not in os.path.basename(f)]
            pick = None
            for pmc in cands:   # the newest file measured on THIS device code; failing that the newest one, flagged stale
                with open(pmc) as f:
                    j = json.load(f)
                if j.get("source_hash") == src:
                    pick = (pmc, j, False)
                    break
                if pick is None and "tcp_tcc_read_req" in j.get("closest", {}):
                    pick = (pmc, j, True)
            if pick is not None:
                pmc, j, pmc_stale = pick
                pmc_file = os.path.relpath(pmc, ROOT)
                c = j["closest"]
                # gathers at the calibrated x1; the coalesced ray-record reads (32 B per query) are tallied at half: + 16 B per query
                traffic = round((c["hbm_bytes"] + 16.0 * in_queue) / n_launch, 1)
                pmc_fetch_factor = ((j.get("fetch_size_factor") or {}).get("closest"), j.get("fetch_size_factor_source"))
                lane_util = c.get("valu_lane_util"); tcp_acc = c.get("tcp_accesses"); tcp_insts = c.get("vmem_rd_insts")
                l2_req = c.get("tcp_tcc_read_req"); tcc_hit = c.get("tcc_hit"); tcc_miss = c.get("tcc_miss")
        # What binds this kernel, and what is MEASURED about it. The 11 MB BVH is served by LDS copies, the vector L1, the XCDs' L2 and the
        # Infinity Cache: SURVEY 8d's byte model (every node / triangle a ray touches priced as an HBM byte) comes out above the HBM peak and is
        # no bandwidth; the HBM position (`hbm`) is an order of magnitude below. The headline fraction is what the counters say reached the L2:
        # TCP_TCC_READ_REQ (vector L1 -> L2 read requests, 64 B each) per launch / launch duration, against the guide's L2-resident row-gather
        # rate (MI355X_MICROARCH.md "Indexed rows: gather": 16.8-18.8 TB/s chip-wide; Infinity-Cache-resident rows 8.6 TB/s). The algorithmic
        # figure (every gathered byte, wherever it was served from) stays beside it as `algorithmic`; it is a throughput, not a utilisation.
        gather_bytes = 64.0 * tot["closest_nodes"] / 2.0 + 48.0 * tot["closest_prims"] + 48.0 * in_queue
        gather = (gather_bytes / n_launch) / launch_s / 1e9
        as_ran_s = mx_tot["ms_closest"] * 1e-3 * world / n_launch
        alone_s = mx_tot["ms_closest_isolated"] * 1e-3 * world / n_launch
        fetch_factor = fetch_src = None
        if traffic is not None:
            fetch_factor, fetch_src = pmc_fetch_factor
        l2_bytes = None if l2_req is None else 64.0 * l2_req / n_launch
        l2_gbps = None if l2_bytes is None else l2_bytes / launch_s / 1e9
        # lane-level accesses the byte model implies: four 16-byte words per pair node (two nodes visited), three per triangle test, two ray words + one hit word per query
        alg_accesses = 4.0 * tot["closest_nodes"] / 2.0 + 3.0 * tot["closest_prims"] + 3.0 * in_queue
        roofline = {"kernel": "closest-hit BVH traversal, one launch pair per bounce: k_trace_tiles_f32 (camera rays, per-patch sub-trees in LDS) / k_trace_pt_f32<false> (bounces 1..) + k_trace_pairs_f32<false> (small queues)",
                    "bound": "l2_gather",
                    "achieved": None if l2_gbps is None else round(l2_gbps, 1), "peak": GATHER_L2_GBS, "unit": "GB/s",
                    "frac": None if l2_gbps is None else round(l2_gbps / GATHER_L2_GBS, 4),
                    "traffic": traffic,
                    "what": "MEASURED: vector-L1 -> L2 read requests of the closest-hit launches (PMC TCP_TCC_READ_REQ, separate --pmc pass of this workload, x 64 B) per launch / "
                            "average launch duration, against the guide's L2-resident row-gather rate. `traffic` / `hbm`: fabric-side bytes (FETCH_SIZE / WRITE_SIZE passes). "
                            "`algorithmic`: the bytes the rays gather wherever they are served from (LDS copies, vector L1, L2) - a throughput, not a utilisation",
                    "pmc_file": pmc_file, "pmc_stale": pmc_stale,
                    "l2_bytes_per_launch": None if l2_bytes is None else round(l2_bytes, 1),
                    "l2_hit_rate": None if not tcc_hit else round(tcc_hit / (tcc_hit + (tcc_miss or 0.0)), 4),
                    # share of the algorithmic lane-level accesses that never reach the vector L1: served by the LDS copies (treelet, per-patch sub-trees)
                    "lds_served_share": None if tcp_acc is None else round(1.0 - tcp_acc / alg_accesses, 4),
                    "avg_launch_ms": round(launch_s * 1e3, 4), "launches": int(n_launch),
                    "measured": "HIP events on the handle's stream around every closest-hit launch, frames ONE AT A TIME after the timed region (shadow launches beside "
                                "them on the second stream, as in the product): the configuration in which an event pair = rocprofv3's per-dispatch duration "
                                "(profiles/*_kernel_stats_1flight.csv)",
                    # every byte the rays gather (64-B pair-node lines: nodes visited / 2; 48-B triangles tested; 48 B ray + hit record per query; exact device counters of a
                    # counting frame of this workload) per launch / launch duration - last round's headline, now beside the measured one
                    "algorithmic": {"gathered_bytes_per_launch": round(gather_bytes / n_launch, 1), "GBps": round(gather, 1), "over_l2_peak": round(gather / GATHER_L2_GBS, 4),
                                    "over_infinity_cache_rate": round(gather / GATHER_IC_GBS, 4), "lane_accesses_per_launch": round(alg_accesses / n_launch, 1)},
                    # the same events inside the timed region (frames in flight share the chip: an event pair then also holds the wait for compute units)
                    "as_ran": {"frames_in_flight": nfl, "avg_launch_ms": round(as_ran_s * 1e3, 4),
                               "frac": round(l2_bytes / as_ran_s / 1e9 / GATHER_L2_GBS, 4) if (as_ran_s > 0 and l2_bytes is not None) else None},
                    # the same launches with the shadow launches back on the main stream and one frame at a time: the kernel alone on the chip
                    "alone": {"avg_launch_ms": round(alone_s * 1e3, 4), "frac": round(l2_bytes / alone_s / 1e9 / GATHER_L2_GBS, 4) if (alone_s > 0 and l2_bytes is not None) else None},
                    # HBM position: fabric-side bytes per launch from the committed PMC passes (Infinity-Cache hits included: an upper bound of the HBM bytes)
                    "hbm": None if traffic is None else {"bytes_per_launch": traffic, "GBps": round(traffic / launch_s / 1e9, 1), "peak": HBM_PEAK_GBS,
                                                          "frac": round(traffic / launch_s / 1e9 / HBM_PEAK_GBS, 4), "fetch_size_factor": fetch_factor, "fetch_size_factor_source": fetch_src},
                    # SURVEY 8d's algorithmic-byte figure, kept for the record: NOT a bandwidth (cache-resident tree), its ratio to the HBM peak is not a fraction of anything
                    "survey_8d": {"algorithmic_bytes_per_launch": round(bytes_closest / n_launch, 1), "GBps_if_every_byte_came_from_hbm": round(achieved, 1),
                                  "ratio_to_hbm_peak": round(achieved / HBM_PEAK_GBS, 4), "bytes_per_query": round(bytes_closest / max(1.0, tot["closest_queries"]), 1),
                                  "note": "28 B ray + 16 B hit + 32 B per node visited + 48 B per triangle tested; not a bandwidth: the tree never leaves the caches"},
                    "valu_lane_util": lane_util,
                    # The unit this kernel keeps busiest is the CU's vector L1 (TCP): a 16-byte load whose 64 lanes read their own BVH nodes occupies
                    # it for 39 cycles (tools/micro/tcp_gather2.hip, L1 hits), 0.61 cycles per lane-level access; accesses per launch from the
                    # committed PMC pass (TCP_TOTAL_CACHE_ACCESSES). frac = TCP cycles needed / (256 CUs x launch duration x 2.4 GHz), L1 misses not priced.
                    "l1_gather": None if tcp_acc is None else l1_gather_bound(tcp_acc / n_launch, None if tcp_insts is None else tcp_insts / n_launch, launch_s, alone_s),
                    "nodes_per_query": round(tot["closest_nodes"] / max(1.0, tot["closest_queries"]), 2),
                    "tris_per_query": round(tot["closest_prims"] / max(1.0, tot["closest_queries"]), 2)}
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(scene, args.cpu_seconds)
        out = {
            "metric": "Mrays/s (BVH ray queries, closest+any) @ 100k-tri heightfield %d^2 %dspp depth %d%s" % (args.res, args.spp, args.depth, "" if args.config == "cfg4" else " + microfacet BSDFs + area lights (BASELINE cfg5)"),
            "value": round(value, 3), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE %s: procedural heightfield %d triangles%s, %dx%d, %d spp, Path max_depth %d, HaltonSampler, RealisticCamera, box filter; %s BVH; film in interleaved 16-row bands, one gather to rank 0 per frame; %d frame(s) in flight" % (
                args.config, scene.desc.n_prims, "" if args.config == "cfg4" else " (Plastic + Metal, Trowbridge-Reitz), 4 diffuse sphere area lights", W, H, args.spp, args.depth, "reference-exact (Q26/Q27)" if args.compat_bvh else "fixed-bvh", nfl),
                "collective": collective, "comm_world": (comm.world if comm is not None else world),
                "triangles": int(scene.desc.n_prims), "bvh_nodes": int(scene.desc.n_bvh_nodes), "bvh_depth": int(scene.desc.bvh_depth)},
            "roofline": roofline, "cpu_baseline": cpu,
            "camera_mrays_per_s": round(tot["camera_rays"] / (ms_per_step * 1e-3) / 1e6, 3),
            "camera_samples": int(tot["camera_samples"]), "camera_rays": int(tot["camera_rays"]),
            "closest_queries": int(tot["closest_queries"]), "root_culled": int(tot["root_culled"]), "sky_culled": int(tot["sky_culled"]), "any_queries": int(tot["any_queries"]),
            "any_nodes_per_query": round(tot["any_nodes"] / max(1.0, tot["any_queries"]), 2), "any_tris_per_query": round(tot["any_prims"] / max(1.0, tot["any_queries"]), 2),
            "kernel_ms_per_frame": {k: round(mx_tot[k], 3) for k in ("ms_raygen", "ms_closest", "ms_any", "ms_shade", "ms_film", "ms_total")},
            "kernel_ms_per_frame_one_at_a_time": {k.replace("ms1_", "ms_"): round(mx_tot[k], 3) for k in ("ms1_raygen", "ms1_closest", "ms1_any", "ms1_shade", "ms1_film", "ms1_total")},
            # per rank, per frame of the timed region (HIP events on each rank's own stream): the render (ms_total: camera kernel to film kernel) and the frame's
            # collective (ms_gather: around the grouped ncclSend / ncclRecv or the ncclReduce of rrt_film_gather) - max and min over the ranks, so that a scaling
            # run can tell render imbalance between the ranks (max - min of ms_total) from the time the collective takes (0 with one rank / the gloo rehearsal)
            "ranks": {"ms_total": {"max": round(mx_tot["ms_total"], 3), "min": round(mn_tot["ms_total"], 3)},
                      "ms_gather": {"max": round(mx_tot["ms_gather"], 3), "min": round(mn_tot["ms_gather"], 3)}},
            # bounce rays answered by the horizon tables and camera rays answered by the root-box test are queries of the reference (BVHAccel::intersect calls that
            # return false) and count in `value`; the rate of the queries that went through a traversal launch is given beside it
            "traversed_mrays_per_s": round((queries - tot["root_culled"] - tot["sky_culled"]) / (ms_per_step * 1e-3) / 1e6, 3),
            "host_scene_build_s": round(t_build, 3), "handles_create_s": round(t_handles, 3), "horizon_tables_build_s": round(counted["s_horizon_build"], 3),
        }
        print(json.dumps(out))
    if comm is not None:
        comm.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
