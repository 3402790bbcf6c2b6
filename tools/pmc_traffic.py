"""HBM traffic per kernel family from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (one directory per pass).

rocprofv3 reports both counters in KiB-like units of 1024 B (derived from TCC_EA0_RDREQ / WRREQ). Per MI355X_MICROARCH.md (HBM section)
FETCH_SIZE on gfx950 tallies the 128-B requests of a wide coalesced streaming read at 64 B (x2), and "other access widths are uncalibrated":
tools/micro/fetch_calib.hip (profiles/r3_fetch_calibration.txt) calibrates the traversal kernels' shape - every lane reading the four 16-byte
words of its own 64-byte node, each node once out of 2 GiB: FETCH_SIZE x 1024 = 1.003 x the bytes (64-B requests, tallied in full; Infinity-Cache
hits are counted: an 8 MiB table read 16 times over reports its L2 misses). So the traversal families get x1 on their gathers; their coalesced
part - the 32 B of ray record per query - is tallied at half, which the caller adds from its query count (bench.py). The streaming families
(raygen, shading, film: coalesced 16-byte-per-lane records) keep x2.
The profiled command is `bench.py --steps 1 --warmup 0 --frames-in-flight 1`; its first frame is the counting frame (generic
counting kernels), the product kernels appear in the others (their number is derived from the dispatch count).
"""
import collections, csv, glob, hashlib, json, os, re, sys
root = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def kernel_source_hash():   # = bench.py kernel_source_hash(): ties the file to the device code it was measured on
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "rs_ray_toy_amd", "csrc", "device", "*"))):
        if f.endswith((".hpp", ".hip")):
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]
fam = collections.OrderedDict([
    ("closest", r"k_trace_(pt|pairs)_f32<false[,>]|k_trace_tiles_f32"), ("any", r"k_trace_(pt|pairs)_f32<true[,>]|k_shadow_lists_f32"),
    ("raygen", r"k_raygen|k_pixel_offsets"), ("shade", r"k_shade"), ("film", r"k_film|k_accumulate")])
tot = {k: collections.defaultdict(float) for k in fam}
disp = {k: collections.defaultdict(int) for k in fam}
for f in glob.glob(f"{root}/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for k, pat in fam.items():
            if re.search(pat, r["Kernel_Name"]):
                tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k][r["Counter_Name"]] += 1
# product frames in the profiled run = closest-hit dispatches / 16 (8 bounces x the two regime kernels; the counting frame uses the generic kernels)
# (argv[2]: closest-hit dispatches per frame of other workloads - BASELINE cfg5 renders 16 passes x 16 bounces x 2 = 512)
PER_FRAME = int(sys.argv[2]) if len(sys.argv) > 2 else 16
FRAMES = max(1, round(disp["closest"]["FETCH_SIZE"] / PER_FRAME))
out = {"units": "bytes per frame; FETCH_SIZE x1024 x factor (traversal gathers x1, streaming families x2: profiles/r3_fetch_calibration.txt), WRITE_SIZE x1024",
       "fetch_size_factor": {"closest": 1.0, "any": 1.0, "raygen": 2.0, "shade": 2.0, "film": 2.0},
       "fetch_size_factor_source": "profiles/r3_fetch_calibration.txt (tools/micro/fetch_calib.hip): 64-B-node gather 1.003, coalesced 16 B per lane 0.500 of the known bytes",
       "coalesced_record_note": "closest / any: the coalesced ray-record reads (32 B per query) are tallied at half by FETCH_SIZE; + 16 B x queries restores them (bench.py does)",
       "frames_profiled": FRAMES, "source_hash": kernel_source_hash()}
for k in fam:
    nf = FRAMES if k in ("closest", "any") else FRAMES + 1     # raygen / shading / film kernels also run in the counting frame
    raw = tot[k]["FETCH_SIZE"] * 1024 / nf
    rd = raw * out["fetch_size_factor"][k]
    wr = tot[k]["WRITE_SIZE"] * 1024 / nf
    out[k] = {"fetch_size_raw_bytes": raw, "read_bytes": rd, "write_bytes": wr, "hbm_bytes": rd + wr, "dispatches_per_frame": disp[k]["FETCH_SIZE"] / nf}
# lane-level accesses of the vector L1 (TCP): what the traversal kernels queue for (tools/micro/tcp_gather2.hip: 0.61 cycles of the CU's TCP each)
for k in fam:
    acc = tot[k].get("TCP_TOTAL_CACHE_ACCESSES_sum", 0.0)
    if acc > 0:
        nf = FRAMES if k in ("closest", "any") else FRAMES + 1
        out[k]["tcp_accesses"] = acc / nf
        # vector L1 -> L2 read requests (64 B each) and the L2's hits / misses: what bench.py's measured `frac` is made of
        for name, key in (("TCP_TCC_READ_REQ_sum", "tcp_tcc_read_req"), ("TCC_HIT_sum", "tcc_hit"), ("TCC_MISS_sum", "tcc_miss")):
            if tot[k].get(name, 0.0) > 0: out[k][key] = tot[k][name] / nf
        if tot[k].get("SQ_INSTS_VMEM_RD", 0.0) > 0:   # wave-level vector loads: accesses / (64 x this) = how full the loading waves are
            out[k]["vmem_rd_insts"] = tot[k]["SQ_INSTS_VMEM_RD"] / nf
# VALU lane utilisation of the traversal kernels where the SQ pass was collected too (tools/pmc.sh pass 1)
for k in ("closest", "any"):
    tc, ai = tot[k].get("SQ_THREAD_CYCLES_VALU", 0.0), tot[k].get("SQ_ACTIVE_INST_VALU", 0.0)
    if ai > 0:
        out[k]["valu_lane_util"] = round(tc / (ai * 16.0) / 4.0, 4)   # thread-cycles per active VALU cycle, of 64 lanes (16 lanes x 4 cycles)
print(json.dumps(out, indent=1))
