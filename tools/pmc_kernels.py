"""Per-kernel PMC summary (sums over dispatches) from tools/pmc.sh passes: lane utilisation, VALU share, etc."""
import collections, csv, glob, re, sys
root = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(f"{root}/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void rrtd::", "").replace("rrtd::", "")
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0)):
    if c.get("SQ_WAVES", 0) < 1000: continue
    g = lambda n: c.get(n, 0.0)
    lanes = g("SQ_THREAD_CYCLES_VALU") / max(1.0, g("SQ_ACTIVE_INST_VALU")) / 64.0 * 4 if g("SQ_ACTIVE_INST_VALU") else 0
    print(f"{k[:44]:44s} waves {g('SQ_WAVES'):.3g} valu {g('SQ_INSTS_VALU'):.3g} salu {g('SQ_INSTS_SALU'):.3g} vmemrd {g('SQ_INSTS_VMEM_RD'):.3g} vmemwr {g('SQ_INSTS_VMEM_WR'):.3g} lds {g('SQ_INSTS_LDS'):.3g} "
          f"lane_util {g('SQ_THREAD_CYCLES_VALU') / max(1.0, g('SQ_ACTIVE_INST_VALU') * 16):.2f} valu_busy {g('SQ_ACTIVE_INST_VALU') * 4 / max(1.0, g('SQ_BUSY_CYCLES')):.3f} "
          f"wave_cycles/inst {g('SQ_WAVE_CYCLES') * 4 / max(1.0, g('SQ_INSTS_VALU') + g('SQ_INSTS_SALU')):.1f} wait_any {g('SQ_WAIT_ANY') / max(1.0, g('SQ_WAVE_CYCLES')):.2f} "
          f"tcp_acc {g('TCP_TOTAL_CACHE_ACCESSES_sum'):.3g} tcc_req {g('TCP_TCC_READ_REQ_sum'):.3g} tcc_hit {g('TCC_HIT_sum'):.3g} tcc_miss {g('TCC_MISS_sum'):.3g}")
