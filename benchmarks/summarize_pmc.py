"""Summarise a rocprofv3 --pmc counter_collection CSV: per kernel name, the counters summed over
all dispatches, plus MFMA utilisation (MFMA busy cycles / CU busy cycles / 4 SIMDs) where both
counters were collected."""
import collections, csv, glob, sys

d = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
seen = set()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:70]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (k, r.get("Dispatch_Id"))
    if key not in seen:
        seen.add(key)
        calls[k] += 1
names = sorted({c for v in agg.values() for c in v})
order = sorted(agg, key=lambda k: -agg[k].get("SQ_BUSY_CU_CYCLES", agg[k].get("SQ_WAVE_CYCLES", 0.0)))[:top]
print("counters:", ", ".join(names))
for k in order:
    v = agg[k]
    extra = ""
    if v.get("SQ_BUSY_CU_CYCLES") and "SQ_VALU_MFMA_BUSY_CYCLES" in v:
        extra = f"  mfma_util={v['SQ_VALU_MFMA_BUSY_CYCLES'] / (4.0 * v['SQ_BUSY_CU_CYCLES']):.3f}"
    if v.get("SQ_ACTIVE_INST_LDS") and "SQ_LDS_BANK_CONFLICT" in v:
        extra += f"  lds_conflict/active={v['SQ_LDS_BANK_CONFLICT'] / v['SQ_ACTIVE_INST_LDS']:.2f}"
    print(f"{k:70s} calls {calls[k]:5d}" + extra)
    print("      " + "  ".join(f"{c}={v[c]:.4g}" for c in names if c in v))
