#!/usr/bin/env python3
"""Join the LDS probe's dispatches (benchmarks/lds_probe.bin under rocprofv3 --pmc) with its pattern names:
what SQ_LDS_BANK_CONFLICT reports for access patterns whose cycle cost is known from the probe's own timing."""
import csv, glob, sys
from collections import defaultdict

d = sys.argv[1]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
by = defaultdict(dict)
for r in rows:
    if "probe<" not in r["Kernel_Name"]:
        continue                                  # the copies of the cycle count back to the host
    by[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
names0 = ["linear", "row_frag<64> (swz64)", "rows128B xor(row&7)", "row_frag<128> (swz128)", "rows256B xor(row&15)",
          "rows128B unswizzled", "rows128B xor(row>>1)", "Stage64<64> commit", "Stage64<128> commit", "8 lanes/row, no swizzle"]
labels = []
for waves in (1, 4):
    for mode in range(3):
        for pat in range(4 if mode == 1 else 10):
            for rep in range(2):
                if mode == 1:
                    lab = f"tr16_b64 hs{128 if pat & 1 else 64} {'natural' if (pat >> 1) & 1 else 'acc-order'}"
                else:
                    lab = ("ds_read_b128  " if mode == 0 else "ds_write_b128 ") + names0[pat]
                labels.append((waves, lab, rep))
ids = sorted(by)
print(f"{'waves':>5}  {'pattern':<44} {'LDS insts':>10} {'conflict cycles':>16} {'per inst':>9}")
for i, did in enumerate(ids[: len(labels)]):
    waves, lab, rep = labels[i]
    if rep == 0:
        continue
    c = by[did]
    inst = c.get("SQ_ACTIVE_INST_LDS", 0.0) or c.get("SQ_INSTS_LDS", 0.0)
    n = c.get("SQ_INSTS_LDS", 0.0)
    conf = c.get("SQ_LDS_BANK_CONFLICT", 0.0)
    print(f"{waves:>5}  {lab:<44} {n:>10.0f} {conf:>16.0f} {conf / n if n else 0:>9.2f}")
