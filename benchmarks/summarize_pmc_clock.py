"""Per run directory of gemm_nt2_pmc.sh (one rocprofv3 --kernel-trace --pmc run each): for every kernel the mean
dispatch time, the clock the chip held (GRBM_GUI_ACTIVE / 8 XCDs / time), the MFMA-busy share of the CU-busy
cycles and the shares of the wave cycles spent waiting / issuing."""
import collections, csv, glob, os, sys

root = sys.argv[1]
for run in sorted(glob.glob(os.path.join(root, "*/"))):
    cc = glob.glob(run + "/**/*counter_collection.csv", recursive=True)
    kt = glob.glob(run + "/**/*kernel_trace.csv", recursive=True)
    if not cc:
        continue
    ctr = collections.defaultdict(lambda: collections.defaultdict(float))
    ndisp = collections.defaultdict(set)
    for r in csv.DictReader(open(cc[0])):
        k = r["Kernel_Name"][:60]
        ctr[k][r["Counter_Name"]] += float(r["Counter_Value"])
        ndisp[k].add(r.get("Dispatch_Id"))
    dur = collections.defaultdict(list)
    if kt:
        for r in csv.DictReader(open(kt[0])):
            dur[r["Kernel_Name"][:60]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print(f"== {os.path.basename(run.rstrip('/'))}")
    for k, v in sorted(ctr.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
        if "fill_kernel" in k or "ref_kernel" in k:
            continue
        n = len(ndisp[k])
        d = dur.get(k, [])
        ns = sum(d) / len(d) if d else 0.0
        line = f"  {k:60s} n={n:3d} time {ns / 1e3:8.1f} us"
        if ns and v.get("GRBM_GUI_ACTIVE"):
            line += f"  clock {v['GRBM_GUI_ACTIVE'] / n / 8 / ns:5.2f} GHz"
        if v.get("SQ_BUSY_CU_CYCLES"):
            line += f"  mfma_busy {v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (4 * v['SQ_BUSY_CU_CYCLES']):.3f}"
        wc = v.get("SQ_WAVE_CYCLES", 0)
        if wc:
            line += "  wave: wait_any %.2f wait_inst %.2f active %.2f" % (
                v.get("SQ_WAIT_ANY", 0) / wc, v.get("SQ_WAIT_INST_ANY", 0) / wc, v.get("SQ_ACTIVE_INST_ANY", 0) / wc)
            line += f"  valu_insts/wave_cycle {v.get('SQ_INSTS_VALU', 0) / wc:.3f}"
        print(line)
