"""Summarise a rocprofv3 --pmc ... --kernel-trace run (csv output) per kernel name prefix:
effective clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel time; matrix-pipe busy = SQ_VALU_MFMA_BUSY_CYCLES /
(GRBM_GUI_ACTIVE / 8 * 1024 SIMDs).  Usage: python tools/pmc_summary.py <dir> [kernel-substring] [--raw]"""
import csv, glob, os, sys
from collections import defaultdict

raw = "--raw" in sys.argv                      # also print every counter's total (instruction-mix passes)
argv = [a for a in sys.argv if a != "--raw"]
d = argv[1]
sub = argv[2] if len(argv) > 2 else "conv_h3"
cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
acc = defaultdict(lambda: defaultdict(float))
seen = defaultdict(set)
for r in csv.DictReader(open(cc)):
    name, ns = dur.get(r["Dispatch_Id"], (r["Kernel_Name"], 0))
    if sub not in name:
        continue
    key = name.split("(")[0][:90]
    acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in seen[key]:
        seen[key].add(r["Dispatch_Id"])
        acc[key]["_ns"] += ns
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]["_ns"]):
    t = v["_ns"] * 1e-9
    line = "%-90s launches %5d  time %.3f s" % (k, len(seen[k]), t)
    if "GRBM_GUI_ACTIVE" in v and t > 0:
        cyc = v["GRBM_GUI_ACTIVE"] / 8
        line += "  clock %.3f GHz" % (cyc / t / 1e9)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in v:
            line += "  mfma busy %.1f %%" % (100 * v["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024))
    for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"):
        if c in v and v.get("SQ_WAVE_CYCLES"):
            line += "  %s %.1f %%" % (c, 100 * v[c] / v["SQ_WAVE_CYCLES"])
    print(line)
    if raw:
        print("    " + "  ".join("%s %.4e" % (c, x) for c, x in sorted(v.items()) if c != "_ns"))
