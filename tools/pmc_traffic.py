"""Per-kernel FETCH_SIZE / WRITE_SIZE (KiB counters) from two rocprofv3 --pmc passes (csv), GB per launch.
gfx950: FETCH_SIZE counts wide coalesced reads at half their bytes (MI355X guide) -- corrected value in brackets.
Usage: python tools/pmc_traffic.py <fetch_dir> <write_dir> [--json profiles/traffic.json --bench-json <bench line of the profiled run>]

With --json the per-kernel figures are also written as JSON, stamped with the hash of the kernel sources they were
measured on (jax_nbody_emulator_with_dj_amd._lib.source_hash): bench.py reports `roofline.traffic` from that file only
while the hash still matches the sources it runs, and null otherwise -- a number measured on other kernels never
survives a kernel change."""
import csv, glob, json, os, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def load(d, counter):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    tot, n = defaultdict(float), defaultdict(set)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"][:64]
        tot[k] += float(r["Counter_Value"]) * 1024.0
        n[k].add(r["Dispatch_Id"])
    return tot, n


args = sys.argv[1:]
out_json = workload = None
if "--json" in args:
    i = args.index("--json"); out_json = args[i + 1]; del args[i:i + 2]
if "--bench-json" in args:
    i = args.index("--bench-json")
    workload = json.loads([l for l in open(args[i + 1]) if l.startswith("{")][-1])["config"]["traffic_key"]
    del args[i:i + 2]
ft, fn = load(args[0], "FETCH_SIZE")
wt, wn = load(args[1], "WRITE_SIZE")
kern = {}
for k in sorted(ft, key=lambda k: -(ft[k] + wt.get(k, 0))):
    L = len(fn[k])
    w = wt.get(k, 0) / max(len(wn.get(k, [1])), 1)
    print("%-64s launches=%4d  FETCH_SIZE=%.3f GB/launch (x2 gfx950 correction: %.3f)  WRITE_SIZE=%.3f GB/launch"
          % (k, L, ft[k] / L / 1e9, 2 * ft[k] / L / 1e9, w / 1e9))
    kern[k] = {"launches": L, "fetch_bytes_corrected": 2 * ft[k] / L, "write_bytes": w, "traffic_bytes": 2 * ft[k] / L + w}
if out_json:
    from jax_nbody_emulator_with_dj_amd import _lib
    json.dump({"build": _lib.source_hash(), "workload": workload, "unit": "bytes per launch, average over the kernel's launches; "
               "FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, separate rocprofv3 --pmc passes", "kernels": kern},
              open(out_json, "w"), indent=1)
    print("wrote", out_json)
