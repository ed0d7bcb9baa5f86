#!/usr/bin/env python3
"""Average the rocprofv3 --pmc counter_collection CSVs per kernel (skipping warm-up dispatches).
  pmc_summary.py <dir>                              print + write <dir>/summary.txt
  pmc_summary.py <dir> --traffic-json KEY SOURCE    also record the HBM bytes per launch of the two apply kernels under
                                                    KEY ("bp3_p4_n54") in bench_traffic.json at the repository root
Bytes (MI355X_MICROARCH.md, HBM section, and profiles/r2/pmc_calibration.txt for this library's own access widths: the
4-B and 8-B gathers are 128-B requests too): read = 2 * FETCH_SIZE KiB (= TCC_EA0_RDREQ * 128 B), written = WRITE_SIZE KiB."""
import csv, sys, collections, json
from pathlib import Path
root = Path(sys.argv[1])
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in root.rglob("*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][-90:]
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = []
avg = {}
for k, d in acc.items():
    if "at::native" in k:
        continue
    out.append(f"== {k}")
    for c, v in sorted(d.items()):
        v = v[3:] if len(v) > 6 else v
        avg.setdefault(k, {})[c] = sum(v)/len(v)
        out.append(f"   {c:32s} n={len(v):3d} avg={sum(v)/len(v):.4g}")
txt = "\n".join(out)
print(txt)
(root / "summary.txt").write_text(txt + "\n")

if "--traffic-json" in sys.argv:
    i = sys.argv.index("--traffic-json")
    key, source = sys.argv[i + 1], sys.argv[i + 2]
    def kernel_bytes(match):
        for k, d in avg.items():
            if match(k) and "FETCH_SIZE" in d and "WRITE_SIZE" in d:
                return {"kernel": k.strip(), "read_bytes": 2*1024*d["FETCH_SIZE"], "written_bytes": 1024*d["WRITE_SIZE"],
                        "FETCH_SIZE_KiB": d["FETCH_SIZE"], "WRITE_SIZE_KiB": d["WRITE_SIZE"],
                        "TCC_EA0_RDREQ": d.get("TCC_EA0_RDREQ_sum"), "TCC_EA0_WRREQ": d.get("TCC_EA0_WRREQ_sum")}
        return None
    l2e = kernel_bytes(lambda k: "k_pa_apply<" in k or "apply<" in k)
    e2l = kernel_bytes(lambda k: "k_scatter<0, false>" in k)
    if not l2e:
        sys.exit("no k_pa_apply dispatches with FETCH_SIZE and WRITE_SIZE found")
    f = Path(__file__).resolve().parent.parent / "bench_traffic.json"
    db = json.loads(f.read_text()) if f.exists() else {}
    db[key] = {"bytes_per_launch": l2e["read_bytes"] + l2e["written_bytes"], "source": source, "l2e": l2e, "e2l": e2l,
               "e2l_bytes_per_launch": (e2l["read_bytes"] + e2l["written_bytes"]) if e2l else None,
               "correction": "read = 2 x FETCH_SIZE (128-B requests tallied at 64 B on gfx950), written = WRITE_SIZE"}
    f.write_text(json.dumps(db, indent=1) + "\n")
    print(f"bench_traffic.json[{key}] = {db[key]['bytes_per_launch']:.4g} B per k_pa_apply launch")
