#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (one directory per pass) into profiles/<name>.json:
per kernel (name, grid) the mean counter values per launch; FETCH_SIZE/WRITE_SIZE are in KiB.
HBM traffic per launch follows MI355X_MICROARCH.md: FETCH_SIZE under-reports wide (16 B/lane)
streaming reads by exactly 2x on gfx950, WRITE_SIZE is exact -> traffic = 2*FETCH + WRITE."""
import collections
import csv
import glob
import json
import sys

out, dirs = sys.argv[1], sys.argv[2:]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in dirs:
    for f in glob.glob(f"{d}/*/*counter_collection.csv") + glob.glob(f"{d}/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"].split("(")[0], int(r["Grid_Size"]) // int(r["Workgroup_Size"]))
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = []
for (name, grid), d in agg.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    e = {"kernel": name, "workgroups": grid, "launches": max(len(v) for v in d.values()), **{k: round(v, 1) for k, v in m.items()}}
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        e["hbm_traffic_bytes_per_launch"] = (2.0 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024.0
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m and m["GRBM_GUI_ACTIVE"] > 0:
        e["mfma_pipe_util"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (m["GRBM_GUI_ACTIVE"] / 8.0), 3)
    res.append(e)
res.sort(key=lambda e: -e.get("GRBM_GUI_ACTIVE", e.get("FETCH_SIZE", 0)))
json.dump(res, open(out, "w"), indent=1)
# provenance sidecar: bench.py reports it next to roofline.traffic (the GPU box has no .git: the caller passes GIT_HEAD)
import os

head = os.environ.get("GIT_HEAD", "").strip()
tag = os.path.basename(out).replace("_pmc.json", "").replace(".json", "")
if tag == "pmc":  # gpurun_out/<tag>/pmc.json
    tag = os.path.basename(os.path.dirname(os.path.abspath(out)))
json.dump({"tag": tag, "git_head": head or "unknown"},
          open(out.replace(".json", ".meta.json"), "w"), indent=1)
for e in res[:16]:
    print({k: e[k] for k in ("kernel", "workgroups", "launches", "hbm_traffic_bytes_per_launch", "mfma_pipe_util") if k in e})
