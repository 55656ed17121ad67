#!/usr/bin/env python3
"""rocprofv3 --pmc passes (tools/prof_pmc.sh <tag> <n_vcf>) -> profiles/<round>_pmc_per_launch.json and
profiles/traffic.json (HBM bytes per k_classify launch, read by bench.py for roofline.traffic).
usage: python3 tools/pmc_to_profiles.py gpurun_out/pmc_<tag> <n_vcf> [round-prefix]"""
import collections
import csv
import glob
import json
import os
import sys

src, nv = sys.argv[1], int(sys.argv[2])
prefix = sys.argv[3] if len(sys.argv) > 3 else "r01"
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        if not any(x in k for x in ("k_classify", "k_compact", "k_finalize")):
            continue
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[(k, row["Counter_Name"])] += 1
out = {k: {c: v / cnt[(k, c)] for c, v in sorted(d.items())} for k, d in sorted(agg.items())}
suffix = sys.argv[4] if len(sys.argv) > 4 else ""          # "alleles": the allele-extended instantiation's passes (RARGS="0 30"): its own file, traffic.json untouched
for k, d in out.items():                                     # per round of 256 records: what the instruction counts mean
    if "k_classify" in k and "SQ_INSTS_VALU" in d:
        rounds = nv * 1000000 / 256.0
        d["derived"] = {"valu_per_round": d["SQ_INSTS_VALU"] / rounds, "salu_per_round": d.get("SQ_INSTS_SALU", 0) / rounds, "lds_per_round": d.get("SQ_INSTS_LDS", 0) / rounds,
                        "valu_busy_of_kernel_cycles": d["SQ_INSTS_VALU"] * 4 / 1024 / (d["GRBM_GUI_ACTIVE"] / 8) if d.get("GRBM_GUI_ACTIVE") else None,
                        "hbm_bytes": (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0 if "FETCH_SIZE" in d and "WRITE_SIZE" in d else None,
                        "note": "wave-instructions per round of 256 records; VALU busy = instructions x 4 cycles / 1 024 SIMDs over GRBM_GUI_ACTIVE / 8 XCDs"}
json.dump(out, open(os.path.join(root, "%s_pmc_per_launch%s.json" % (prefix, "_" + suffix if suffix else "")), "w"), indent=1)
if suffix:
    print(json.dumps({k: v.get("derived") for k, v in out.items() if "k_classify" in k}))
    sys.exit(0)
kc = next(v for k, v in out.items() if "k_classify<false, false>" in k or k.endswith("k_classify<false>"))
fetch_kb, write_kb = kc["FETCH_SIZE"], kc["WRITE_SIZE"]
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from quasimodo_amd._lib import kernel_source_id
traffic = {"vcfs": nv, "records": 1000000, "hbm_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0,
           "kernels_build": kernel_source_id(),
           "fetch_size_kb": fetch_kb, "write_size_kb": write_kb,
           "note": "k_classify<false,false>, rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes; FETCH_SIZE doubled per "
                   "MI355X_MICROARCH.md (gfx950 counts 128-B requests at 64 B for wide coalesced reads); KB -> bytes"}
json.dump(traffic, open(os.path.join(root, "traffic.json"), "w"), indent=1)
print(json.dumps(traffic))
