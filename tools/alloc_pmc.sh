#!/bin/bash
# Which counters move with the allocation-to-allocation spread of k_classify?  tools/alloc_probe.py (the same batch created 6 times
# in one process) under rocprofv3 --pmc, one counter group per pass; per dispatch of k_classify: duration and counters.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
pass() {
  name=$1; shift
  rm -rf /tmp/apmc_$name
  REPS=6 STEPS=3 rocprofv3 --pmc "$@" --output-format csv -d /tmp/apmc_$name -- python3 $R/tools/alloc_probe.py > /tmp/apmc_$name.log 2>&1
  python3 - "$name" <<'PY'
import csv, glob, sys, collections
name = sys.argv[1]
rows = collections.OrderedDict()
for f in glob.glob("/tmp/apmc_%s/**/*counter_collection.csv" % name, recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_classify<false, false>" not in r["Kernel_Name"]: continue
        d = rows.setdefault(int(r["Dispatch_Id"]), {"dur": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 if "End_Timestamp" in r else 0.0})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
print("== pass", name)
for k, (did, d) in enumerate(sorted(rows.items())):
    print(k, " ".join("%s=%.4g" % (a, b) for a, b in d.items()))
PY
}
pass a GRBM_GUI_ACTIVE GRBM_UTCL2_BUSY
pass b TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum
pass c TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_BUBBLE_sum TCC_LATENCY_FIFO_FULL_sum
