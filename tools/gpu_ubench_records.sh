#!/bin/bash
# records microbenchmark (k_search's fetch pattern, three lane mappings) with and without TCP counters
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R/tools/ubench
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 records.hip -o records 2>/dev/null
./records | tee $O/records.txt
cd /tmp && export TMPDIR=/tmp && rm -rf $O/pmc_records
timeout 200 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d $O/pmc_records -o r -- $R/tools/ubench/records > /dev/null 2>&1
python - <<'PY'
import csv, os, collections
f=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/pmc_records/r_counter_collection.csv"
rows=list(csv.DictReader(open(f)))
by=collections.OrderedDict()
for r in rows:
    k=(r["Dispatch_Id"], r["Kernel_Name"][:40])
    by.setdefault(k, {})[r["Counter_Name"]]=float(r["Counter_Value"]); by[k]["ns"]=float(r["End_Timestamp"])-float(r["Start_Timestamp"])
for (d,k),v in by.items():
    print(d, k, {a: ("%.3g"%b) for a,b in v.items()})
PY
