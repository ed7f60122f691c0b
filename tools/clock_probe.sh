#!/bin/bash
# Effective shader clock of the residual convolution: SQ_BUSY_CYCLES (summed over the 32 shader engines) / 32 / kernel
# duration from the same rocprofv3 pass.  Usage: [DGNN_CONV_ABLATE=n] tools/clock_probe.sh <tag>
tag=${1:-clk}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/clock_$tag
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out -o p --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python $root/bench.py --steps 5 --warmup 2 --cpu-sample 0 > $out/log 2>&1
python - $out <<'PY'
import csv, glob, sys, collections
rows = collections.defaultdict(dict)
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "conv3x3_board_kernel" not in r["Kernel_Name"]: continue
        d = rows[r["Dispatch_Id"]]
        d[r["Counter_Name"]] = float(r["Counter_Value"])
        d["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        d["grid"] = int(r["Grid_Size"])
full = [d for d in rows.values() if d.get("ns", 0) > 40000]
n = len(full)
ns = sum(d["ns"] for d in full) / n
busy = sum(d["SQ_BUSY_CYCLES"] for d in full) / n / 32
mfma = sum(d["SQ_VALU_MFMA_BUSY_CYCLES"] for d in full) / n / 1024
gui = sum(d["GRBM_GUI_ACTIVE"] for d in full) / n
print(f"{n} launches: {ns/1e3:.1f} us, SQ busy {busy:.0f} cycles -> {busy/ns:.3f} GHz shader clock; MFMA busy {mfma:.0f} cycles/SIMD ({100*mfma/busy:.1f} %); GRBM_GUI_ACTIVE {gui:.0f}")
PY
