#!/bin/bash
# SQ counters per kernel for a command (GPU box; counters only, no tracing domains besides the kernel trace).
# usage: tools/pmc_kernels.sh <tag> <python script + args...>
tag=$1; shift
O=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf /tmp/pmc_$tag
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/pmc_$tag -- python3 "$@" > /dev/null 2>&1
f=$(find /tmp/pmc_$tag -name '*counter_collection.csv' | head -1)
python3 - "$f" "$O/${tag}_sq_counters.txt" <<'PY'
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
seen = set()
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::|void |HIP_vector_type<[^>]*>|\(.*$", "", r["Kernel_Name"])
    acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (name, r["Dispatch_Id"])
    if key not in seen:
        seen.add(key); n[name] += 1
out = open(sys.argv[2], "w")
hdr = "%-46s %6s %10s %8s %8s %8s %9s %9s" % ("kernel", "launch", "wave_cyc/l", "parked", "stalled", "issuing", "mfma_busy", "lds_confl")
print(hdr); out.write(hdr + "\n")
for name, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:14]:
    wc = c.get("SQ_WAVE_CYCLES", 0) or 1
    busy = c.get("SQ_BUSY_CYCLES", 0) or 1
    line = "%-46s %6d %10.0f %7.1f%% %7.1f%% %7.1f%% %8.1f%% %8.2f%%" % (
        name[:46], n[name], wc / n[name], 100 * c.get("SQ_WAIT_ANY", 0) / wc, 100 * c.get("SQ_WAIT_INST_ANY", 0) / wc,
        100 * c.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (wc * 4.0),
        100 * c.get("SQ_LDS_BANK_CONFLICT", 0) / (wc * 4.0))
    print(line); out.write(line + "\n")
out.write("\nparked = SQ_WAIT_ANY, stalled = SQ_WAIT_INST_ANY, issuing = SQ_ACTIVE_INST_ANY, each / SQ_WAVE_CYCLES (quad-cycles);\n"
          "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES (cycles) / (4 x SQ_WAVE_CYCLES) = MFMA-pipe cycles per cycle of wave lifetime: with ONE\n"
          "wave per SIMD (the tap-fused weight gradient) it is the SIMD's MFMA utilisation over the whole kernel, prologue and\n"
          "accumulator store included; with w waves per SIMD multiply by w.  lds_confl = SQ_LDS_BANK_CONFLICT per wave cycle.\n")
PY
