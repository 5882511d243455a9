"""which hardware queue each of the step's lanes landed on: python tools/queue_map.py <kernel_trace.csv>
(rocprofv3 --kernel-trace of a bench run; kernels are attributed to lanes by family)"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
byq = collections.defaultdict(collections.Counter)
for r in rows:
    n = r["Kernel_Name"]
    fam = ("wgrad/slab" if ("wgrad" in n or "slab_reduce" in n) else "adam/pack" if ("adam" in n or "pack_all" in n) else
           "nms/rois/mask-loss" if any(t in n for t in ("nms_", "decode_score", "mask_rois", "psroi_loss", "shuffle_perm")) else
           "bn" if ("bn_" in n or "colreduce" in n) else "conv" if "conv" in n or "block" in n else
           "rccl" if ("ccl" in n.lower() or "AllReduce" in n) else "other")
    byq[r["Queue_Id"]][fam] += 1
for q, c in sorted(byq.items()):
    print("queue %s: %s" % (q, dict(c)))
