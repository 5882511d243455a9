"""The 3x3 conv kernels of the step, one shape after the other, for a counter pass (GPU box):

    rocprofv3 --kernel-trace --pmc <counters> --output-format csv -d /tmp/cc -- python3 tools/conv_counters.py run
    python3 tools/conv_counters.py report /tmp/cc <out.txt> [second pass dir]

`run` launches every (shape, tile) of CASES 6 times over 4 rotating operand sets; `report` groups the counter rows by
(kernel, grid size) -- the same instance on two shapes stays two rows -- and prints the SQ fractions the round-5 verdict
asked for.  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles per wave, SQ_VALU_MFMA_BUSY_CYCLES and
SQ_LDS_* count cycles (MI355X_MICROARCH.md, cycle constants)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# (label, B, H, Cin, Cout, tile)
CASES = [
    ("gemm192x128 72^2 128->256 B8", 8, 72, 128, 256, 12),
    ("gemm192x128 36^2 256->512 B8", 8, 36, 256, 512, 12),
    ("gemm192x128 72^2 128->256 B32", 32, 72, 128, 256, 12),
    ("gemm192x128 36^2 256->512 B32", 32, 36, 256, 512, 12),
    ("gemm192x128 18^2 512->1024 B32", 32, 18, 512, 1024, 12),
    ("halo<8,3,4> 36^2 256->512 B8", 8, 36, 256, 512, 16),
    ("halo<8,3,2> 18^2 512->1024 B8", 8, 18, 512, 1024, 18),
    ("halo<8,3,1> 18^2 1024->512 B8", 8, 18, 1024, 512, 19),
    ("halo<8,3,4> 72^2 128->256 B32", 32, 72, 128, 256, 16),
    ("flat 72^2 256->128 B8", 8, 72, 256, 128, 25),
]


def run():
    import torch
    import disyolo_amd  # noqa: F401
    from disyolo_amd import lib as L
    dev = torch.device("cuda:0")
    bf = torch.bfloat16
    only = os.environ.get("CC_ONLY")
    for label, B, H, Cin, Cout, tile in CASES:
        if only and only not in label:
            continue
        NSET = 4
        xs = [torch.randn(B, H, H, Cin, device=dev).to(bf) for _ in range(NSET)]
        ws = [(torch.randn(Cout, 9 * Cin, device=dev) * 0.02).to(bf) for _ in range(NSET)]
        ys = [torch.empty(B, H, H, Cout, dtype=bf, device=dev) for _ in range(NSET)]
        sc = torch.ones(Cout, device=dev)
        sh = torch.zeros(Cout, device=dev)
        ds = [L.make_conv_desc(xs[i], ws[i], ys[i], 3, 1, scale=sc, shift=sh, leaky=True, tile=tile) for i in range(NSET)]
        got = L.conv2d_tile(ds[0])
        for r in range(6):
            for i in range(NSET):
                L.conv2d_fwd(ds[i])
        torch.cuda.synchronize()
        # stand-alone time of the same launches (HIP events on the launch stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for r in range(5):
            for i in range(NSET):
                L.conv2d_fwd(ds[i])
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / (5 * NSET) * 1e3
        fl = 2.0 * B * H * H * Cout * Cin * 9
        print("CASE %-34s runs %s grid-key B=%d: %.1f us = %.0f TFLOP/s" % (label, got, B, us, fl / us / 1e6), flush=True)
        del xs, ws, ys, ds


def report(dirs, out):
    import csv
    import re
    import collections
    import glob
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(set)
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                name = re.sub(r"\(anonymous namespace\)::|void |dyconv::|HIP_vector_type<[^>]*>|\(.*$", "", r["Kernel_Name"])
                if "conv" not in name:
                    continue
                key = (name, int(r["Grid_Size"]) // max(1, int(r.get("Workgroup_Size", 1) or 1)))
                acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
                n[key, r["Counter_Name"]].add((d, r["Dispatch_Id"]))       # (a counter collected in both passes: both count)
    lines = []
    hdr = "%-44s %6s %9s %7s %7s %7s %7s %9s %8s %8s %7s" % (
        "kernel", "blocks", "wavecyc/l", "parked", "stalled", "st_lds", "issuing", "mfma/simd", "lds_act", "lds_cnfl", "coexec")
    lines.append(hdr)
    for key, c in sorted(acc.items(), key=lambda kv: (kv[0][0], kv[0][1])):
        def per(name):
            k = len(n[key, name])
            return c.get(name, 0.0) / k if k else float("nan")
        wc = per("SQ_WAVE_CYCLES")          # quad-cycles summed over waves, per launch
        busy = per("SQ_BUSY_CYCLES")
        lines.append("%-44s %6d %9.0f %6.1f%% %6.1f%% %6.1f%% %6.1f%% %8.1f%% %7.1f%% %7.2f%% %6.1f%%" % (
            key[0][:44], key[1], wc, 100 * per("SQ_WAIT_ANY") / wc, 100 * per("SQ_WAIT_INST_ANY") / wc,
            100 * per("SQ_WAIT_INST_LDS") / wc, 100 * per("SQ_ACTIVE_INST_ANY") / wc,
            100 * per("SQ_VALU_MFMA_BUSY_CYCLES") / (4.0 * wc), 100 * per("SQ_LDS_IDX_ACTIVE") / (4.0 * wc),
            100 * per("SQ_LDS_BANK_CONFLICT") / (4.0 * wc), 100 * per("SQ_VALU_MFMA_COEXEC_CYCLES") / (4.0 * wc)))
        extra = []
        for nm in ("SQ_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_INSTS_LDS", "SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_INSTS_MFMA", "SQ_ACTIVE_INST_LDS",
                   "SQ_INST_CYCLES_VMEM", "GRBM_GUI_ACTIVE", "SQ_WAVES"):
            if len(n[key, nm]):
                extra.append("%s=%.4g" % (nm, per(nm)))
        if extra:
            lines.append("        " + " ".join(extra))
    lines.append("")
    lines.append("parked = SQ_WAIT_ANY, stalled = SQ_WAIT_INST_ANY (st_lds = its SQ_WAIT_INST_LDS part), issuing = SQ_ACTIVE_INST_ANY, each / SQ_WAVE_CYCLES;")
    lines.append("mfma/simd, lds_act, lds_cnfl, coexec = SQ_VALU_MFMA_BUSY_CYCLES, SQ_LDS_IDX_ACTIVE, SQ_LDS_BANK_CONFLICT, SQ_VALU_MFMA_COEXEC_CYCLES")
    lines.append("(cycles) / (4 x SQ_WAVE_CYCLES quad-cycles) = per cycle of ONE wave's lifetime: x waves per SIMD = the SIMD's MFMA pipe, x waves per CU = the CU's LDS array.")
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        report([sys.argv[2]] + sys.argv[4:], sys.argv[3])
