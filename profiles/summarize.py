"""Condense the rocprofv3 output of profiles/collect.sh into the committed per-round summaries.

    python profiles/summarize.py r2        # reads gpurun_out/prof_r2/, writes profiles/r2_*.{csv,json}

* <round>_kernel_stats.csv   -- the --stats table of the kernel-trace pass (per kernel: calls, average, share);
* <round>_pmc_counters.csv   -- per kernel and counter, the per-dispatch average of every --pmc pass, with derived
                                rows: HBM bytes per launch (FETCH_SIZE and WRITE_SIZE are reported in KiB; on gfx950
                                FETCH_SIZE counts 64-byte requests as 32, hence x2), GB/s over the traced average
                                duration, and MFMA busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs x kernel cycles);
* <round>_fc1_traffic.json   -- HBM bytes per launch of the dominant kernel (the fc1 GEMM), read by bench.py;
* <round>_forward_traffic.json -- HBM bytes of a whole forward (all kernels), read by bench.py.
"""
from __future__ import annotations

import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIMDS = 256 * 4


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    name = re.sub(r"\(.*\)$", "", name)
    return name.strip()


def main(rnd: str) -> None:
    src = os.path.join(ROOT, "gpurun_out", f"prof_{rnd}")
    out = os.path.join(ROOT, "profiles")
    # ---- kernel stats
    stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
    avg_ns = {}
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        with open(os.path.join(out, f"{rnd}_kernel_stats.csv"), "w", newline="") as f:
            f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline (MI355X)\n")
            w = csv.writer(f)
            w.writerow(["kernel", "calls", "total_ns", "avg_ns", "percent", "min_ns", "max_ns"])
            for r in rows:
                k = short(r["Name"])
                avg_ns[k] = float(r["AverageNs"])
                w.writerow([k, r["Calls"], r["TotalDurationNs"], f"{float(r['AverageNs']):.0f}", r["Percentage"],
                            r["MinNs"], r["MaxNs"]])
    # ---- kernel stats of the training step (bench.py --mode train), when collected
    for tag, cmd in (("train_bf16", "--mode train --dtype bf16 --batch 256"), ("train_f32", "--mode train --dtype f32 --batch 64")):
        tstats = glob.glob(os.path.join(src, tag, "**", "*kernel_stats.csv"), recursive=True)
        if not tstats:
            continue
        with open(os.path.join(out, f"{rnd}_{tag}_kernel_stats.csv"), "w", newline="") as f:
            f.write(f"# rocprofv3 --kernel-trace --stats -- python3 bench.py {cmd} --steps 3 --warmup 1 (MI355X, ViT-B/16)\n")
            w = csv.writer(f)
            w.writerow(["kernel", "calls", "total_ns", "avg_ns", "percent", "min_ns", "max_ns"])
            for r in csv.DictReader(open(tstats[0])):
                w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], f"{float(r['AverageNs']):.0f}", r["Percentage"],
                            r["MinNs"], r["MaxNs"]])
    # ---- counters
    agg = collections.defaultdict(list)
    for path in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            agg[(short(r["Kernel_Name"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
    per = collections.defaultdict(dict)
    for (k, c), v in agg.items():
        per[k][c] = (sum(v) / len(v), len(v))
    with open(os.path.join(out, f"{rnd}_pmc_counters.csv"), "w", newline="") as f:
        f.write("# rocprofv3 --kernel-trace --pmc <counters> -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline; one\n"
                "# pass per counter group; per-dispatch averages; derived rows are marked '='\n")
        w = csv.writer(f)
        w.writerow(["kernel", "counter", "avg_per_dispatch", "dispatches"])
        for k in sorted(per):
            c = per[k]
            for name in sorted(c):
                w.writerow([k, name, f"{c[name][0]:.1f}", c[name][1]])
            if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                hbm = c["FETCH_SIZE"][0] * 1024 * 2 + c["WRITE_SIZE"][0] * 1024
                w.writerow([k, "=HBM_BYTES (2*FETCH_SIZE KiB + WRITE_SIZE KiB)", f"{hbm:.0f}", ""])
                if k in avg_ns:
                    w.writerow([k, "=HBM_GBps (over traced average duration)", f"{hbm / avg_ns[k]:.1f}", ""])
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c and c["GRBM_GUI_ACTIVE"][0] > 0:
                cycles = c["GRBM_GUI_ACTIVE"][0] / 8.0     # summed over the 8 XCDs
                w.writerow([k, "=MFMA_BUSY_FRACTION (busy cycles / (1024 SIMDs x kernel cycles))",
                            f"{c['SQ_VALU_MFMA_BUSY_CYCLES'][0] / (SIMDS * cycles):.4f}", ""])
    # ---- fc1 traffic for bench.py
    # fc1 = the LN + GELU epilogue (3): one launch that also runs the leftover-tile slices (gemm_pp_shared_kernel), or,
    # in trees before that, a whole-tile launch (gemm_pp_kernel<3, 8, false..>) plus a slice launch (<3, 2, true..>)
    fc1 = ([k for k in per if k.startswith("gemm_pp_shared_kernel<3,")]
           or [k for k in per if k.startswith("gemm_pp_kernel<3, 8, false")])
    if fc1 and "FETCH_SIZE" in per[fc1[0]] and "WRITE_SIZE" in per[fc1[0]]:
        c = per[fc1[0]]
        slice_k = [k for k in per if k.startswith("gemm_pp_kernel<3, 2, true")]
        hbm = c["FETCH_SIZE"][0] * 2048 + c["WRITE_SIZE"][0] * 1024
        extra = 0.0
        if slice_k and "FETCH_SIZE" in per[slice_k[0]]:
            extra = per[slice_k[0]]["FETCH_SIZE"][0] * 2048 + per[slice_k[0]]["WRITE_SIZE"][0] * 1024
        json.dump({"config": "vit_b16", "batch_per_gpu": 256, "kernel": fc1[0],
                   "fetch_size_kib": c["FETCH_SIZE"][0], "write_size_kib": c["WRITE_SIZE"][0],
                   "slice_launch_bytes": extra, "hbm_bytes_per_launch": hbm + extra,
                   "note": "FETCH_SIZE x2 (gfx950 counts 64-byte requests as 32 bytes); separate --pmc passes; one "
                           "fc1 GEMM, leftover-tile slices included"},
                  open(os.path.join(out, f"{rnd}_fc1_traffic.json"), "w"), indent=1)
    # ---- HBM bytes of a whole forward: sum over kernels of (bytes per dispatch) x (dispatches per forward); the
    # dispatch counts come from the kernel-trace pass, one embed_finalize launch = one forward
    if stats:
        calls = {short(r["Name"]): int(r["Calls"]) for r in rows}
        fwd = max([v for k, v in calls.items() if k.startswith("embed_finalize_kernel<unsigned short")] or [0])
        if fwd:
            total, parts = 0.0, {}
            for k, c in per.items():
                if "FETCH_SIZE" in c and "WRITE_SIZE" in c and k in calls and calls[k] >= fwd:
                    b = (c["FETCH_SIZE"][0] * 2048 + c["WRITE_SIZE"][0] * 1024) * (calls[k] / fwd)
                    total += b
                    parts[k] = round(b)
            json.dump({"config": "vit_b16", "batch_per_gpu": 256, "hbm_bytes_per_forward": round(total),
                       "hbm_bytes_per_image": round(total / 256), "per_kernel_bytes_per_forward": parts,
                       "note": "sum over kernels launched at least once per forward of (2 x FETCH_SIZE KiB + WRITE_SIZE KiB) "
                               "per dispatch x dispatches per forward; FETCH_SIZE doubled per the gfx950 correction; "
                               "FETCH_SIZE counts what an XCD's L2 misses, whether HBM or the Infinity Cache serves it"},
                      open(os.path.join(out, f"{rnd}_forward_traffic.json"), "w"), indent=1)
    print("wrote", sorted(os.listdir(out)))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r1")
