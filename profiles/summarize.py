"""Condense the rocprofv3 output of profiles/collect.sh into the committed per-round summaries.

    python profiles/summarize.py r3                      # reads gpurun_out/prof_r3/, writes profiles/r3_*.{csv,json}
    python profiles/summarize.py r3 vit_b16_384 64       # reads gpurun_out/prof_r3_vit_b16_384/, writes profiles/r3_vit_b16_384_*

The JSON files carry the commit and a digest of the kernel sources they were collected on; bench.py reports their
figures only while that digest matches the tree it runs from.

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


def csrc_digest() -> str:
    """sha256 over the kernel sources (file names and contents, sorted).  Same function in bench.py."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "vision-transformer_amd", "csrc")
    for path in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h")) + glob.glob(os.path.join(d, "*.cpp"))):
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def provenance() -> dict:
    """Which tree the counters describe.  Summarise right after collecting, before editing any kernel source."""
    import subprocess
    try:
        commit = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
        dirty = bool(subprocess.check_output(["git", "-C", ROOT, "status", "--porcelain", "--", "vision-transformer_amd/csrc"], text=True).strip())
    except (OSError, subprocess.CalledProcessError):
        commit, dirty = None, None
    return {"commit": commit, "csrc_dirty_vs_commit": dirty, "csrc_sha256": csrc_digest()}


def newest(pattern: str):
    """The most recently written file matching ``pattern`` (a pass directory keeps the files of earlier collections: every
    gpurun call merges its output into the local tree, named by the profiled process id), or None."""
    files = glob.glob(pattern, recursive=True)
    return max(files, key=os.path.getmtime) if files else None


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    name = re.sub(r"\(.*\)$", "", name)
    return name.strip()


def main(rnd: str, config: str = "vit_b16", batch: int = 256) -> None:
    src = os.path.join(ROOT, "gpurun_out", f"prof_{rnd}" if config == "vit_b16" else f"prof_{rnd}_{config}")
    out = os.path.join(ROOT, "profiles")
    wl = "" if config == "vit_b16" else f"--config {config} --batch {batch} "
    rnd_in = rnd
    if config != "vit_b16":
        rnd = f"{rnd}_{config}"
    prov = provenance()
    # ---- kernel stats
    stats = newest(os.path.join(src, "trace", "**", "*kernel_stats.csv"))
    avg_ns = {}
    if stats:
        rows = list(csv.DictReader(open(stats)))
        with open(os.path.join(out, f"{rnd}_kernel_stats.csv"), "w", newline="") as f:
            f.write(f"# rocprofv3 --kernel-trace --stats -- python3 bench.py {wl}--steps 10 --warmup 3 --no-cpu-baseline (MI355X); "
                    f"commit {prov['commit']}, csrc {prov['csrc_sha256']}\n")
            w = csv.writer(f)
            w.writerow(["kernel", "calls", "total_ns", "avg_ns", "percent", "min_ns", "max_ns"])
            for r in rows:
                k = short(r["Name"])
                avg_ns[k] = float(r["AverageNs"])
                w.writerow([k, r["Calls"], r["TotalDurationNs"], f"{float(r['AverageNs']):.0f}", r["Percentage"],
                            r["MinNs"], r["MaxNs"]])
    # ---- kernel stats of the training step (bench.py --mode train), when collected
    for tag, cmd in (("train_bf16", "--mode train --dtype bf16 --batch 256"), ("train_f32", "--mode train --dtype f32 --batch 64")):
        tstats = newest(os.path.join(src, tag, "**", "*kernel_stats.csv"))
        if not tstats:
            continue
        with open(os.path.join(out, f"{rnd}_{tag}_kernel_stats.csv"), "w", newline="") as f:
            f.write(f"# rocprofv3 --kernel-trace --stats -- python3 bench.py {cmd} --steps 3 --warmup 1 (MI355X, ViT-B/16)\n")
            w = csv.writer(f)
            w.writerow(["kernel", "calls", "total_ns", "avg_ns", "percent", "min_ns", "max_ns"])
            for r in csv.DictReader(open(tstats)):
                w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], f"{float(r['AverageNs']):.0f}", r["Percentage"],
                            r["MinNs"], r["MaxNs"]])
    # ---- counters
    agg = collections.defaultdict(list)
    for passdir in sorted(glob.glob(os.path.join(src, "pmc_*"))):
        path = newest(os.path.join(passdir, "**", "*counter_collection.csv"))
        if path is None:
            continue
        for r in csv.DictReader(open(path)):
            agg[(short(r["Kernel_Name"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
    per = collections.defaultdict(dict)
    for (k, c), v in agg.items():
        per[k][c] = (sum(v) / len(v), len(v))
    with open(os.path.join(out, f"{rnd}_pmc_counters.csv"), "w", newline="") as f:
        f.write(f"# rocprofv3 --kernel-trace --pmc <counters> -- python3 bench.py {wl}--steps 2 --warmup 1 --no-cpu-baseline; one\n"
                f"# pass per counter group; per-dispatch averages; derived rows are marked '='; commit {prov['commit']}, csrc {prov['csrc_sha256']}\n")
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
            if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"][0] > 0:
                w.writerow([k, "=WAIT_ANY_FRACTION (SQ_WAIT_ANY / SQ_WAVE_CYCLES)",
                            f"{c['SQ_WAIT_ANY'][0] / c['SQ_WAVE_CYCLES'][0]:.4f}", ""])
    # ---- fc1 traffic for bench.py
    # fc1 = the LN + GELU epilogue (3): one launch that also runs the leftover-tile slices (gemm_pp_shared_kernel), or,
    # in trees before that, a whole-tile launch (gemm_pp_kernel<3, 8, false..>) plus a slice launch (<3, 2, true..>)
    fc1 = ([k for k in per if k.startswith("gemm_pp_shared_kernel<3,")]
           or [k for k in per if k.startswith("gemm_pp_kernel<3,")])
    if fc1 and "FETCH_SIZE" in per[fc1[0]] and "WRITE_SIZE" in per[fc1[0]]:
        c = per[fc1[0]]
        slice_k = [k for k in per if k.startswith("gemm_pp_kernel<3, 2, true")]
        hbm = c["FETCH_SIZE"][0] * 2048 + c["WRITE_SIZE"][0] * 1024
        extra = 0.0
        if slice_k and "FETCH_SIZE" in per[slice_k[0]]:
            extra = per[slice_k[0]]["FETCH_SIZE"][0] * 2048 + per[slice_k[0]]["WRITE_SIZE"][0] * 1024
        json.dump({"config": config, "batch_per_gpu": batch, **prov, "kernel": fc1[0],
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
            json.dump({"config": config, "batch_per_gpu": batch, **prov, "hbm_bytes_per_forward": round(total),
                       "hbm_bytes_per_image": round(total / batch), "per_kernel_bytes_per_forward": parts,
                       "note": "sum over kernels launched at least once per forward of (2 x FETCH_SIZE KiB + WRITE_SIZE KiB) "
                               "per dispatch x dispatches per forward; FETCH_SIZE doubled per the gfx950 correction; "
                               "FETCH_SIZE counts what an XCD's L2 misses, whether HBM or the Infinity Cache serves it"},
                      open(os.path.join(out, f"{rnd}_forward_traffic.json"), "w"), indent=1)
    print("wrote", sorted(os.listdir(out)))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r3", sys.argv[2] if len(sys.argv) > 2 else "vit_b16",
         int(sys.argv[3]) if len(sys.argv) > 3 else 256)
