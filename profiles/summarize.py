#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (gpurun_out/prof/...) into the small summaries kept under
profiles/.  Kernel names are shortened (torch's are kilobytes long).

    python profiles/summarize.py stats <kernel_stats.csv> <out.csv>
    python profiles/summarize.py pmc   <fetch_counter_collection.csv> <write_counter_collection.csv> \
                                       <kernel-substring> <rows_per_launch> <row_read_bytes> <out.json> [commit]
    python profiles/summarize.py counters <counter_collection.csv> <kernel-substring> <out.txt> [more csv ...]
(profiles/collect.sh and profiles/collect_batch.sh run them on the GPU box.)

PMC correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read (16 B/lane), so
read bytes = FETCH_SIZE * 1024 * 2; WRITE_SIZE * 1024 is taken as is.
"""
import csv
import json
import sys


def short(name: str, n: int = 110) -> str:
    name = name.replace("(anonymous namespace)::", "")
    return name if len(name) <= n else name[:n] + "..."


def kernel_source_hash(root=None):
    import os as _os
    root = root or _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
    """Hash of the part of csrc/u8.hip that defines u8_scan_kernel (device helpers + the kernel):
    edits elsewhere in the file do not void a PMC profile of that kernel."""
    import hashlib
    import os
    src = open(os.path.join(root, "quantization_amd", "csrc", "u8.hip"), "rb").read()
    a = src.find(b"// ------------------------------------------------------------------------------ device helpers")
    b = src.find(b"// NQ (2, 4, 8) queries per row read on the vector ALU")
    region = src[a:b] if 0 <= a < b else src
    return hashlib.sha256(region).hexdigest()[:16]


def counters(paths, needle, dst):
    """Mean per launch of every counter collected for kernels whose name contains `needle`."""
    acc, durs = {}, []
    for path in paths:
        seen = set()
        for r in csv.DictReader(open(path)):
            if needle not in r["Kernel_Name"]:
                continue
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            key = r.get("Dispatch_Id", r["Start_Timestamp"])
            if key not in seen:
                seen.add(key)
                durs.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    lines = [f"kernel name contains: {needle}",
             f"launches sampled: {len(durs)}   mean duration under the counter passes: {sum(durs) / max(1, len(durs)) / 1e6:.3f} ms"]
    for name in sorted(acc):
        v = acc[name]
        lines.append(f"{name:34s} mean per launch {sum(v) / len(v):.4e}   ({len(v)} samples)")
    mean = {k: sum(v) / len(v) for k, v in acc.items()}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in mean and "GRBM_GUI_ACTIVE" in mean:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs (256 CUs x 4)
        kernel_cycles = mean["GRBM_GUI_ACTIVE"] / 8
        busy = mean["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024
        lines.append(f"derived: kernel {kernel_cycles:.4e} shader cycles (GRBM_GUI_ACTIVE / 8 XCDs); matrix pipe busy "
                     f"{busy:.4e} cycles per SIMD = {100 * busy / kernel_cycles:.1f} % of the kernel")
    if "FETCH_SIZE" in mean:
        lines.append(f"derived: HBM read {mean['FETCH_SIZE'] * 1024 * 2 / 1e9:.3f} GB per launch (FETCH_SIZE KiB x 1024 x 2, the guide's gfx950 correction)")
    if "TCC_HIT_sum" in mean:
        lines.append(f"derived: L2 hits {mean['TCC_HIT_sum'] * 128 / 1e9:.1f} GB per launch at 128 B per request")
    open(dst, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


def stats(src, dst):
    rows = list(csv.DictReader(open(src)))
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                        r["MinNs"], r["MaxNs"], r["StdDev"]])


def pmc(fetch_csv, write_csv, needle, rows_per_launch, row_read_bytes, dst, commit="unknown"):
    def collect(path, counter):
        vals, durs = [], []
        for r in csv.DictReader(open(path)):
            if needle in r["Kernel_Name"] and r["Counter_Name"] == counter:
                vals.append(float(r["Counter_Value"]))
                durs.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        return vals, durs

    fv, fd = collect(fetch_csv, "FETCH_SIZE")
    wv, _ = collect(write_csv, "WRITE_SIZE")
    fetch_kib = sum(fv) / len(fv)
    write_kib = sum(wv) / len(wv)
    read_bytes = fetch_kib * 1024 * 2
    write_bytes = write_kib * 1024
    out = {
        "kernel": needle, "rows_per_launch": rows_per_launch, "launches_sampled": len(fv),
        "FETCH_SIZE_KiB_mean": fetch_kib, "WRITE_SIZE_KiB_mean": write_kib,
        "read_bytes_per_launch": read_bytes,
        "read_bytes_note": "FETCH_SIZE*1024*2 (gfx950: FETCH_SIZE counts half of a 16 B/lane stream)",
        "write_bytes_per_launch": write_bytes,
        "traffic_bytes_per_launch": read_bytes + write_bytes,
        "algorithmic_read_bytes_per_launch": rows_per_launch * row_read_bytes,
        "algorithmic_write_bytes_per_launch": rows_per_launch * 4,
        "traffic_over_algorithmic": (read_bytes + write_bytes) / (rows_per_launch * (row_read_bytes + 4)),
        "mean_kernel_ns_in_fetch_pass": sum(fd) / len(fd),
        # bench.py quotes this file as roofline.traffic only while the kernel source is the one profiled
        "commit": commit,
        "kernel_source_sha256_16": kernel_source_hash(),
    }
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "counters":
    counters([sys.argv[2]] + sys.argv[5:], sys.argv[3], sys.argv[4])
    sys.exit(0)

if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5]), int(sys.argv[6]), sys.argv[7],
            sys.argv[8] if len(sys.argv) > 8 else "unknown")
