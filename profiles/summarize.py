#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (gpurun_out/prof/...) into the small summaries kept under
profiles/.  Kernel names are shortened (torch's are kilobytes long).

    python profiles/summarize.py stats <kernel_stats.csv> <out.csv>
    python profiles/summarize.py pmc   <fetch_counter_collection.csv> <write_counter_collection.csv> \
                                       <kernel-substring[|more]> <rows_per_launch> <row_read_bytes> <out.json> [commit] \
                                       [key = u8_scan | bin_scan | pq_scan_m<m> | u8_batch<Q>_<dim> | bin_batch<Q>_<dim>] \
                                       [launches per scan / step] [score bytes written per row]
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


def kernel_source_hash(root=None, key="u8_scan"):
    """Hash of the source that defines the profiled kernel: a PMC profile is quoted by bench.py only while it matches.
      u8_scan        the part of csrc/u8.hip that defines u8_scan_kernel (device helpers + the kernel)
      pq_scan_m<m>   the scan section of csrc/pq.hip (everything in front of the encode section)
      bin_scan       csrc/bin.hip in front of its "many queries on the matrix cores" part;   bin_batch...   that part + batch_common.hpp
      u8_batch...    csrc/u8_batch.hip + batch_common.hpp"""
    import hashlib
    import os
    root = root or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "quantization_amd", "csrc")
    rd = lambda name: open(os.path.join(csrc, name), "rb").read()
    if key.startswith("u8_scan"):
        src = rd("u8.hip")
        a = src.find(b"// ------------------------------------------------------------------------------ device helpers")
        b = src.find(b"// NQ (2, 4, 8) queries per row read on the vector ALU")
        region = src[a:b] if 0 <= a < b else src
    elif key.startswith("pq_scan"):
        src = rd("pq.hip")
        b = src.find(b"// ------------------------------------------------------------------------------ encode")
        region = src[:b] if b > 0 else src
    elif key.startswith("bin_scan") or key.startswith("bin_batch"):
        src = rd("bin.hip")
        cut = src.find(b"// ============================================================================= many queries on the matrix cores")
        cut = cut if cut > 0 else len(src)
        region = src[:cut] if key.startswith("bin_scan") else src[cut:] + rd("batch_common.hpp")
    elif key.startswith("u8_batch"):
        region = rd("u8_batch.hip") + rd("batch_common.hpp")
    else:
        raise ValueError(key)
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


def pmc(fetch_csv, write_csv, needle, rows_per_launch, row_read_bytes, dst, commit="unknown", key="u8_scan",
        launches_per_unit=1, alg_write_bytes_per_row=4):
    """`needle`: kernel-name substring(s), '|'-separated.  A unit (one scan of the store, one batch step) may be several
    launches of the matching kernels (PQ rows of several LUT slices: one launch per slice): the bytes of all matching
    launches are summed and divided by the number of units."""
    needles = needle.split("|")

    def collect(path, counter):
        vals, durs = [], []
        for r in csv.DictReader(open(path)):
            if any(nd in r["Kernel_Name"] for nd in needles) and r["Counter_Name"] == counter:
                vals.append(float(r["Counter_Value"]))
                durs.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        # the same kernel template also serves short helper passes (the batched top-k's pivot sample): only launches of
        # at least half the longest one are the scan / the filter pass
        keep = [i for i, d in enumerate(durs) if d >= 0.5 * max(durs)]
        return [vals[i] for i in keep], [durs[i] for i in keep]

    fv, fd = collect(fetch_csv, "FETCH_SIZE")
    wv, _ = collect(write_csv, "WRITE_SIZE")
    units_f, units_w = len(fv) / launches_per_unit, len(wv) / launches_per_unit
    fetch_kib = sum(fv) / units_f
    write_kib = sum(wv) / units_w
    read_bytes = fetch_kib * 1024 * 2
    write_bytes = write_kib * 1024
    alg_read, alg_write = rows_per_launch * row_read_bytes, rows_per_launch * alg_write_bytes_per_row
    out = {
        "kernel": needle, "key": key, "rows_per_launch": rows_per_launch, "launches_sampled": len(fv),
        "launches_per_unit": launches_per_unit,
        "FETCH_SIZE_KiB_mean": fetch_kib, "WRITE_SIZE_KiB_mean": write_kib,
        "read_bytes_per_launch": read_bytes,
        "read_bytes_note": "FETCH_SIZE*1024*2 (gfx950: FETCH_SIZE counts half of a 16 B/lane stream); per unit = all "
                           "launches of one scan / one batch step",
        "write_bytes_per_launch": write_bytes,
        "traffic_bytes_per_launch": read_bytes + write_bytes,
        "algorithmic_read_bytes_per_launch": alg_read,
        "algorithmic_write_bytes_per_launch": alg_write,
        "read_over_algorithmic_read": read_bytes / alg_read,
        "traffic_over_algorithmic": (read_bytes + write_bytes) / (alg_read + alg_write),
        "mean_kernel_ns_in_fetch_pass": sum(fd) / len(fd),
        "kernel_ns_per_unit_in_fetch_pass": sum(fd) / units_f,
        # bench.py quotes this file as roofline.traffic only while the kernel source is the one profiled
        "commit": commit,
        "kernel_source_sha256_16": kernel_source_hash(None, key),
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
            sys.argv[8] if len(sys.argv) > 8 else "unknown", sys.argv[9] if len(sys.argv) > 9 else "u8_scan",
            int(sys.argv[10]) if len(sys.argv) > 10 else 1, int(sys.argv[11]) if len(sys.argv) > 11 else 4)
