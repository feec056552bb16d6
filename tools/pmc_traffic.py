#!/usr/bin/env python
"""HBM bytes per conv launch from two rocprofv3 PMC passes (run on the GPU box, see the recipe below) ->
profiles/r05_pmc_traffic.json, which bench.py reports as roofline.traffic (it carries the hash of the conv sources it was measured
on; bench.py reports null when the sources have changed since).

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch -o fetch --output-format csv -- \
        python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt --no-latency --no-other-configs --pipeline 0 --micro 1
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write -o write --output-format csv -- \
        python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-latency --no-other-configs --pipeline 0 --micro 1
    python $R/tools/pmc_traffic.py $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write $R/gpurun_out/pmc_traffic.json

Counters are collected in their own passes (--kernel-trace only).  Units and the gfx950 correction follow
/opt/skills/guides/MI355X_MICROARCH.md: both counters are in KiB; FETCH_SIZE reports half of the bytes of wide
coalesced streaming reads on gfx950, so reads are doubled; WRITE_SIZE is taken as is."""
import csv
import glob
import json
import os
import sys


FORWARDS = 9          # full 64-image forwards of `bench.py --steps 1 --warmup 1 --micro 1`: 2 initial + 1 warm-up + 1 timed + 1 exactness check + 4 profiled
PER_FORWARD = 69      # MFMA conv launches per forward


def total(dirname, counter):
    """Sum of `counter` over the conv launches of the full-size forwards.  The calibration pass of disco_finalize (2 images, every producer
    at least twice) comes first in dispatch order and is dropped: its launches are ~3 % of a 64-image launch each and would only dilute
    a per-launch average."""
    files = glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit("no counter_collection.csv under " + dirname)
    rows = []
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") == counter and "conv3x3_mx_kernel" in row.get("Kernel_Name", ""):
                    rows.append((int(row.get("Dispatch_Id", len(rows))), float(row["Counter_Value"])))
    rows.sort()
    keep = FORWARDS * PER_FORWARD
    if len(rows) < keep:
        raise SystemExit("%d conv launches, expected at least %d" % (len(rows), keep))
    rows = rows[len(rows) - keep:]
    per_layer = [0.0] * PER_FORWARD            # launch i of a forward = layer i (the order of tools/profile_layers.py's table)
    for i, (_, v) in enumerate(rows):
        per_layer[i % PER_FORWARD] += v / FORWARDS
    return sum(v for _, v in rows), len(rows), per_layer


def mfma_busy(sq_dir):
    """SQ_VALU_MFMA_BUSY_CYCLES / (4 SQ_BUSY_CU_CYCLES) over every conv3x3_mx_kernel launch of an SQ pass (four SIMDs per CU): the share of
    the CUs' busy time in which a matrix pipe was executing."""
    busy = cu = 0.0
    for f in glob.glob(os.path.join(sq_dir, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if "conv3x3_mx_kernel" not in row.get("Kernel_Name", ""):
                    continue
                if row["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
                    busy += float(row["Counter_Value"])
                elif row["Counter_Name"] == "SQ_BUSY_CU_CYCLES":
                    cu += float(row["Counter_Value"])
    return round(busy / (4.0 * cu), 4) if cu else None


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    sq_dir = sys.argv[4] if len(sys.argv) > 4 else None
    fetch, nf, fetch_l = total(fetch_dir, "FETCH_SIZE")
    write, nw, write_l = total(write_dir, "WRITE_SIZE")
    with open(os.path.splitext(out)[0] + "_layers.txt", "w") as fh:
        fh.write("# HBM-side MB per conv launch by position in the forward (reads = FETCH_SIZE x 2 KiB, writes = WRITE_SIZE KiB); the positions are the\n"
                 "# rows of profiles/r05_final_layers.txt (0-17 SpixelNet, 18-44 ColorProbNet, 45-68 HourGlass2)\n")
        for i in range(PER_FORWARD):
            fh.write("%3d  read %8.1f MB  write %8.1f MB\n" % (i, fetch_l[i] * 2048 / 1e6, write_l[i] * 1024 / 1e6))
    if not nf or nf != nw:
        raise SystemExit("launch counts differ: %d vs %d" % (nf, nw))
    rd = fetch * 1024 * 2 / nf
    wr = write * 1024 / nw
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    json.dump({
        "source_hash": bench.source_hash(),
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) around "
                  "`python bench.py --steps 1 --warmup 1 --no-alt --no-latency --no-other-configs --pipeline 0 --micro 1` (tools/pmc_traffic.py)",
        "unit_note": "counter unit is KiB; per MI355X_MICROARCH.md the gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of "
                     "wide coalesced streaming reads, so reads are doubled below; WRITE_SIZE is uncalibrated and taken as is",
        "conv_launches (conv3x3_mx_kernel, all arithmetics; 64-image forwards only: the calibration pass is dropped)": nf, "fetch_kib_sum_raw": fetch, "write_kib_sum": write,
        "hbm_read_bytes_per_launch_corrected": int(rd), "hbm_write_bytes_per_launch": int(wr),
        "hbm_bytes_per_launch": int(rd + wr),
        "mfma_busy": mfma_busy(sq_dir) if sq_dir else None,
        "mfma_busy_note": "SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES) summed over the conv3x3_mx_kernel launches of a third pass "
                          "(--pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES ...)"}, open(out, "w"), indent=1)
    print(open(out).read())


if __name__ == "__main__":
    main()
