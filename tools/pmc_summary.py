"""Summarise rocprofv3 --pmc passes (separate runs for FETCH_SIZE and WRITE_SIZE, CSV output) into the per-launch HBM
traffic figures bench.py's roofline objects quote.

    python tools/pmc_summary.py <dir with the FETCH_SIZE pass> <dir with the WRITE_SIZE pass> <out dir (profiles/)> [<dir with the SQ pass> [batches per engine pass]]

The optional SQ pass (--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE)
becomes pmc_mfma.json: per kernel, matrix-pipe busy cycles against the cycles its waves were resident and against the
launch's duration x SIMDs (MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES counts cycles, SQ_WAVE_CYCLES quad-cycles).

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 64 B per 128-B request of a wide coalesced read,
so it is doubled; WRITE_SIZE is exact.  Both are reported by rocprofv3 in units of 1024 B.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def per_kernel(path, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


def mfma_summary(sdir, out):
    names = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_BF16", "GRBM_GUI_ACTIVE"]
    acc = {n: per_kernel(sdir, n) for n in names}
    kernels = {"chain_kernel": "chain_kernel", "conv2_kernel<false, false>": "conv2_kernel (conv2)", "conv2_kernel<true, false>": "conv2_kernel<LINEAR> (linear_out)",
               "attention_kernel": "attention_kernel", "genmax_kernel": "genmax_kernel", "conv1_kernel": "conv1_kernel"}
    c = int(sys.argv[5]) if len(sys.argv) > 5 else 10
    rec = {"source": "rocprofv3 --pmc " + " ".join(names) + f" (its own pass, --kernel-trace only), bench.py --steps {2 * c} --warmup {c} --streams 1 --coalesce {c} "
                     f"(one pipeline, {c} batches of 32 per pass: the kernels have the GPU to themselves)",
           "units": "SQ_VALU_MFMA_BUSY_CYCLES: cycles, summed over SIMDs; SQ_WAVE_CYCLES / SQ_BUSY_CYCLES: quad-cycles (x4); "
                    "GRBM_GUI_ACTIVE: cycles summed over the 8 XCDs (/8); MOPS_BF16 x 512 = bf16 MFMA FLOPs",
           "kernels": {}}
    for needle, label in kernels.items():
        def tot(n):
            return [v for k, vs in acc[n].items() if needle in k for v in vs]
        busy, wave, gui, mops = tot("SQ_VALU_MFMA_BUSY_CYCLES"), tot("SQ_WAVE_CYCLES"), tot("GRBM_GUI_ACTIVE"), tot("SQ_INSTS_VALU_MFMA_MOPS_BF16")
        if not busy:
            continue
        n = len(busy)
        b, w, g, m = sum(busy) / n, 4 * sum(wave) / n, sum(gui) / n / 8, sum(mops) / n
        rec["kernels"][label] = {
            "launches_sampled": n, "mfma_busy_cycles_per_launch": round(b), "wave_cycles_per_launch": round(w),
            "gui_active_cycles_per_launch": round(g), "bf16_mfma_flops_per_launch": round(m * 512),
            "mfma_busy_over_wave_cycles": round(b / w, 4) if w else None,  # share of the resident waves' time with the matrix pipe busy
            "mfma_busy_over_chip_simd_cycles": round(b / (g * 1024), 4) if g else None}  # 256 CUs x 4 SIMDs for the launch's duration
    json.dump(rec, open(os.path.join(out, "pmc_mfma.json"), "w"), indent=2)
    print("mfma", json.dumps(rec["kernels"]))


def main():
    fdir, wdir, out = sys.argv[1:4]
    if len(sys.argv) > 4:
        mfma_summary(sys.argv[4], out)
    fetch, write = per_kernel(fdir, "FETCH_SIZE"), per_kernel(wdir, "WRITE_SIZE")
    c = int(sys.argv[5]) if len(sys.argv) > 5 else 10  # batches of 32 x 1000 frames per engine pass in the profiled runs
    targets = {"row_chain": ("chain_kernel", f"chain_kernel (all 21 launches of an engine pass of {c} batches: 12 encoder at {8000 * c} rows, 9 decoder-side)"),
               "conv2": ("conv2_kernel<false, false>", f"conv2_kernel<false, false> (LDS-DMA implicit GEMM, {c} batches of 32 x 1000 frames per launch)"),
               "linear_out": ("conv2_kernel<true, false>", f"conv2_kernel<true, false> (linear_out on the LDS-DMA tile kernel, {8000 * c} x 5120 -> 256)")}
    for tag, (needle, label) in targets.items():
        fk = [v for k, vs in fetch.items() if needle in k for v in vs]
        wk = [v for k, vs in write.items() if needle in k for v in vs]
        if not fk or not wk:
            print("no samples for", needle)
            continue
        f_kb, w_kb = sum(fk) / len(fk), sum(wk) / len(wk)
        rec = {"kernel": label,
               "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --steps {2 * c} --warmup {c} --streams 1 --coalesce {c} (one pipeline, {c} batches of 32 per engine pass: launches as wide as the timed run's)",
               "batches_per_engine_pass": c, "launches_sampled": len(fk), "FETCH_SIZE_KB_per_launch": round(f_kb), "WRITE_SIZE_KB_per_launch": round(w_kb),
               "correction": "gfx950 FETCH_SIZE counts 64 B per 128-B request for wide (16 B/lane) coalesced reads: doubled "
                             "(MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact",
               "hbm_bytes_per_launch": round((2 * f_kb + w_kb) * 1024)}
        json.dump(rec, open(os.path.join(out, f"pmc_{tag}.json"), "w"), indent=2)
        print(tag, rec)


if __name__ == "__main__":
    main()
