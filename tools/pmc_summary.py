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
    kernels = {"chain_kernel": "chain_kernel", "conv2_kernel<false, false, false>": "conv2_kernel (conv2)", "conv2_kernel<true, false, false>": "conv2_kernel<LINEAR> (linear_out)",
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


def by_grid(path, counter, needle):
    """{Grid_Size (threads): [counter values]} of the kernels whose name contains `needle`."""
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and needle in r["Kernel_Name"]:
                acc[int(r["Grid_Size"])].append(float(r["Counter_Value"]))
    return acc


def main():
    """Traffic per launch, grouped by launch width (Grid_Size).  A kernel runs at several widths in one engine pass (the chain
    kernel: 12 encoder launches at B x T' rows, 9 decoder-side ones at B x U) and, unless the profiled command is
    `bench.py --exit-after-timed`, at the widths of the one-batch passes bench.py adds after its timed region: a mean over all
    launches of a kernel mixes them (round 2's pmc_row_chain.json did: 194 MB where the 80,000-row launch moves 375).  Here:
    `by_grid` = every width seen; `widest_launch` = the encoder-sized launch; `hbm_bytes_per_launch` = the mean over the launches
    of the passes of the given width only (grids that occur in a ten-batch pass), i.e. per launch like bench.py's `achieved`."""
    fdir, wdir, out = sys.argv[1:4]
    if len(sys.argv) > 4 and sys.argv[4] not in ("", "-"):
        mfma_summary(sys.argv[4], out)
    c = int(sys.argv[5]) if len(sys.argv) > 5 else 10  # batches of 32 x 1000 frames per engine pass in the profiled runs
    rows_per_batch = int(sys.argv[6]) if len(sys.argv) > 6 else 8000
    targets = {"row_chain": ("chain_kernel", 128, f"chain_kernel (21 launches per engine pass: 12 encoder launches at {rows_per_batch * c} rows, 9 decoder-side)"),
               "conv2": ("conv2_kernel<false, false, false>", 256, f"conv2_kernel<false, false, false> (LDS-DMA implicit GEMM, {c} batches of 32 x 1000 frames per launch)"),
               "linear_out": ("conv2_kernel<true, false, false>", 256, f"conv2_kernel<true, false, false> (linear_out on the LDS-DMA tile kernel, {rows_per_batch * c} x 5120 -> 256)")}
    for tag, (needle, wg_rows, label) in targets.items():
        fg, wg = by_grid(fdir, "FETCH_SIZE", needle), by_grid(wdir, "WRITE_SIZE", needle)
        if not fg or not wg:
            print("no samples for", needle)
            continue
        grids = {}
        for g in sorted(set(fg) & set(wg), reverse=True):
            f_kb, w_kb = sum(fg[g]) / len(fg[g]), sum(wg[g]) / len(wg[g])
            grids[g] = {"workgroups": g // 256, "launches_sampled": len(fg[g]), "FETCH_SIZE_KB": round(f_kb), "WRITE_SIZE_KB": round(w_kb),
                        "hbm_bytes": round((2 * f_kb + w_kb) * 1024)}
        widest = max(grids)
        # launches that belong to a pass of `c` batches: the widest grid, and every grid that occurs as often per widest launch
        # as a pass prescribes and is not also the widest grid of a one-batch pass (chain: 625 encoder / 63 one-batch encoder workgroups)
        # Launches that belong to a pass of `c` batches: everything but the grids of a ONE-batch pass, which only occur when the
        # profiled command was not `bench.py --exit-after-timed` (bench.py then adds one-batch passes after its timed region):
        # the one-batch encoder-sized launch has 1 / c of the widest launch's rows, and its decoder-side launches are narrower still
        one_wgs = -(-(widest // 256 * wg_rows // c) // wg_rows) if c > 1 else -1
        has_one_batch_pass = any(abs(g // 256 - one_wgs) <= 1 for g in grids)
        wide = [g for g in grids if not has_one_batch_pass or g // 256 > one_wgs + 1 or g == widest]
        if has_one_batch_pass and tag == "row_chain":  # decoder-side launches of the wide pass: wider than the one-batch encoder launch
            wide = [g for g in grids if g // 256 > one_wgs + 1]
        n = sum(grids[g]["launches_sampled"] for g in wide)
        mean = sum(grids[g]["hbm_bytes"] * grids[g]["launches_sampled"] for g in wide) / max(1, n)
        rec = {"kernel": label,
               "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, program directly after --), bench.py --streams 1 --coalesce {c} "
                         f"(one pipeline, {c} batches of 32 per engine pass: launches as wide as the timed run's), grouped by Grid_Size",
               "batches_per_engine_pass": c,
               "correction": "gfx950 FETCH_SIZE counts 64 B per 128-B request for wide (16 B/lane) coalesced reads: doubled "
                             "(MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact; both in KB of 1024 B",
               "by_grid": {str(g): v for g, v in grids.items()},
               "widest_launch": dict(grids[widest], grid=widest),
               "grids_of_the_wide_pass": wide,
               "hbm_bytes_per_launch": round(mean)}
        json.dump(rec, open(os.path.join(out, f"pmc_{tag}.json"), "w"), indent=2)
        print(tag, json.dumps({k: rec[k] for k in ("widest_launch", "grids_of_the_wide_pass", "hbm_bytes_per_launch")}))


if __name__ == "__main__":
    main()
