#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel stats + PMC passes) into the small tables committed under profiles/.

    python3 profiles/summarize.py <stats_dir> <pmc_fetch_dir> <pmc_write_dir> <pmc_sq_dir> <out_prefix>

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE are collected in separate
passes, are in KiB, and on gfx950 FETCH_SIZE reports half the bytes of wide streaming reads -> read bytes = 2 * 1024 *
FETCH_SIZE, write bytes = 1024 * WRITE_SIZE.  Clock = GRBM_GUI_ACTIVE / 8 / duration.
"""
import collections
import csv
import glob
import json
import re
import sys


def kname(n):
    m = re.search(r"(fwht_adaround_adam_multi_kernel|adaround_fwht_multi_kernel|weight_layouts_all_kernel|conv_wgrad_flat3_kernel|conv_flat3_kernel|head_loss_stage2|conv_igemm3_kernel|conv_wgrad3p_kernel|conv_wgrad3_kernel|wgrad3_reduce_kernel|wgrad_reduce_multi_kernel|adaround_adam_multi_kernel|step_prologue\w*|uaq_\w+_multi_kernel|fwht_multi_kernel|l2_tanh_head_stage\d|weight_layouts_multi_kernel|weight_layout3_multi_kernel|weight_layout3_kernel|adaround_multi_kernel|adam_multi_kernel|channel_sum_stage\d|conv_igemm_kernel|conv_wgrad_kernel|head_\w+_kernel|conv_splitk_finish_kernel|tiny_pw_\w+_kernel|wgrad_reduce_kernel|"
                  r"weight_\w+_layout_kernel|adaround_\w+_kernel|uaq_\w+_kernel|adam_kernel|l2_loss_stage1|nq_sum_stage2|"
                  r"tanh_out_bwd_kernel|gather_u8_kernel|fwht_kernel|round_loss\w*|ps_gelu_bwd_kernel|frame_sse_kernel|"
                  r"scale_init_kernel)(<[^>]*>)?", n)
    return (m.group(1) + (m.group(2) or "")) if m else n[:60]


def load_pmc(d):
    f = glob.glob(d + "/*/*_counter_collection.csv")[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        key = (kname(r["Kernel_Name"]), r["Grid_Size"])
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[key]["dur_ns:" + r["Counter_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return agg


def main():
    stats_dir, fetch_dir, write_dir, sq_dir, out = sys.argv[1:6]
    rows = list(csv.DictReader(open(glob.glob(stats_dir + "/*/*_kernel_stats.csv")[0])))
    with open(out + "_kernel_stats.csv", "w") as f:
        f.write("kernel,calls,total_ms,avg_us,percent\n")
        for r in rows:
            f.write(f'"{kname(r["Name"])}",{r["Calls"]},{float(r["TotalDurationNs"]) / 1e6:.3f},'
                    f'{float(r["AverageNs"]) / 1e3:.1f},{float(r["Percentage"]):.2f}\n')
    fetch, write, sq = load_pmc(fetch_dir), load_pmc(write_dir), load_pmc(sq_dir)
    table = []
    for key in fetch:
        name, grid = key
        c = fetch[key]
        # the convolution / head kernels, and (round 4) every other kernel of the iteration that lasts >= 8 us: the parameter
        # side (fake-quant, FWHT, operand layouts, slab reduction, d(alpha)+Adam) is a sixth of a NeRV-3M + Hadamard step
        if not (name.startswith("conv_") or name.startswith("head_")) and sum(c["dur_ns:FETCH_SIZE"]) / len(c["FETCH_SIZE"]) < 8e3:
            continue
        # kernels that are not this library's (ATen elementwise / MIOpen / rocBLAS launches of the one-time set-up: synthetic
        # frames, the FP32 ConvNeXt encoder precompute) are not part of an iteration
        if name.startswith("void at::") or name.startswith("naive_conv") or name.startswith("Cijk_") or "miopen" in name.lower():
            continue
        n = len(c["FETCH_SIZE"])
        dur = sum(c["dur_ns:FETCH_SIZE"]) / n
        rd = 2 * 1024 * sum(c["FETCH_SIZE"]) / n
        wr = 1024 * sum(write[key]["WRITE_SIZE"]) / max(len(write[key]["WRITE_SIZE"]), 1) if key in write else None
        clk = sum(c["GRBM_GUI_ACTIVE"]) / n / 8 / dur if c.get("GRBM_GUI_ACTIVE") else None
        row = dict(kernel=name, grid=int(grid), launches=n, dur_us=round(dur / 1e3, 1), hbm_read_MB=round(rd / 1e6, 1),
                   hbm_write_MB=None if wr is None else round(wr / 1e6, 1), clock_GHz=None if clk is None else round(clk, 2))
        # a clock above the chip's 2.4 GHz maximum on a dispatch long enough for GRBM_GUI_ACTIVE / 8 / duration to mean
        # something (guide: the quotient reads high below ~0.3 ms) is a broken pass, not a measurement (round 2 committed
        # one: 3.49 GHz on the dominant kernel); such rows are marked and make_traffic.py refuses them
        if clk is not None and clk > 2.45 and dur >= 300e3:
            row["clock_suspect"] = True
        if key in sq:
            s = {k: sum(v) / len(v) for k, v in sq[key].items() if not k.startswith("dur_ns")}
            sdur = sum(sq[key]["dur_ns:SQ_WAVE_CYCLES"]) / len(sq[key]["dur_ns:SQ_WAVE_CYCLES"])
            wc = s.get("SQ_WAVE_CYCLES", 0) or 1
            # the clock of THIS pass when it was collected with it (round 3: GRBM_GUI_ACTIVE rides along in the SQ pass, so
            # the busy fraction and its clock come from the same dispatches), else the fetch pass's
            clk_sq = s["GRBM_GUI_ACTIVE"] / 8 / sdur if s.get("GRBM_GUI_ACTIVE") else clk
            if clk_sq is not None and clk_sq > 2.45 and sdur >= 300e3:
                row["clock_suspect"] = True
            if "GRBM_GUI_ACTIVE" in s:
                row["clock_GHz_sq_pass"] = round(clk_sq, 2)
            cyc = (clk_sq or 2.2) * sdur * 1024  # SIMD-cycles available during the dispatch (256 CUs x 4 SIMDs)
            row.update(mfma_busy_frac=round(s.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / cyc, 3),
                       wait_any=round(s.get("SQ_WAIT_ANY", 0) / wc, 3), wait_inst=round(s.get("SQ_WAIT_INST_ANY", 0) / wc, 3),
                       active=round(s.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3),
                       valu_active=round(s.get("SQ_ACTIVE_INST_VALU", 0) / wc, 3),
                       lds_bank_conflict_cycles=int(s.get("SQ_LDS_BANK_CONFLICT", 0)))
        table.append(row)
    table.sort(key=lambda r: -r["dur_us"] * r["launches"])
    json.dump(table, open(out + "_pmc_conv_kernels.json", "w"), indent=1)
    for r in table[:24]:
        print(r)


if __name__ == "__main__":
    main()
