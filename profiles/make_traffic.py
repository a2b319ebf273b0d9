#!/usr/bin/env python3
"""profiles/hbm_traffic.json (what bench.py reports as roofline.traffic) from a PMC summary made by summarize.py.

    python3 profiles/make_traffic.py profiles/r04_f_pmc_conv_kernels.json profiles/r04_f_nerv_pmc_conv_kernels.json \
        profiles/r04_fp32_pmc_conv_kernels.json profiles/r04_f_uvg_pmc_conv_kernels.json

bytes = read + write per launch at per-GPU batch 2: FETCH_SIZE x 2 (gfx950 reports half the bytes of 16-byte-per-lane
streaming reads, MI355X_MICROARCH.md §HBM; separate --pmc pass) + WRITE_SIZE (its own pass).  FETCH_SIZE counts the L2's
fabric-side requests, Infinity-Cache hits included (same guide), so for kernels whose workgroups re-read a tile that is
still resident in the 256 MiB cache it is an UPPER reading of the HBM bytes.  bench.py only reports an entry while the
live kernel duration is within 20 % of `dur_us` recorded here (profiled runs are a few % slower than plain ones).
"""
import json
import sys

# bench key -> (kernel name in the PMC summary, grid size, tag of the summary file it must come from)
# (conv_igemm3_kernel<MI, WPE, XS>: channel blocks per tile, waves per SIMD the build aims at, input as split {hi | lo} words)
MAP = {
    # HNeRV Bunny_1280x640_3M, B = 2, bf16x3 (r04_f)
    "conv_wgrad3_k5_44_148": ("conv_wgrad3p_kernel<5, 6, 1, 4>", 129024, "r04_f_pmc"),
    "conv_igemm3_k5_44_148": ("conv_igemm3_kernel<5, 2, true>", 819200, "r04_f_pmc"),
    "conv_igemm3_k5_148_44": ("conv_igemm3_kernel<3, 2, true>", 409600, "r04_f_pmc"),
    "conv_wgrad3_k5_53_176": ("conv_wgrad3p_kernel<4, 7, 1, 4>", 129024, "r04_f_pmc"),
    "conv_igemm3_k5_53_176": ("conv_igemm3_kernel<4, 2, true>", 307200, "r04_f_pmc"),
    "conv_igemm3_k5_176_53": ("conv_igemm3_kernel<4, 2, true>", 102400, "r04_f_pmc"),
    # NeRV Bunny_1280x640_3M + Hadamard (r04_f_nerv): the `nerv` object's dominant kernel
    "conv_igemm3_k3_24_96": ("conv_igemm3_kernel<3, 3, true>", 819200, "r04_f_nerv_pmc"),
    # the same HNeRV workload with exact-fp32 convolutions (r04_fp32): the `fp32` object
    "conv_igemm_k5_44_148": ("conv_igemm_kernel<10>", 819200, "r04_fp32_pmc"),
    "conv_igemm_k5_148_44": ("conv_igemm_kernel<3>", 819200, "r04_fp32_pmc"),
    "conv_wgrad_k5_44_148": ("conv_wgrad_kernel<10, 3>", 130560, "r04_fp32_pmc"),
    # HNeRV UVG 960x1920 ~12M (r04_f_uvg): the `uvg` object
    "conv_wgrad3_k5_89_296": ("conv_wgrad3p_kernel<4, 7, 1, 4>", 128000, "r04_f_uvg_pmc"),
    "conv_igemm3_k5_89_296": ("conv_igemm3_kernel<4, 2, true>", 4608000, "r04_f_uvg_pmc"),
    "conv_igemm3_k5_296_89": ("conv_igemm3_kernel<3, 2, true>", 1843200, "r04_f_uvg_pmc"),
}


def main(*srcs):
    rows = []
    for src_ in srcs:   # (the HNeRV set, optionally the NeRV set: its 24 -> 96 forward is the `nerv` object's dominant kernel)
        rows += [dict(r, _src=src_) for r in json.load(open(src_))]
    out = {}
    for key, (name, grid, tag) in MAP.items():
        for r in rows:
            src = r["_src"]
            if r["kernel"] == name and (grid is None or r["grid"] == grid) and tag in src:
                if r.get("clock_suspect") or ((r.get("clock_GHz") or 0) > 2.45 and r["dur_us"] >= 300):
                    raise SystemExit(f"{name}: clock {r.get('clock_GHz')} GHz on a {r['dur_us']} us dispatch -- broken PMC pass, "
                                     "take it again (profiles/README.md)")
                rd, wr = r["hbm_read_MB"] * 1e6, (r["hbm_write_MB"] or 0) * 1e6
                out[key] = dict(bytes=int(rd + wr), read_bytes=int(rd), write_bytes=int(wr), read_factor=2, batch=2,
                                kernel=name, dur_us=r["dur_us"], mfma_busy=r.get("mfma_busy_frac"), source=src)
                break
    json.dump(out, open("profiles/hbm_traffic.json", "w"), indent=1)
    for k, v in out.items():
        print(k, v["bytes"], v["dur_us"])


if __name__ == "__main__":
    main(*sys.argv[1:])
