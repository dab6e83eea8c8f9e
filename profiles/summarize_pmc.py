#!/usr/bin/env python3
"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `python3 bench.py` into profiles/pmc_traffic.json.

    python3 profiles/summarize_pmc.py <fetch counter_collection.csv> <write counter_collection.csv> [out.json]

Per MI355X_MICROARCH.md (HBM / rocprofv3 section): counters are collected in separate --pmc passes together with
--kernel-trace only; FETCH_SIZE / WRITE_SIZE are in KiB (x1024 -> bytes here); on gfx950 FETCH_SIZE under-reports
streamed reads by 2x, which this script re-checks on every run with a kernel whose read volume is known exactly
(dec_out_bwd_kernel<1> reads the 2.147e9-byte hidden activation H once): the `calibration` entry
shows raw and corrected figures side by side.  hbm_bytes_per_launch = 2 * FETCH + WRITE, mean over launches.
"""
import csv
import json
import sys
from collections import defaultdict

ENTRY = {'conv1_fwd_img_kernel': 'tvae_conv1_fwd', 'conv1_wgrad_img_kernel': 'tvae_conv1_wgrad',
         'conv1_fwd_x6_kernel': 'tvae_conv1_fwd_x6', 'conv1_wgrad_x6_kernel': 'tvae_conv1_wgrad_x6',
         'dy_split3_kernel': 'tvae_dy_split3',
         # dense_x6_kernel<XV, parts>: XV 0 = operand from memory (spectral GEMM of the convolution, decoder layers
         # without the recomputed first layer), 2 = forward with the recomputed first-layer operand, 1 / 3 = data gradient
         # with the implicit gradient operand (generic / two-valued); the bench line looks its launch up by grid size
         'dense_x6_kernel<0, 3>': 'tvae_linear_fwd_x6', 'dense_x6_kernel<2, 3>': 'tvae_linear_fwd_x6',
         'dense_x6_kernel<1, 3>': 'tvae_linear_dgrad_x6', 'dense_x6_kernel<3, 3>': 'tvae_linear_dgrad_x6',
         'dense_x6_kernel<4, 3>': 'tvae_linear_dgrad_x6',        # round 3: two-valued + row sums of H (no dec_out_bwd pass)
         # the same launches in the h3 arithmetic (parts = 2: the default since round 3)
         'dense_x6_kernel<0, 2>': 'tvae_linear_fwd_x6', 'dense_x6_kernel<2, 2>': 'tvae_linear_fwd_x6',
         'dense_x6_kernel<1, 2>': 'tvae_linear_dgrad_x6', 'dense_x6_kernel<3, 2>': 'tvae_linear_dgrad_x6',
         'dense_x6_kernel<4, 2>': 'tvae_linear_dgrad_x6',
         # round 4: the data gradient from sign bits (XV 5) and the lean-epilogue instances <XV, parts, EPI>
         'dense_x6_kernel<5, 2>': 'tvae_linear_dgrad_x6', 'dense_x6_kernel<5, 3>': 'tvae_linear_dgrad_x6',
         'dense_x6_kernel<2, 2, 1>': 'tvae_linear_fwd_x6', 'dense_x6_kernel<2, 3, 1>': 'tvae_linear_fwd_x6',
         'dense_x6_kernel<5, 2, 2>': 'tvae_linear_dgrad_x6', 'dense_x6_kernel<5, 3, 2>': 'tvae_linear_dgrad_x6',
         'dense_x6_plain4_kernel': 'tvae_spectral_fwd', 'dense_x6_xres_kernel': 'tvae_spectral_fwd',
         'dft_out_ring_kernel': 'tvae_dft_out', 'dft_dy_ring_kernel': 'tvae_dft_dy',
         'enc_tail_wgrad_x6_kernel': 'tvae_enc_tail_wgrad_x6', 'dft_dbank_kernel': 'tvae_dft_dbank',
         'dft_spectra_kernel': 'tvae_dft_spectra', 'dft_spectra_x_kernel': 'tvae_dft_spectra', 'dft_dbank_x_kernel': 'tvae_dft_dbank',
         'dft_dbank_mf_kernel': 'tvae_dft_dbank',
         'dense_wgrad_x6_dma_kernel<true': 'tvae_linear_wgrad_x6', 'dense_wgrad_x6_dma_kernel<false': 'tvae_spectral_wgrad', 'dense_wgrad_x6_wide_kernel': 'tvae_spectral_wgrad',
         'dft_out_mf_kernel': 'tvae_dft_out', 'dft_dy_mf_kernel': 'tvae_dft_dy',
         'dft_out_gen_kernel': 'tvae_dft_out', 'dft_dy_gen_kernel': 'tvae_dft_dy',
         'gemm_f32_glds_kernel': 'tvae_conv2_fwd', 'gemm_f32_glds2_kernel': 'tvae_conv2_dgrad',
         'heads_fwd_kernel': 'tvae_heads_fwd', 'heads_bwd_kernel': 'tvae_heads_bwd',
         'enc_tail_fwd_x6_kernel': 'tvae_enc_tail_fwd_x6', 'enc_tail_dgrad_x6_kernel': 'tvae_enc_tail_dgrad_x6',
         'gemm_f32_kernel<': 'tvae_conv2_wgrad',
         # calibration kernels with an exactly known read volume: round <= 2 dec_out_bwd_kernel<1> (reads H [512][B*4096] once);
         # round 3 (that pass is folded into the data-gradient launch): heads_bwd_kernel<7> in its sums-only form reads
         # H [128][N] + dheads [7][N] = 540 N bytes and writes next to nothing
         'dec_out_bwd_kernel<1>': 'calibration_dec_out_bwd'}


def per_kernel(path, counter):
    """bytes per launch, grouped by (kernel name, grid size): one kernel can be launched with several shapes per step"""
    acc = defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row['Counter_Name'] == counter:
                acc[row['Kernel_Name'] + ' @grid' + row['Grid_Size']].append(float(row['Counter_Value']) * 1024.0)
    return acc


def main():
    fetch = per_kernel(sys.argv[1], 'FETCH_SIZE')
    write = per_kernel(sys.argv[2], 'WRITE_SIZE')
    out = {}
    for name, vals in fetch.items():
        for key, entry in ENTRY.items():
            if key in name:
                fb = sum(vals) / len(vals)
                wv = write.get(name, [0.0])
                wb = sum(wv) / len(wv)
                # several launch shapes per kernel: keep the one with the largest grid (for dense_x6_kernel that is the
                # decoder layer, forward and data-gradient launches averaged)
                grid = int(name.rsplit('@grid', 1)[1])
                out[entry + '@grid%d' % grid] = {'kernel': name[:64], 'grid': grid, 'launches': len(vals),
                                                 'fetch_size_bytes_raw': fb, 'write_size_bytes': wb,
                                                 'hbm_bytes_per_launch': 2.0 * fb + wb}
                if entry in out and out[entry]['grid'] >= grid:
                    continue
                out[entry] = {'kernel': name[:64], 'grid': grid, 'launches': len(vals), 'fetch_size_bytes_raw': fb,
                              'write_size_bytes': wb, 'hbm_bytes_per_launch': 2.0 * fb + wb,
                              'note': 'FETCH_SIZE doubled (gfx950 reports 1/2 of streamed bytes; see calibration entry)'}
    dst = sys.argv[3] if len(sys.argv) > 3 else 'profiles/pmc_traffic.json'
    json.dump(out, open(dst, 'w'), indent=1)
    for k, v in sorted(out.items()):
        print(f"{k:26s} fetch_raw {v['fetch_size_bytes_raw']/1e9:8.3f} GB  write {v['write_size_bytes']/1e9:7.3f} GB  "
              f"hbm {v['hbm_bytes_per_launch']/1e9:8.3f} GB  ({v['launches']} launches)")


if __name__ == '__main__':
    main()
