#!/usr/bin/env python3
"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `python3 bench.py` into profiles/pmc_traffic.json.

    python3 profiles/summarize_pmc.py <fetch counter_collection.csv> <write counter_collection.csv> [out.json]

Per MI355X_MICROARCH.md (HBM / rocprofv3 section): counters are collected in separate --pmc passes together with
--kernel-trace only; FETCH_SIZE / WRITE_SIZE are in KiB (x1024 -> bytes here); on gfx950 FETCH_SIZE under-reports
streamed reads by 2x, which this script re-checks on every run with a kernel whose read volume is known exactly
(dec_out_bwd_kernel<1> / outer_mask_kernel<1> read the 2.147e9-byte hidden activation H once and write as much): the `calibration` entry
shows raw and corrected figures side by side.  hbm_bytes_per_launch = 2 * FETCH + WRITE, mean over launches.
"""
import csv
import json
import sys
from collections import defaultdict

ENTRY = {'conv1_fwd_img_kernel': 'tvae_conv1_fwd', 'conv1_wgrad_img_kernel': 'tvae_conv1_wgrad',
         'conv1_fwd_x6_kernel': 'tvae_conv1_fwd_x6', 'conv1_wgrad_x6_kernel': 'tvae_conv1_wgrad_x6',
         'dy_split3_kernel': 'tvae_dy_split3',
         'outer_mask_kernel<1>': 'calibration_outer_mask', 'dec_out_bwd_kernel<1>': 'calibration_dec_out_bwd'}


def per_kernel(path, counter):
    acc = defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row['Counter_Name'] == counter:
                acc[row['Kernel_Name']].append(float(row['Counter_Value']) * 1024.0)
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
                # the largest launch shape of a kernel dominates; keep the entry with the most bytes
                if entry in out and out[entry]['fetch_size_bytes_raw'] > fb:
                    continue
                out[entry] = {'kernel': name[:64], 'launches': len(vals), 'fetch_size_bytes_raw': fb,
                              'write_size_bytes': wb, 'hbm_bytes_per_launch': 2.0 * fb + wb,
                              'note': 'FETCH_SIZE doubled (gfx950 reports 1/2 of streamed bytes; see calibration entry)'}
    dst = sys.argv[3] if len(sys.argv) > 3 else 'profiles/pmc_traffic.json'
    json.dump(out, open(dst, 'w'), indent=1)
    for k, v in out.items():
        print(f"{k:26s} fetch_raw {v['fetch_size_bytes_raw']/1e9:8.3f} GB  write {v['write_size_bytes']/1e9:7.3f} GB  "
              f"hbm {v['hbm_bytes_per_launch']/1e9:8.3f} GB  ({v['launches']} launches)")


if __name__ == '__main__':
    main()
