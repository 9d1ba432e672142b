#!/usr/bin/env python3
"""Condense rocprofv3 --pmc runs of bench.py (gpurun_out/pmc_<COUNTER>/...) into profiles/<tag>_pmc_summary.json.

Passes (each its own run, as MI355X_MICROARCH.md prescribes -- FETCH_SIZE and WRITE_SIZE do not fit one pass):
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_FETCH_SIZE -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE ...
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE ...
Corrections applied (guide, HBM section): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the
bytes of wide coalesced reads -> doubled.  GRBM_GUI_ACTIVE comes back summed over the 8 XCDs -> divided by 8.
"""
import collections
import re
import csv
import glob
import json
import os
import sys

# 'conv_patch_bf16_bn128' = both instantiations of the 128-channel-tile kernel (forward with statistics, data gradient)
KERNELS = {'conv3x3_patch_k<128': 'conv_patch_bf16_bn128', 'conv3x3_patch_k<128, true': 'conv_patch_bf16_bn128_fwd',
           'conv3x3_patch_k<128, false': 'conv_patch_bf16_bn128_dgrad', 'wgrad3x3_patch_k<128, 64': 'wgrad_patch_bf16_128x64'}


# BASELINE config 2 (second set of passes, directory suffix 'c2'): the Pong-class band kernels
# config 2 (template arguments of the band kernel: input channels / 16, output channels / 16, BatchNorm input, statistics, ...)
KERNELS_C2 = {'conv3x3_small_band_k<': 'conv_small_band_bf16', 'conv3x3_small_band_k<1, [12], false': 'conv_small_band_bf16_c16',
              'conv3x3_small_band_k<2, [12], false': 'conv_small_band_bf16_c32', 'conv3x3_small_band_k<1, [12], true': 'conv_small_band_bf16_c16_bnin',
              'conv3x3_small_band_k<2, [12], true': 'conv_small_band_bf16_c32_bnin', 'wgrad3x3_small_band_k<': 'wgrad_small_band_bf16', 'wgrad3x3_small16_k<': 'wgrad_small16_bf16'}


def load(name):
    fs = glob.glob(f'gpurun_out/pmc_{name}/*/*counter_collection.csv')
    if not fs:
        return []
    return list(csv.DictReader(open(max(fs, key=os.path.getmtime))))


def main(tag, lib_sha16=None):
    out = {}
    if lib_sha16:   # the library the counters were collected on (tools/records.sh): bench.py quotes `traffic` only from a summary
        out['lib_sha16'] = lib_sha16   # whose stamp equals the loaded library's
    for suffix, kernels in (('', KERNELS), ('c2', KERNELS_C2)):
        summarise(out, suffix, kernels)
    # configs 3 and 5 (round 6): the same patch / weight-gradient kernels on those workloads' shapes, labels suffixed _c3 / _c5
    for suffix in ('c3', 'c5'):
        summarise(out, suffix, {k: v + '_' + suffix for k, v in KERNELS.items()})
    json.dump(out, open(f'profiles/{tag}_pmc_summary.json', 'w'), indent=1)
    print(json.dumps(out, indent=1))


def summarise(out, suffix, kernels):
    rows = {n: load(n + suffix) for n in ('FETCH_SIZE', 'WRITE_SIZE', 'SQ_VALU_MFMA_BUSY_CYCLES')}
    if not all(rows.values()):
        return
    for key, label in kernels.items():
        acc = collections.defaultdict(lambda: [0.0, 0])
        dur = [0.0, 0]
        for n, rs in rows.items():
            for r in rs:
                if re.search(key, r['Kernel_Name']):   # (keys are regular expressions)
                    a = acc[r['Counter_Name']]
                    a[0] += float(r['Counter_Value']); a[1] += 1
                    if n == 'FETCH_SIZE':
                        dur[0] += int(r['End_Timestamp']) - int(r['Start_Timestamp']); dur[1] += 1
        if 'FETCH_SIZE' not in acc or not dur[1]:   # no launch of this kernel in the profiled run
            continue
        per = {c: v / n for c, (v, n) in acc.items()}
        fetch_b = 2.0 * per['FETCH_SIZE'] * 1024.0
        write_b = per['WRITE_SIZE'] * 1024.0
        gui = per['GRBM_GUI_ACTIVE'] / 8.0
        out[label] = dict(
            launches_sampled=acc['FETCH_SIZE'][1],
            avg_launch_us=dur[0] / dur[1] / 1e3,
            hbm_read_bytes_per_launch=fetch_b, hbm_write_bytes_per_launch=write_b,
            hbm_bytes_per_launch=fetch_b + write_b,
            mfma_busy_cycles_per_launch=per['SQ_VALU_MFMA_BUSY_CYCLES'],
            gpu_active_cycles_per_launch=gui,
            mfma_util=per['SQ_VALU_MFMA_BUSY_CYCLES'] / (gui * 1024.0),
            effective_clock_ghz=gui / (dur[0] / dur[1]),
            lds_bank_conflict_frac=per['SQ_LDS_BANK_CONFLICT'] / max(1.0, per['SQ_LDS_IDX_ACTIVE']))


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else 'r01', sys.argv[2] if len(sys.argv) > 2 else None)
