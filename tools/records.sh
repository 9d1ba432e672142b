#!/bin/bash
# Round records on ONE box, stamped with the library that produced them:  bash tools/records.sh <tag>
#   -> gpurun_out/<tag>_*  (bench records of configs 4 / 2 / 3 / 5 / fp32, rocprofv3 kernel stats with the side streams on and off for
#      configs 4, 2, 3 and 5, timeline + overlap reports, the PMC summary of the patch / weight-gradient / band kernels) and
#      gpurun_out/<tag>_MANIFEST.json: library sha256[:16] (the one build() prints), git head, one sha256 per file.
# Copy what is quoted into profiles/ (tools/collect.sh <tag>).  bench.py only quotes `roofline.traffic` from a PMC summary whose
# lib_sha16 equals the loaded library's.
T=${1:-r06x}; R=$PWD; O=$R/gpurun_out; mkdir -p $O
SHA=$(sha256sum $R/keypoints_amd/csrc/libkp_hip.so | cut -c1-16)
echo "library sha256[:16] = $SHA"
C2="--model_type VGG_PONG_LAYERNECK --model_in_channels 1 --model_z_channels 16 --model_keypoints 4 --size 84 --batch 256 --precision bf16"
C3="--net keynet --precision fp16"
C5="--size 256 --model_keypoints 30 --batch 256 --precision fp16"
Q="--no-cpu-baseline --extra 0"
# ---- PMC passes first (bench.py reads the summary for roofline.traffic of the records below)
if [ "${KP_SKIP_PMC:-0}" != 1 ]; then
cd /tmp && export TMPDIR=/tmp
for set in "" "c2" "c3" "c5"; do
  ARGS=""; [ "$set" = c2 ] && ARGS="$C2"; [ "$set" = c3 ] && ARGS="$C3"; [ "$set" = c5 ] && ARGS="$C5"
  for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
    n=$(echo $c | cut -d' ' -f1)$set
    rm -rf $O/pmc_$n
    timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$n -- python $R/bench.py --steps 2 --warmup 1 --profile-steps 0 --streams 0 $Q $ARGS > $O/pmc_$n.log 2>&1
  done
done
cd $R && python tools/pmc_summary.py $T $SHA > /dev/null && cp profiles/${T}_pmc_summary.json $O/
rm -rf $O/pmc_*   # (the raw counter CSVs are hundreds of MB)
fi
cd $R
# ---- bench records (the driver's command first)
python bench.py --steps 20 --warmup 10 --out $O/${T}_bench.json > $O/${T}_bench.line 2>$O/${T}_bench.err
python bench.py --steps 30 --warmup 5 $Q --streams 0 --out $O/${T}_bench_streams0.json > /dev/null 2>&1
python bench.py --steps 30 --warmup 5 $Q --precision bf16 --out $O/${T}_bench_bf16.json > /dev/null 2>&1
python bench.py --steps 100 --warmup 16 $Q $C2 --out $O/${T}_cfg2_bench.json > /dev/null 2>&1
python bench.py --steps 100 --warmup 16 $Q $C2 --graph 1 --out $O/${T}_cfg2_bench_graph.json > /dev/null 2>&1
python bench.py --steps 10 --warmup 3 $Q $C3 --out $O/${T}_cfg3_bench.json > /dev/null 2>&1
python bench.py --steps 5 --warmup 4 $Q $C5 --out $O/${T}_cfg5_bench.json > /dev/null 2>&1
python bench.py --steps 5 --warmup 4 $Q --precision fp32 --out $O/${T}_fp32_bench.json > /dev/null 2>&1
# ---- rocprofv3 kernel stats / timelines
cd /tmp && export TMPDIR=/tmp
prof() {  # name, streams, steps, bench args...
  local name=$1 s=$2 k=$3; shift 3
  rm -rf /tmp/ks_$name
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$name -o k -- python $R/bench.py --steps $k --warmup 2 --profile-steps 0 --streams $s $Q "$@" > /dev/null 2>&1
  cp $(find /tmp/ks_$name -name "*kernel_stats.csv" | head -1) $O/${T}_${name}_kernel_stats_streams$s.csv
  local tr=$(find /tmp/ks_$name -name "*kernel_trace.csv" | head -1)
  python $R/tools/overlap_report.py $tr 0.4 > $O/${T}_${name}_overlap_streams$s.txt 2>&1
  if [ $s = 1 ]; then python $R/tools/timeline.py $tr 2 > $O/${T}_${name}_timeline_streams1.txt 2>&1; fi
  echo "steps profiled: $k (+2 warm-up); command: bench.py --steps $k --warmup 2 --profile-steps 0 --streams $s $Q $*" > $O/${T}_${name}_kernel_stats_streams$s.cmd
}
for s in 1 0; do
  prof F $s 4
  prof cfg2 $s 8 $C2
  prof cfg3 $s 4 $C3
  prof cfg5 $s 2 $C5
done
cd $R
# ---- manifest
python - <<PY
import glob, hashlib, json, os, subprocess, time
T, O, SHA = '$T', '$O', '$SHA'
files = sorted(f for f in glob.glob(os.path.join(O, T + '_*')) if not f.endswith('MANIFEST.json'))
man = dict(tag=T, lib_sha16=SHA, lib_sha256=hashlib.sha256(open('keypoints_amd/csrc/libkp_hip.so', 'rb').read()).hexdigest(),
           utc=time.strftime('%Y-%m-%dT%H:%M:%SZ', time.gmtime()),
           gpu=subprocess.run(['bash', '-c', 'rocminfo | grep -m1 "Marketing Name:.*MI" || true'], capture_output=True, text=True).stdout.strip(),
           files={os.path.basename(f): hashlib.sha256(open(f, 'rb').read()).hexdigest()[:16] for f in files})
# the bench line's per-launch time of the dominant family (HIP events, one-stream profile steps) against rocprofv3's of the same
# kernels in the --streams 0 trace: profiled passes clock 2.5-7 % lower (MI355X_MICROARCH.md), so rocprof_over_bench > 1 is expected
import csv, re
def fam_ms(path, pat, steps):
    tot = sum(float(r['TotalDurationNs']) for r in csv.DictReader(open(path)) if re.search(pat, r['Name']))
    return tot / steps / 1e6
man['profile_vs_bench'] = {}
# (the traces hold --steps + 2 warm-up steps: 4 + 2 for model F, 8 + 2 for config 2)
for name, bench, pat, steps in (('F', 'bench_streams0', r'conv3x3_patch_k|conv3x3_mfma_k|conv3x3_pair_k', 6), ('cfg2', 'cfg2_bench', r'conv3x3_small_band_k|conv3x3_small_k', 10)):
    try:
        b = json.load(open(os.path.join(O, f'{T}_{bench}.json')))['roofline']
        ms = fam_ms(os.path.join(O, f'{T}_{name}_kernel_stats_streams0.csv'), pat, steps)
        bms = b['kernels'][b['kernel']]['ms_per_step']
        man['profile_vs_bench'][name] = dict(kernel_family=b['kernel'], bench_family_ms_per_step=bms, rocprof_family_ms_per_step=ms,
                                             rocprof_over_bench=ms / bms, bench_frac=b['frac'], frac_from_rocprof=b['frac'] * bms / ms)
    except Exception as e:
        man['profile_vs_bench'][name] = dict(error=repr(e))
json.dump(man, open(os.path.join(O, T + '_MANIFEST.json'), 'w'), indent=1)
print(json.dumps(man['profile_vs_bench'], indent=1))
for n in ('bench', 'bench_streams0', 'bench_bf16', 'cfg2_bench', 'cfg2_bench_graph', 'cfg3_bench', 'cfg5_bench', 'fp32_bench'):
    try:
        d = json.load(open(os.path.join(O, f'{T}_{n}.json'))); r = d['roofline']
        print(n, round(d['value'], 1), round(d['ms_per_step'], 3), 'frac', round(r['frac'], 3), 'step_hbm', round(r.get('step_hbm_frac') or 0, 3),
              'traffic', r.get('traffic'), 'lib', d['config'].get('lib_sha16'))
    except Exception as e:
        print(n, 'missing', e)
PY
