#!/bin/bash
# MFMA-pipe utilisation and sustained clock of tests/fa_tune variants (PMC pass, kernel trace only):
#   profiles/pmc_tune.sh <variant list, e.g. 0,6,7> [causal 0|1]   -> gpurun_out/pmc_tune/ + pmc_tune_summary.txt
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $REPO/gpurun_out/pmc_tune
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $REPO/gpurun_out/pmc_tune -o run -- $REPO/tests/fa_tune 8 16 4096 128 ${2:-0} --only ${1:-0} --rounds 3 > $REPO/gpurun_out/pmc_tune.log 2>&1
cd $REPO && python3 - <<'PY'
import csv, glob, collections
csv.field_size_limit(1 << 30)
f = glob.glob('gpurun_out/pmc_tune/*counter_collection.csv')[0]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if 'fwd_mfma_kernel' not in r['Kernel_Name']: continue
    a = agg.setdefault(r['Kernel_Name'], collections.defaultdict(list))
    a[r['Counter_Name']].append(float(r['Counter_Value']))
    a['dur'].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
with open('gpurun_out/pmc_tune_summary.txt', 'w') as out:
    for k, v in agg.items():
        n = len(v['GRBM_GUI_ACTIVE']); cyc = sum(v['GRBM_GUI_ACTIVE']) / n / 8; dur = sum(v['dur']) / len(v['dur'])
        line = '%-70s launches %3d  %7.1f us  clock %.3f GHz  MFMA busy %.3f  busy x clock %.3f' % (
            k[k.index('KernelCfg'):][:70], n, dur / 1e3, cyc / dur, sum(v['SQ_VALU_MFMA_BUSY_CYCLES']) / n / 1024 / cyc,
            sum(v['SQ_VALU_MFMA_BUSY_CYCLES']) / n / 1024 / dur)
        print(line); out.write(line + '\n')
PY
