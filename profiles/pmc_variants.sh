#!/bin/bash
# Wave-level counters of tests/fa_tune variants, four PMC passes (kernel trace only, separate runs):
#   profiles/pmc_variants.sh <variant list, e.g. 0,1> [causal 0|1] [tag]   -> gpurun_out/<tag>_pmc_variants.txt
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${3:-pmc}
OUT=$REPO/gpurun_out/${TAG}_pmc_variants
rm -rf $OUT; mkdir -p $OUT
SETS=("SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
      "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE"
      "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_LDS_DATA_FIFO_FULL GRBM_GUI_ACTIVE"
      "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE")
i=0
for set in "${SETS[@]}"; do
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/k$i -o run -- $REPO/tests/fa_tune 8 16 4096 128 ${2:-0} --only ${1:-0} --rounds 2 > $OUT/k$i.log 2>&1
  i=$((i+1))
done
cd $REPO && python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, collections, sys
csv.field_size_limit(1 << 30)
out_dir, tag = sys.argv[1], sys.argv[2]
res = collections.OrderedDict()
for f in sorted(glob.glob(out_dir + '/*/run_counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'fwd_mfma_kernel' not in n: continue
        key = n[n.index('KernelCfg'):][:110]
        d = res.setdefault(key, collections.defaultdict(list))
        d[r['Counter_Name']].append(float(r['Counter_Value']))
        d['_dur'].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
with open(f'gpurun_out/{tag}_pmc_variants.txt', 'w') as out:
    for k, v in res.items():
        avg = {c: sum(x) / len(x) for c, x in v.items()}
        cyc = avg['GRBM_GUI_ACTIVE'] / 8
        wc = avg.get('SQ_WAVE_CYCLES', 0)
        lines = [k, '  kernel %.1f us under PMC, cycles/launch %.0f, clock %.3f GHz' % (avg['_dur'] / 1e3, cyc, cyc / avg['_dur'])]
        for c in sorted(avg):
            if c in ('GRBM_GUI_ACTIVE', '_dur'): continue
            lines.append('  %-28s %14.0f   per SIMD-cycle %.4f   per wave-cycle %.4f' % (c, avg[c], avg[c] / 1024 / cyc, avg[c] / wc if wc else 0))
        print('\n'.join(lines)); out.write('\n'.join(lines) + '\n')
PY
