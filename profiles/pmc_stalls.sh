#!/bin/bash
# Stall-attribution counters for the production kernel (tests/fa_tune variant 0) and for the synthetic
# attention instruction mix (tests/micro/simd_mix), three PMC passes each, kernel trace only.
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_stalls
rm -rf $OUT; mkdir -p $OUT
SETS=("SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
      "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE"
      "SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
      "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_IFETCH SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE")
i=0
for set in "${SETS[@]}"; do
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/k$i -o run -- $REPO/tests/fa_tune 8 16 4096 128 ${1:-0} --only 0 --rounds 2 > $OUT/k$i.log 2>&1
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/m$i -o run -- $REPO/tests/micro/simd_mix --ceiling 1500 > $OUT/m$i.log 2>&1
  i=$((i+1))
done
cd $REPO && python3 - <<'PY'
import csv, glob, collections
csv.field_size_limit(1 << 30)
res = collections.OrderedDict()
for f in sorted(glob.glob('gpurun_out/pmc_stalls/*/run_counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'fwd_mfma_kernel' in n: key = 'attention kernel (production)'
        elif 'mix_kernel<16, 16, 16, 8, 8, 16' in n: key = 'synthetic attention mix (random operands)'
        elif 'mix_kernel' in n: key = 'synthetic MFMA only (random operands)'
        else: continue
        res.setdefault(key, collections.defaultdict(list))[r['Counter_Name']].append(float(r['Counter_Value']))
with open('gpurun_out/pmc_stalls_summary.txt', 'w') as out:
    for k, v in res.items():
        avg = {c: sum(x) / len(x) for c, x in v.items()}
        cyc = avg['GRBM_GUI_ACTIVE'] / 8
        wc = avg.get('SQ_WAVE_CYCLES', 0)
        lines = [k, '  cycles/launch %.0f' % cyc]
        for c in sorted(avg):
            if c == 'GRBM_GUI_ACTIVE': continue
            lines.append('  %-28s %14.0f   per SIMD-cycle %.4f   per wave-cycle %.4f' % (c, avg[c], avg[c] / 1024 / cyc, avg[c] / wc if wc else 0))
        print('\n'.join(lines)); out.write('\n'.join(lines) + '\n')
PY
