#!/bin/bash
# On the GPU box: instruction counters (one PMC pass, --kernel-trace only) of the hg38-size asm bench, per kernel — k_chain_linked_win's
# instructions per anchor. Usage (through gpurun): bash tools/pmc_asm.sh
set -u
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_asm -o pmc -- python $GRAFT_REPO_ROOT/tools/asm_bench.py --ref-mb 3100 --contigs 5 --max-mb 2.5 --no-oracle > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/pmc_asm/**/*counter_collection.csv', recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f[0])):
    k = r['Kernel_Name'].split('(')[0]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
for k in ('k_chain_linked_win', 'k_local_seed', 'k_ext_phase'):
    print(k, dict(agg[k]))
PY
rm -rf gpurun_out/pmc_asm
