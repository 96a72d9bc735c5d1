#!/bin/bash
# On the GPU box: tools/gpu_profile.sh (bench line, kernel traces, PMC passes) + the summaries that go into profiles/ (kernel statistics, concurrency,
# PMC traffic) + a kernel trace of tools/asm_bench.py. Large rocprofv3 outputs are removed; gpurun_out/r03_zz_* are the files to copy into profiles/.
set -u
cd $GRAFT_REPO_ROOT
bash tools/gpu_profile.sh zz pmc > gpurun_out/zz_profile.log 2>&1
O=gpurun_out/zz
DB3=$(find $O/prof3 -name '*.db' | head -1); DB1=$(find $O/prof1 -name '*.db' | head -1)
python tools/trace_db.py $DB3 --csv gpurun_out/r03_zz_kernel_stats_hg38_3streams.csv > gpurun_out/r03_zz_concurrency_hg38_3streams.txt 2>&1
cp $O/bench.json gpurun_out/r03_zz_bench_line_hg38_profile_run.json
python tools/pmc_traffic.py gpurun_out/r03_zz_pmc_hbm_traffic.json $O/bench_pmc_FETCH_SIZE.json $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_INSTS_VALU > gpurun_out/zz_pmc.log 2>&1
# asm: kernel trace of the 8-contig bench
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/asm -o asm -- python $GRAFT_REPO_ROOT/tools/asm_bench.py --no-oracle > $GRAFT_REPO_ROOT/gpurun_out/r03_zz_asm_bench_under_rocprof.json 2>/dev/null
cd $GRAFT_REPO_ROOT
DBA=$(find $O/asm -name '*.db' | head -1)
python tools/trace_db.py $DBA --csv gpurun_out/r03_zz_asm_kernel_stats.csv --skip 0 > /dev/null 2>&1
rm -rf $O/prof3 $O/prof1 $O/asm $O/pmc_*
ls -la gpurun_out | tail -20
head -12 gpurun_out/r03_zz_asm_kernel_stats.csv
