#!/bin/bash
# On the GPU box: tools/gpu_profile.sh <tag> pmc (bench line, kernel traces with the default number of batches in flight and with one, three PMC
# passes) + the summaries that go into profiles/: kernel statistics, concurrency / residency, PMC traffic. Usage: tools/gpu_profile_r4.sh <tag>
set -u
TAG=${1:-r04_zc}
cd $GRAFT_REPO_ROOT
timeout 1500 bash tools/gpu_profile.sh $TAG pmc > gpurun_out/${TAG}_profile.log 2>&1
O=gpurun_out/$TAG
DB3=$(find $O/prof3 -name '*.db' | head -1); DB1=$(find $O/prof1 -name '*.db' | head -1)
python tools/trace_db.py $DB3 --csv gpurun_out/${TAG}_kernel_stats_hg38_5streams.csv --skip 5 > gpurun_out/${TAG}_concurrency_hg38_5streams.txt 2>&1
python tools/trace_db.py $DB1 --csv gpurun_out/${TAG}_kernel_stats_hg38_1stream.csv --skip 1 > gpurun_out/${TAG}_concurrency_hg38_1stream.txt 2>&1
cp $O/bench.json gpurun_out/${TAG}_bench_line_hg38_profile_run.json
python tools/pmc_traffic.py gpurun_out/${TAG}_pmc_hbm_traffic.json $O/bench_pmc_FETCH_SIZE.json $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_INSTS_VALU > gpurun_out/${TAG}_pmc.log 2>&1
rm -rf $O/prof3 $O/prof1 $O/pmc_*
ls -la gpurun_out | grep $TAG
cat gpurun_out/${TAG}_concurrency_hg38_5streams.txt | head -45
tail -3 gpurun_out/${TAG}_pmc.log
