#!/bin/bash
# On the GPU box: the round-6 profile set of ONE bench workload (--config): bench line, rocprofv3 kernel traces with the default number of batches in flight
# and with one, the three PMC passes (separate passes, --kernel-trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes), and the summaries that go
# into profiles/: kernel statistics + concurrency (tools/trace_db.py), PMC traffic / VALU per kernel (tools/pmc_traffic.py, keyed by the workload id so that
# bench.py finds them for that config). Usage: tools/gpu_profile_r6.sh <tag> <config> [steps]      e.g.  tools/gpu_profile_r6.sh r05_hifi hifi_hg38
set -u
TAG=$1; CFG=$2; STEPS=${3:-15}
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
B="--config $CFG --cpu-sample 0 --verify 0 --extra-configs \"\" --no-host-input --steps $STEPS"
eval timeout 600 python bench.py --config $CFG --cpu-sample 0 --verify 16 --extra-configs '""' --steps $STEPS > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
eval timeout 900 rocprofv3 --kernel-trace -d $O/prof3 -o p3 -- python $GRAFT_REPO_ROOT/bench.py $B > $O/bench_prof3.json 2>/dev/null
eval timeout 900 rocprofv3 --kernel-trace -d $O/prof1 -o p1 -- python $GRAFT_REPO_ROOT/bench.py $B --streams 1 --steps 8 > $O/bench_prof1.json 2>/dev/null
for C in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU GRBM_GUI_ACTIVE"; do
  N=$(echo $C | cut -d' ' -f1)
  eval timeout 1200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$N -o pmc -- python $GRAFT_REPO_ROOT/bench.py $B > $O/bench_pmc_$N.json 2>/dev/null
  find $O/pmc_$N -name '*kernel_trace.csv' -delete
done
cd $GRAFT_REPO_ROOT
DB3=$(find $O/prof3 -name '*.db' | head -1); DB1=$(find $O/prof1 -name '*.db' | head -1)
python tools/trace_db.py $DB3 --csv gpurun_out/${TAG}_kernel_stats_5streams.csv --timed $STEPS > gpurun_out/${TAG}_concurrency_5streams.txt 2>&1
python tools/trace_db.py $DB1 --csv gpurun_out/${TAG}_kernel_stats_1stream.csv --timed 8 > gpurun_out/${TAG}_concurrency_1stream.txt 2>&1
cp $O/bench.json gpurun_out/${TAG}_bench_line.json
python tools/pmc_traffic.py gpurun_out/${TAG}_pmc_hbm_traffic.json $O/bench_pmc_FETCH_SIZE.json $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_INSTS_VALU > gpurun_out/${TAG}_pmc.log 2>&1
rm -rf $O/prof3 $O/prof1 $O/pmc_*
head -30 gpurun_out/${TAG}_concurrency_5streams.txt; tail -3 gpurun_out/${TAG}_pmc.log
python - <<P
import json
d = json.loads([l for l in open('$O/bench.json') if l.startswith('{')][-1])
print('$CFG', 'value', round(d['value'], 3), 'ms/step', round(d['ms_per_step'], 2), 'crosscheck', d['oracle_crosscheck'], 'failed', d['failed_reads'], 'host_input', (d.get('host_input') or {}).get('aligned_Gbp_per_s'))
P
