#!/bin/bash
# does it matter on which socket the bench's host threads run? (two NUMA nodes, the GPU hangs on one of them; the process may use all 256 CPUs)
cd $GRAFT_REPO_ROOT
BDF=$(python -c "
import torch; p=torch.cuda.get_device_properties(0); print('%04x:%02x:%02x.0' % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id))" 2>/dev/null)
NODE=$(cat /sys/bus/pci/devices/$BDF/numa_node); LOCAL=$(cat /sys/bus/pci/devices/$BDF/local_cpulist)
if [ "$NODE" = 0 ]; then REMOTE=$(cat /sys/devices/system/node/node1/cpulist); else REMOTE=$(cat /sys/devices/system/node/node0/cpulist); fi
echo "GPU $BDF on NUMA node $NODE, local cpus $LOCAL, remote $REMOTE"
run() { "${@:2}" python bench.py --steps 20 --warmup 5 --cpu-sample 0 --verify 0 --no-host-input --extra-configs "" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('  $1:', round(d['ms_per_step'],3), 'ms', round(d['value'],3), 'Gbp/s, seed', round(d['stage_ms_per_step'][0],1), 'host active', round(d['host_active_ms_per_batch'],2))"; }
for rep in 1 2 3; do run free env; run local taskset -c $LOCAL; run remote taskset -c $REMOTE; done
