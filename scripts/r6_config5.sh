#!/bin/bash
# round 6: the whole config-5 batch on one GPU with per-slice records; the box's NUMA topology; 2-rank farm test on one GPU
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{ echo "== numa =="; lscpu | grep -i -E "numa|socket|^CPU\(s\)|model name"; ls /sys/devices/system/node/ | head; for d in /sys/class/drm/card*/device; do echo "$d $(cat $d/numa_node 2>/dev/null)"; done; nproc; cat /sys/fs/cgroup/cpu.max; } > gpurun_out/r6_numa.txt 2>&1
python -m pytest tests/test_farm_gpu.py -q -m gpu 2>&1 | tail -5 > gpurun_out/r6_farm_test.log
python bench.py --config 5 --farm-slices ${1:-512} --farm-record gpurun_out/r6_config5_512slices_1gpu.json > gpurun_out/r6_config5_stdout.txt 2> gpurun_out/r6_config5_stderr.txt
python scripts/makespan_sim.py gpurun_out/r6_config5_512slices_1gpu.json gpurun_out/r6_config5_makespan.json > gpurun_out/r6_makespan.txt 2>&1
cat gpurun_out/r6_numa.txt gpurun_out/r6_farm_test.log gpurun_out/r6_makespan.txt; tail -3 gpurun_out/r6_config5_stderr.txt
