#!/bin/bash
# round 3, batch 23: HBM traffic (PMC, separate passes) of the final-tree grad kernels: B2 and B3
bash tools/pmc_traffic.sh b23_b2 --no-secondary > gpurun_out/b23_b2_traffic.txt 2>&1; tail -12 gpurun_out/b23_b2_traffic.txt
bash tools/pmc_traffic.sh b23_b3 --workload B3 > gpurun_out/b23_b3_traffic.txt 2>&1; tail -12 gpurun_out/b23_b3_traffic.txt
