#!/bin/bash
# leftover-block cost: the same binary at N = 1024 (32 blocks: four full rounds per workgroup) and N = 1052 (33 blocks)
cd "$(dirname "$0")/bin" || exit 1
for b in "$@"; do for n in 1024 1052 1088 1120; do timeout -k 5 60 ./$b 128 $n 50 2 || echo "$b failed rc=$?"; done; done
