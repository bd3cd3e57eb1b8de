#!/bin/bash
# runs every lab binary once (each a few seconds)
cd "$(dirname "$0")/bin" || exit 1
for b in "$@"; do timeout -k 5 60 ./$b 512 36000 10 || echo "$b failed rc=$?"; done
