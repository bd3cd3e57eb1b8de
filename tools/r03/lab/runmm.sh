#!/bin/bash
cd "$(dirname "$0")/bin" || exit 1
for b in "$@"; do timeout -k 5 120 ./$b 128 16384 3 || echo "$b failed rc=$?"; done
