#!/bin/bash
cd "$(dirname "$0")/bin" || exit 1
for b in "$@"; do timeout -k 5 60 ./$b 128 1052 50 2 || echo "$b failed rc=$?"; done
