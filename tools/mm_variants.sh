#!/bin/bash
# dev: build libmile_hip.so with several k_mm3 tuning macro sets ON THE GPU BOX and time the B4 gradient with each
cd "$(dirname "$0")/.."
for v in "$@"; do
  echo "== $v"
  MILE_HIPCC_FLAGS="$v" python -c "from mile_amd._build import build_library; build_library(force=True)" || exit 1
  timeout -k 10 300 python tools/b4_time.py auto 2>&1 | tail -1
done
