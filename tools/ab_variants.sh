#!/bin/bash
# Interleaved A/B of library builds: tools/ab_dense.py once per library, the whole set repeated (box drift hits all alike).
#   bash tools/ab_variants.sh <rounds> <variant name> [<variant name> ...]      (variants built by tools/build_variant.py)
set -u
rounds=$1; shift
for r in $(seq $rounds); do
  echo "== round $r"
  python tools/ab_dense.py || exit 1
  for v in "$@"; do
    TGCN_LIB_PATH=pytextgcn_amd/lib/variants/libtgcn_$v.so python tools/ab_dense.py || exit 1
  done
done
