#!/bin/bash
# GPU box: A/B of library variants (convectionkernels_amd/lib/variants, `make VARIANT=...`) on one format of tools/fmt_bench.py.
#   tools/ab_fmt.sh etc2rgba 4096 etcw4 ...      ("" = the shipped library, always first)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
FMT=$1; SIZE=$2; shift 2
for v in "" "$@"; do
  if [ -n "$v" ]; then export CVTTMI_LIB=$GRAFT_REPO_ROOT/convectionkernels_amd/lib/variants/libcvtt_mi355x_$v.so; else unset CVTTMI_LIB; fi
  echo "== ${v:-shipped}"
  python tools/fmt_bench.py $FMT $SIZE 5 2>&1 | grep -v amdgpu.ids
done
