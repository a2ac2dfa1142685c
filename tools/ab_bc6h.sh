#!/bin/bash
# GPU box: A/B of the BC6H kernel variants kept under convectionkernels_amd/lib/variants (built with `make VARIANT=...`):
# BASELINE config 3 (4096^2 HDR noise, tools/fmt_bench.py) and the three HDR content families, rate and output digest.
#   tools/ab_bc6h.sh r3 bc6w2 bc6w3 ...      ("" = the shipped library, always first)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in "" "$@"; do
  if [ -n "$v" ]; then export CVTTMI_LIB=$GRAFT_REPO_ROOT/convectionkernels_amd/lib/variants/libcvtt_mi355x_$v.so; else unset CVTTMI_LIB; fi
  echo "== ${v:-shipped}"
  python tools/kernel_resources.py ${CVTTMI_LIB:-convectionkernels_amd/lib/libcvtt_mi355x.so} | grep "bc6h_kernel<false, false>"
  python tools/fmt_bench.py bc6hu 4096 2 2>&1 | grep -v amdgpu.ids
  python tools/bc6h_family_bench.py 2>&1 | grep -v amdgpu.ids
done
