#!/bin/bash
# GPU box: A/B of library variants on the BC7 content families (tools/fmt_bench.py): RGBA noise and opaque noise at 4096^2, photo-like,
# opaque gradients and two colours at SIZE^2 / 16 blocks.   tools/ab_bc7_families.sh 2048 variant ...   ("" = shipped, first)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
SIZE=$1; shift
for v in "" "$@"; do
  if [ -n "$v" ]; then export CVTTMI_LIB=$GRAFT_REPO_ROOT/convectionkernels_amd/lib/variants/libcvtt_mi355x_$v.so; else unset CVTTMI_LIB; fi
  line="${v:-shipped}"
  for f in "bc7 4096" "bc7o 4096" "bc7photo $SIZE" "bc7grad $SIZE" "bc7two $SIZE"; do
    set -- $f
    r=$(python tools/fmt_bench.py $1 $2 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%s %.2f %s' % (d['fmt'], d['mblocks_s'], d['sha'][:6]))")
    line="$line | $r"
  done
  echo "$line"
done
