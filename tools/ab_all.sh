#!/bin/bash
# GPU box: A/B of library variants (convectionkernels_amd/lib/variants, `make VARIANT=... EXTRA=...`) over one line of formats:
# rate (Mblocks/s) and output digest per format.   tools/ab_all.sh variant ...      ("" = the shipped library, always first)
# FORMATS="bc7:4096 bc6hu:4096 ..." overrides the list.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
FORMATS=${FORMATS:-"bc7:4096 bc7o:4096 bc7photo:2048 bc7grad:2048 bc6hu:4096 bc6hs:2048 bc1:4096 bc1x:2048 bc3:4096 etc1:2048 etc2:2048 etc2rgba:4096 etc2pt:2048 eac:4096"}
for v in "" "$@"; do
  if [ -n "$v" ]; then export CVTTMI_LIB=$GRAFT_REPO_ROOT/convectionkernels_amd/lib/variants/libcvtt_mi355x_$v.so; else unset CVTTMI_LIB; fi
  line="${v:-shipped}"
  for f in $FORMATS; do
    r=$(python tools/fmt_bench.py ${f%%:*} ${f#*:} 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%s %.2f %s' % (d['fmt'], d['mblocks_s'], d['sha'][:6]))")
    line="$line | $r"
  done
  echo "$line"
done
