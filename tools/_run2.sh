cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in "" $VARIANTS; do
  if [ -n "$v" ]; then export CVTTMI_LIB=$GRAFT_REPO_ROOT/convectionkernels_amd/lib/variants/libcvtt_mi355x_$v.so; fi
  python tools/bc6h_family_bench.py 2>&1 | grep -v amdgpu.ids
done
