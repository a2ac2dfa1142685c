cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
echo shipped; for i in 1 2; do python tools/fmt_bench.py bc7 4096 5 2>/dev/null | cut -c1-120; done
export CVTTMI_LIB=$GRAFT_REPO_ROOT/convectionkernels_amd/lib/variants/libcvtt_mi355x_persist.so
for g in 0 4096 8192 16384; do echo persist grid $g; CVTTMI_BC7_GRID=$g python tools/fmt_bench.py bc7 4096 5 2>/dev/null | cut -c1-120;  CVTTMI_BC7_GRID=$g python tools/fmt_bench.py bc7photo 2048 3 2>/dev/null | cut -c1-120; done
