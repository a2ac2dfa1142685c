cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/convectionkernels_amd/lib:$LD_LIBRARY_PATH
timeout 600 python -m pytest tests/test_abi.py -x -q -m gpu 2>&1 | tail -3
echo coalesced; convectionkernels_amd/lib/dropin_bench 0.5
echo off; CVTTMI_DROPIN_COALESCE=0 convectionkernels_amd/lib/dropin_bench 0.5
