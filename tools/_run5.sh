cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_etc2.py -x -q -m gpu 2>&1 | tail -3
for f in eac etc2rgba; do python tools/fmt_bench.py $f 16 5 2>&1 | grep -v amdgpu; done
python tools/fmt_bench.py eac 4096 3 2>&1 | grep -v amdgpu
CVTTMI_EAC_SPREAD_MAX=100000000 python tools/fmt_bench.py eac 4096 3 2>&1 | grep -v amdgpu
CVTTMI_EAC_SPREAD_MAX=100000000 python tools/fmt_bench.py eac 1024 3 2>&1 | grep -v amdgpu
CVTTMI_EAC_SPREAD_MAX=0 python tools/fmt_bench.py eac 1024 3 2>&1 | grep -v amdgpu
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/convectionkernels_amd/lib:$LD_LIBRARY_PATH
convectionkernels_amd/lib/dropin_bench 0.5
