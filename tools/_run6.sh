cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_etc2.py -x -q -m gpu 2>&1 | tail -2
for spec in "etc2rgba 4096" "etc2 4096" "etc1 2048" "etc2pt 2048"; do set -- $spec; python tools/fmt_bench.py $1 $2 3 2>&1 | grep -v amdgpu; done
