cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04b
timeout 900 python -m pytest tests/test_bc6h.py tests/test_abi.py -x -q -m gpu 2>&1 | tail -5
python tools/bc6h_family_bench.py 2>&1 | grep -v amdgpu.ids
PROFILE_SKIP_BENCH=1 PROFILE_FORMATS=bc6hu:4096 bash tools/profile_formats.sh r04b > gpurun_out/r04b/prof.log 2>&1
python - <<PY
import json
d=json.load(open("gpurun_out/r04b/fmt/fmt_summary.json"))
for e in d.get("bc6hu",[]): print(e["kernel"][:40], e["dur_us"], e.get("derived"), e.get("mem"), e.get("hbm"))
PY
