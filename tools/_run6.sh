cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_etc2.py tests/test_stress_gpu.py -x -q -m gpu 2>&1 | tail -2
for spec in "etc2rgba 4096" "etc2 4096" "etc1 2048" "etc2pt 2048" "etc2rgba 16"; do set -- $spec; python tools/fmt_bench.py $1 $2 3 2>&1 | grep -v amdgpu; done
PROFILE_SKIP_BENCH=1 PROFILE_FORMATS=etc2rgba:4096 bash tools/profile_formats.sh r04e > gpurun_out/prof_e.log 2>&1
python - <<PY
import json
d=json.load(open("gpurun_out/r04e/fmt/fmt_summary.json"))
for e in d.get("etc2rgba",[]): print(e["kernel"][:40], e["dur_us"], e.get("derived"), e.get("hbm"))
PY
