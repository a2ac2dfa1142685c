import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for v in ("", "s4", "", "s4"):
    lib = os.path.join(ROOT, "convectionkernels_amd/lib/" + ("variants/libcvtt_mi355x_%s.so" % v if v else "libcvtt_mi355x.so"))
    env = dict(os.environ, CVTTMI_LIB=lib)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fmt_bench.py"), "bc7b", "4096", "5"], env=env, capture_output=True, text=True)
    print(v or "default", p.stdout.strip()[-100:], flush=True)
