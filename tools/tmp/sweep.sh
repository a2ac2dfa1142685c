for lib in t1 t2 t4; do
if [ -n "$lib" ]; then export CVTTMI_LIB=convectionkernels_amd/lib/variants/libcvtt_mi355x_$lib.so; else unset CVTTMI_LIB; fi
echo "LIB=$lib"
for rep in 1 2; do for size in 2048 4096; do
python bench.py --size $size --steps 20 --warmup 5 --no-cpu --no-extra 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  %5d  %.1f' % ($size, d['value']), end='')"
done; done; echo
python tools/fmt_bench.py bc7o 4096 3 | tail -1 | cut -c60-120
python tools/family_bench.py 2>&1 | grep "alpha 248"
done
