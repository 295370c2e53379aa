R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; cd $R
for spp in 1 2 4 8 16 32; do
timeout -k 10 120 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --spp $spp 2>>$OUT/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('spp=$spp', d['value'], 'ms/step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms_per_launch'])"
done
