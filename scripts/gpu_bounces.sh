R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; cd $R
for b in 1 2 3 4 5 8; do
timeout -k 10 120 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --bounces $b 2>>$OUT/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('B=$b', d['value'], 'ms/step', d['ms_per_step'], 'kernel_ms', r['kernel_ms_per_launch'], 'rays/sample', r['rays_per_sample'], 'nodes/ray', r['nodes_per_ray'])"
done
