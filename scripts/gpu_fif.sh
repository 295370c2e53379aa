R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; cd $R
for h in 1080 135; do for f in 1 2 3; do
timeout -k 10 120 python bench.py --steps 60 --warmup 6 --no-cpu-baseline --height $h --frames-in-flight $f 2>>$OUT/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('h=$h fif=$f', d['value'], d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms_per_launch'])"
done; done
