R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
run() { label=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps $STEPS --warmup 4 --no-cpu-baseline --no-extra --kernel restart $BARGS 2>>$OUT/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$label', d['value'], d['ms_per_step'], r['kernel_ms_per_launch'])"
}
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "large_scene or wide_walk or trace_rays or config4" > $OUT/pytest_tmp.log 2>&1; echo "pytest subset rc=$? $(tail -1 $OUT/pytest_tmp.log)"
STEPS=10
for t in 341 512 640; do
  BARGS="--atrium" run "atrium pool=lds treelet=$t" PTAMD_TREELET=$t
done
BARGS="--atrium" run "atrium pool=global treelet=640" PTAMD_POOL_LDS=0
BARGS="--tessellate 24" run "tessellated pool=lds treelet=512" PTAMD_TREELET=512
BARGS="--tessellate 24" run "tessellated pool=global" PTAMD_POOL_LDS=0
STEPS=40
for wm in 3 5 8; do for rm in 8 16 32; do
  run "indoor walk_min=$wm round_min=$rm" PTAMD_WALK_MIN=$wm PTAMD_ROUND_MIN=$rm
done; done
run "indoor walk_min=5 round_min=16 div=2" PTAMD_ROUND_DIV=2
run "indoor walk_min=5 round_min=16 div=8" PTAMD_ROUND_DIV=8
BARGS="--frames-in-flight 3" run "indoor fif=3" X=1
