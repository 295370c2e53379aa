R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
run() { label=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra --kernel restart $BARGS 2>>$OUT/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$label', d['value'], d['ms_per_step'])"
}
for v in "t512=" "t1024=-DPT_RS4_THREADS=1024" "t256=-DPT_RS4_THREADS=256"; do
  name=${v%%=*}; flags=${v#*=}
  make -s -B lib EXTRA_HIPFLAGS="$flags" 2>>$OUT/flags.err || { echo "$name: build failed"; continue; }
  for t in 341 700 1000; do
    BARGS="--atrium" run "$name atrium treelet=$t" PTAMD_TREELET=$t
  done
  BARGS="--tessellate 24" run "$name tessellated treelet=341" PTAMD_TREELET=341
  BARGS="--tessellate 24" run "$name tessellated treelet=700" PTAMD_TREELET=700
done
make -s -B lib 2>>$OUT/flags.err
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "large_scene or wide_walk or trace_rays" > $OUT/pytest_tmp.log 2>&1; echo "pytest subset rc=$? $(tail -1 $OUT/pytest_tmp.log)"
