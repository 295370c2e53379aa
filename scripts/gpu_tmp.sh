R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; cd $R
timeout -k 10 500 python scripts/band_proxy.py --ranks 8 4 2 --in-flight 1 3 --interleave 8 --out $OUT/band_proxy_interleaved8.json; echo "proxy8 rc=$?"
timeout -k 10 300 python scripts/band_proxy.py --ranks 8 --in-flight 3 --interleave 0 --out $OUT/band_proxy_contiguous.json; echo "proxy contiguous rc=$?"
timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('1 GPU', d['value'], d['ms_per_step'])"
