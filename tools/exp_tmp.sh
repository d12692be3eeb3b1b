timeout -k 10 1000 python bench.py --no-cpu-baseline > gpurun_out/bench_bp.json 2> gpurun_out/bench_bp.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_bp.json'))
b=d['bipartite_pp']; print(b.get('value'), b.get('ms_per_step'), b.get('c_biplan'), b.get('error'))
PY
