timeout -k 10 1000 python bench.py --no-cpu-baseline > gpurun_out/bench_probe.json 2> gpurun_out/bench_probe.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_probe.json'))
print(d['value'], d['ms_per_step'])
print(json.dumps(d['roofline_mfma'], indent=1)[:1800])
PY
