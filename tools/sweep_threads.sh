for t in 8 10 12 8 10 12; do
  timeout 300 python bench.py --reg-threads $t --no-cpu-baseline --steps 4 2>&1 | tail -1 > gpurun_out/sweep_$t.json
  python -c "import json; d=json.load(open('gpurun_out/sweep_$t.json')); print('threads $t', d['value'], d['ms_per_step'], d['config']['register_ms_per_step'])"
done
