for t in 1 2 3 4 6 8 12 16; do
  timeout 300 python bench.py --reg-threads $t --no-cpu-baseline --no-pcie --steps 4 --warmup 1 2>/dev/null | tail -1 > /tmp/sweep_$t.json
  python -c "import json; d=json.load(open('/tmp/sweep_$t.json')); print('threads $t', round(d['ms_per_step'],1), round(d['config']['register_ms_per_step'],1), round(d['config']['pairwise_ms_per_step'],1))"
done
