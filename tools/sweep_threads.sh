for rep in 1 2 3; do for t in 16 12 10 8; do
  timeout 300 python bench.py --reg-threads $t --no-cpu-baseline --no-pcie --steps 6 --warmup 2 2>/dev/null | tail -1 > /tmp/sweep_$t.json
  python -c "import json; d=json.load(open('/tmp/sweep_$t.json')); print('threads $t', round(d['ms_per_step'],1), round(d['config']['register_ms_per_step'],1), round(d['config']['pairwise_ms_per_step'],1))"
done; done
