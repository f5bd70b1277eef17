for f in 0.15 0.10 0.05 0.03; do
  NODE_DEFERRED_FRAGILE=$f python bench.py --config 3 --steps 200 --warmup 10 --no-roofline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
fb = d.get('fresh_batches') or {}
print('FRAGILE=$f fixed batch: %.0f images/s, retries %s, miss_events %s, dead %.2f | fresh: %.0f images/s, retries %s, dead %s' % (d['value'], d['config']['retries'], d['config']['miss_events'], d['config']['dead_steps_per_step'], fb.get('value', 0), fb.get('retries'), fb.get('dead_steps_per_step')))"
done
