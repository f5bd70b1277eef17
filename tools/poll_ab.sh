#!/bin/bash
# read-back by a device-written pinned record + host spin (NODE_TUNE_POLL_READBACK=1, default) against hipMemcpyAsync + hipStreamSynchronize (=0):
# the drop-in region of the bench line (a read-back per solve), two rounds each
cd $GRAFT_REPO_ROOT
O=gpurun_out/poll_ab.txt
rm -f $O
timeout 600 python -m pytest tests/test_gpu_solve.py tests/test_gpu_golden.py -x -q 2>&1 | tail -2
for round in 1 2; do
  for v in 1 0; do
    NODE_TUNE_POLL_READBACK=$v timeout 300 python bench.py --steps 30 --warmup 5 --no-roofline --no-cpu-baseline --no-fresh > /tmp/b.json 2>/tmp/b.err || tail -3 /tmp/b.err
    python - "$v" <<'PY' >> $O
import json, sys
d = json.loads(open('/tmp/b.json').read().strip().splitlines()[-1])
print('NODE_TUNE_POLL_READBACK=%s  deferred %.0f images/s (%.3f ms/step)   drop-in %.0f images/s (%.3f ms/step)   ratio %.3f'
      % (sys.argv[1], d['value'], d['ms_per_step'], d['dropin']['value'], d['dropin']['ms_per_step'], d['dropin']['value'] / d['value']))
PY
  done
done
cat $O
for v in 1 0; do echo "== NODE_TUNE_POLL_READBACK=$v"; NODE_TUNE_POLL_READBACK=$v timeout 300 python tools/dropin_time.py 2>&1 | grep -v amdgpu.ids | tail -9; done >> $O
tail -20 $O
