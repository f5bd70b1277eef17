cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04r
for g in "" "--graphs"; do
python bench.py --steps 30 --warmup 5 --no-pmc --no-cpu-baseline --no-fresh --no-dropin --no-roofline $g > gpurun_out/r04r/bench$g.json 2>gpurun_out/r04r/err$g.txt
python -c "
import json
d=json.loads([l for l in open('gpurun_out/r04r/bench$g.json') if l.startswith('{')][-1])
print('graphs=[$g] cfg2', round(d['value']), d['ms_per_step'], d['step_ms']['median'])" || tail -5 gpurun_out/r04r/err$g.txt
done
