# the step entry points with a whole-batch caller: one launch on the main stream (CC4_WHOLE_BATCH_STEPS=1, the default) vs a launch per episode group
for w in 0 1 0 1; do
  CC4_WHOLE_BATCH_STEPS=$w python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; r=json.loads(sys.stdin.readlines()[-1]); p=r['policy_in_loop']; print('whole_batch_steps=$w', 'headline', round(r['value']/1e6,1), 'policy_in_loop', round(p['value']/1e6,1), 'grouped', round(p['grouped']['value']/1e6,1))"
  CC4_WHOLE_BATCH_STEPS=$w python tools/host_step_probe.py 8192 200 2>&1 | tail -2
  CC4_WHOLE_BATCH_STEPS=$w python tools/host_step_probe.py 2048 200 2>&1 | tail -2
done
