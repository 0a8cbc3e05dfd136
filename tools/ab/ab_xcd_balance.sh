# XCD share balancing of the persistent kernel on / off (CC4_PERSIST_BALANCE), one box
for rep in 1 2; do
for b in 0 1; do
  for K in 20 500; do
    CC4_PERSIST_BALANCE=$b python bench.py --steps $K --warmup 5 --no-alt --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readlines()[-1]); print('balance=$b K=$K', round(r['value']/1e6,1),'M', r['roofline']['kernel'], 'step_us', round(r['ms_per_step']*1e3,2))"
  done
done
done
CC4_PERSIST_BALANCE=2 CC4_PERSIST_TIMELINE=1 timeout 300 python tools/persist_timeline.py 2>&1 | grep "timeline\|balance" | cut -c1-300
CC4_PERSIST_VERIFY=1 timeout 300 python tools/verify_probe.py 2>&1 | tail -4
