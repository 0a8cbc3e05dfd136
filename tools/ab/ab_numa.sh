# runs of episodes dealt to the CUs by the half of the memory their rows lie in (CC4_PERSIST_NUMA=1, probed at first use) vs in slot order (0); one box
for n in 8192 5632 7001; do CC4_PERSIST_NUMA=2 CC4_PERSIST_VERIFY=1 timeout 300 python tools/verify_probe.py $n 1 60 2>&1 | tail -2; done
CC4_PERSIST_NUMA=2 CC4_PERSIST_VERIFY=1 timeout 300 python tools/verify_probe.py 8192 0 40 2>&1 | tail -2
for rep in 1 2 3; do
for b in 0 1; do
  for K in 20 500; do
    CC4_PERSIST_NUMA=$b python bench.py --steps $K --warmup 5 --no-alt --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readlines()[-1]); print('numa=$b K=$K', round(r['value']/1e6,1),'M', r['roofline']['kernel'], 'step_us', round(r['ms_per_step']*1e3,2))"
  done
done
done
CC4_PERSIST_NUMA=1 CC4_PERSIST_TIMELINE=1 timeout 300 python tools/persist_timeline.py 2>&1 | grep "timeline" | head -22 | cut -c1-200
