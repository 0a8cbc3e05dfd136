for rep in 1 2; do
for o in 0 4; do
  for K in 20 500; do
    CC4_PERSIST_ORDER=$o python bench.py --steps $K --warmup 5 --no-alt --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readlines()[-1]); print('order=$o K=$K', round(r['value']/1e6,1),'M', r['roofline']['kernel'])"
  done
done
done
CC4_PERSIST_VERIFY=1 timeout 300 python tools/verify_probe.py 2>&1 | tail -4
