#!/usr/bin/env python
"""Single-episode wrapper surface: microseconds per BlueFlatWrapper.step (bench.py host_api_rates)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
print(json.dumps(bench.host_api_rates(eval_eps=1), indent=1))
