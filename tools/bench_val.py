"""Prints value and ms_per_step of the last JSON line on stdin (helper for A/B runs of bench.py)."""
import json
import sys
d = json.loads(sys.stdin.read().strip().split('\n')[-1])
print(sys.argv[1] if len(sys.argv) > 1 else '', d['value'], d['ms_per_step'])
