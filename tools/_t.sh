#!/bin/bash
timeout 600 python tools/fuzz_seq.py 150 11 2>&1 | tail -4
timeout 600 python tools/fuzz_seq.py 150 13 2>&1 | tail -4
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline_fresh']['vs_resident_ms_per_step'], d['verified'])"
