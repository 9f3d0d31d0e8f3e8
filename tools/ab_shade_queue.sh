#!/bin/bash
# sweep of the shade work queue: main-pass pieces per 12-pixel group (DRT_SHADE_SUBS) x main pieces between two tail items (DRT_TAIL_PERIOD, 0 = default)
for subs in 1 2 3 4 6 12; do for per in 0 1 2; do
echo -n "subs=$subs period=$per "; DRT_SHADE_SUBS=$subs DRT_TAIL_PERIOD=$per timeout -k 10 200 python bench.py --no-cpu-baseline --no-oneshot 2>/dev/null | python -c "
import sys, json
j = json.loads([l for l in sys.stdin if l[0] == chr(123)][0]); print(j['value'], j['roofline']['kernel_ms_per_step'])"
done; done
