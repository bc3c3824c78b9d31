python tools/bench_tracking.py --batch 4096 --frames 60 --check 1 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['value']), d['ms_per_stage'], d['parity_vs_oracle_chain'])"
