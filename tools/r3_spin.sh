# repeated bench runs on one box; usage: bash tools/r3_spin.sh
for a in "" "--spinup-ms 0" "" "--spinup-ms 0" ""; do python bench.py --no-cpu-baseline $a 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('[$a]', d['value'], d['ms_per_step'], d['ms_per_step_reference_lists'], (d.get('roofline') or {}).get('avg_ms'), d.get('spinup'))"; done
