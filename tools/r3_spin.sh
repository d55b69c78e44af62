for sp in 0 400 2000 0 400; do python bench.py --no-cpu-baseline --spinup-ms $sp 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('spinup', $sp, d['value'], d['ms_per_step'], d['ms_per_step_reference_lists'], [(k['kernel'],k['avg_ms']) for k in d['roofline_kernels']][:3], d['roofline']['avg_ms'])"; done
