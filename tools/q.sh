q() { python bench.py --no-cpu --no-plain --no-lex --no-sets --no-dropin --no-config4 --no-config1 $2 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['level0_kernels']; print('$1', d['value'], d['ms_per_step'], 'down', k['plane_down']['avg_us'], 'up', k['plane_up']['avg_us'])"; }
