# N fresh bench processes on this box, headline path only: value, ms per step, which kind of process, roofline frac, matrix-free SpMV us,
# the device route's setup
# (round 5: profiles/r05_bench_repeats.txt; round 6: r06_bench_repeats.txt — one call of this script per box)
n=${1:-5}
echo "box $(hostname) $(date -u +%H:%M:%S) commit ${OMG_GIT_HEAD:-?}"
for i in $(seq 1 $n); do
  timeout 200 python bench.py --no-cpu --no-plain --no-lex --no-sets --no-dropin --no-config4 --no-config1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f V-cycles/s  %.4f ms  %-4s  frac %.3f  SpMV %.1f us (%.2f)  device setup %.3f s' % (d['value'], d['ms_per_step'], d['config'].get('process_population_which'), d['roofline']['frac'], d['fine_grid_spmv']['avg_launch_us'], d['fine_grid_spmv']['frac'], d['config'].get('setup_device_s') or 0))"
done
