# the driver's short run with exactly W warm-up steps (--min-warmup-ms 0) or extended to 30 ms (default), alternating
for rep in $(seq 1 ${REPS:-6}); do for mw in 0 30; do
  python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 --min-warmup-ms $mw 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('min-warmup $mw', '%.4g ms/step %.4f kernel %.2f warm-up steps %d' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']*1e3, d['warmup_steps_run']))"
done; done
