# kernel time of the plain C2 rollout over the number of steps per launch: fixed cost per launch and cost per step
for n in 16 32 64 128 256 512; do
  python bench.py --no-extras --no-cpu-baseline --env-steps $n --steps 200 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('N=$n kernel %.2f us  end-to-end %.2f us per launch' % (d['roofline']['avg_launch_ms']*1e3, d['ms_per_step']*1e3))"
done
