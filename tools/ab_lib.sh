#!/bin/bash
# same-box A/B of two builds of the library through bench.py:  tools/ab_lib.sh <variant.so> [bench args]
V=$1; shift
for rep in 1 2 3; do
  for lib in default "$V"; do
    if [ "$lib" = default ]; then unset ML2048_HIP_LIB; else export ML2048_HIP_LIB=$lib; fi
    python bench.py --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$lib'.split('/')[-1], '%.3e'%d['value'], '%.3f ms'%d['ms_per_step'], 'kernel %.1f us'%(r['avg_launch_ms']*1e3))"
  done
done
