#!/bin/bash
# developer ablation sweep: bench the headline step with each ablated library
for a in 0 256; do
  if [ $a = 0 ]; then unset PLSR_LIB; else export PLSR_LIB=/root/repo/build_exp/libplsr_a$a.so; fi
  timeout -k 10 120 python bench.py --steps 5 --warmup 2 --no-cpu 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('ablate $a step %.3f boot %.3f perm %.3f'%(d['ms_per_step'], r['avg_launch_ms'], r['perm_kernel']['avg_launch_ms']))
" || exit 1
done
