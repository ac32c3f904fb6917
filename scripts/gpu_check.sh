#!/bin/bash
# quick GPU loop: parity tests, then bench lines (both algos optional)
if [ -z "$NOTEST" ]; then python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log; tail -4 gpurun_out/pytest_gpu.log; fi
for ARGS in "$@"; do
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline $ARGS 2>&1 | grep -v amdgpu.ids | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('ARGS[$ARGS] value %.3e'%d['value'],'ms/step %.3f'%d['ms_per_step'],{k:round(v,3) for k,v in d['kernel_ms'].items()},d['mass_conserved'])
    else: print(l)
"
done
