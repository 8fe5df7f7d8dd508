#!/bin/bash
# A/B on ONE box: the headline bench line of several builds of the library (rays_amd/lib/librays_hip_<name>.so), twice each,
# alternating.   bash tools/variant_bench.sh name1 name2 ... [-- bench.py arguments]
names=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do names+=("$1"); shift; done; [ "$1" == "--" ] && shift
for rep in 1 2; do for n in "${names[@]}"; do
  l=$PWD/rays_amd/lib/librays_hip_$n.so; [ "$n" == "default" ] && l=$PWD/rays_amd/lib/librays_hip.so
  RAYS_HIP_LIB=$l python bench.py --no-cpu-baseline --no-host-entry --no-pipelined "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$n', 'ms_per_step %.4f' % d['ms_per_step'], 'kernel_ms %.4f' % d['roofline']['kernel_ms'], d['config']['kernel'], 'exact %.4f' % d.get('ms_per_step_exact', 0.))"
done; done
