#!/bin/bash
O=gpurun_out/r06v; mkdir -p $O
for n in 4 8; do
  timeout 900 bash tools/multirank_one_gpu.sh $n --steps 5 --warmup 2 --no-cpu-baseline --sustain-seconds 0.5 > $O/multirank_$n.json 2> $O/multirank_$n.err
  echo "ranks $n rc=$?"; tail -c 1500 $O/multirank_$n.json; echo; tail -5 $O/multirank_$n.err
done
