#!/bin/bash
# A/B the whole encoder on ONE box: scripts/ab.sh "ENV1=.. ENV2=.." "ENV..." ...   (each arg = one configuration)
for rep in 1 2; do
  for cfg in "$@"; do
    env $cfg python bench.py --steps 50 --warmup 3 --no-cpu-baseline --no-parity-live --no-exact-line | python scripts/ab_fmt.py "$cfg"
  done
done
