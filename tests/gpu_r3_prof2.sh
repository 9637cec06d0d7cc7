#!/bin/bash
# round 3 final-binary profiles, part 2: the other BASELINE configurations
for c in c5 c2 c1; do
  bash tests/gpu_profile.sh r03a_$c --config $c
done
