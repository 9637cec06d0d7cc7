#!/bin/bash
# kernel statistics + bench line of every BASELINE configuration, then the c3 PMC passes: bash tests/gpu_profile_all.sh <tag>
TAG=${1:-run}
bash tests/gpu_profile.sh $TAG
for c in c5 c2 c1; do bash tests/gpu_profile.sh ${TAG}_$c --config $c; done
