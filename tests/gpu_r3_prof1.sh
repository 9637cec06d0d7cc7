#!/bin/bash
# round 3 final-binary profiles, part 1: c3 (default) kernel statistics + bench line + per-launch tables
bash tests/gpu_profile.sh r03a
ls gpurun_out/prof_r03a* | head
