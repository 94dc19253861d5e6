#!/bin/bash
# Builds the in-process stack sampler and runs tools/host_profile.py with it:  gpurun -- 'bash tools/sample_stacks.sh <tag>'
set -u
TAG=${1:-stacks}; R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT $R/tools/_bin
gcc -O1 -shared -fPIC -o $R/tools/_bin/libstack_sampler.so $R/tools/_src/stack_sampler.c -lpthread -ldl
SR_STACKS_OUT=$OUT/stacks_raw.txt timeout 300 python $R/tools/host_profile.py 12 > $OUT/host.txt 2>&1
gzip -f $OUT/stacks_raw.txt
