#!/bin/bash
# Native stacks of the training process at random moments (where does the host thread sit while the GPU works?):
#   gpurun -- 'bash tools/sample_host_stacks.sh <tag> [samples]'
set -u
TAG=${1:-stacks}; NS=${2:-12}
R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
SR_HOST_POLL=${SR_HOST_POLL:-1} python $R/tools/host_profile.py 600 > $OUT/run.txt 2>&1 &
PID=$!
sleep 25          # import + scene + warm-up
for i in $(seq 1 $NS); do
  timeout 20 rocgdb -p $PID -batch -ex "thread apply all bt 14" > $OUT/bt_$i.txt 2>&1
  sleep 0.37
done
kill $PID
python - $OUT $NS <<'PY'
import sys, re, collections
out, ns = sys.argv[1], int(sys.argv[2])
for i in range(1, ns + 1):
    txt = open(f'{out}/bt_{i}.txt', errors='replace').read()
    th = re.split(r'\nThread \d+ ', txt)
    print(f'--- sample {i}: {len(th) - 1} threads')
    for t in th[1:]:
        fr = re.findall(r'#\d+\s+(?:0x[0-9a-f]+ in )?([^\s(]+)', t)
        if any('Py' in f or 'hip' in f.lower() or 'hsa' in f.lower() for f in fr[:14]):
            print('   ', ' < '.join(fr[:10]))
PY
