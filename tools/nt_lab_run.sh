#!/bin/bash
# One GPU call of tools/nt_lab.hip experiments: gpurun -- 'bash tools/nt_lab_run.sh r06'
TAG=${1:-r06}; R=$PWD; OUT=$R/gpurun_out/${TAG}_lab; mkdir -p $OUT
L=$R/tools/_bin/nt_lab
[ -x $L ] || hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-function -o $L $R/tools/nt_lab.hip
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1
{
echo "== K sweep, relu, two workgroups per CU"
for K in 128 256 512 1024 2048; do $L 262144 512 $K 2 0 0 0 0 10; done
echo "== softplus / none at K=512"
$L 262144 512 512 1 0 0 0 0 10; $L 262144 512 512 0 0 0 0 0 10
echo "== one workgroup per CU (100 KB of LDS)"
for K in 512 2048; do $L 262144 512 $K 2 0 0 0 100000 10; done
echo "== staggered first wave (mode 1 = wave_id parity, 2 = tg_id parity)"
for m in 1 2; do for f in 0.25 0.5 0.75; do $L 262144 512 512 2 $m $f 0 0 10; done; done
echo "== resident grid of 512, plain and staggered"
$L 262144 512 512 2 0 0 512 0 10
for m in 1 2; do $L 262144 512 512 2 $m 0.5 512 0 10; done
echo "== 86016 rows (template batch)"
$L 86016 512 512 2 0 0 0 0 20; $L 86016 512 512 2 1 0.5 0 0 20; $L 86016 512 512 2 2 0.5 0 0 20
echo "== dumps"
$L 262144 512 512 2 0 0 0 0 5 $OUT/stamps_base.csv
$L 262144 512 512 2 1 0.5 0 0 5 $OUT/stamps_stag1.csv
$L 262144 512 512 2 2 0.5 0 0 5 $OUT/stamps_stag2.csv
$L 262144 512 512 2 0 0 512 0 5 $OUT/stamps_resident.csv
$L 262144 512 512 1 0 0 0 0 5 $OUT/stamps_softplus.csv
} 2>&1 | tee $OUT/lab.txt
for f in base stag1 stag2 resident softplus; do echo "== $f"; python $R/tools/nt_lab_report.py $OUT/stamps_$f.csv; done 2>&1 | tee $OUT/report.txt
# counters of the plain launch
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_$i -- $L 262144 512 512 2 0 0 0 0 3 > $OUT/pmc_$i.log 2>&1
done
ls $OUT
