#!/bin/bash
# PMC passes over the fit kernels alone (tools/fit_only.py): MFMA busy of the Cholesky kernels, VALU activity of K1
TAG=${1:-r02x}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for C in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $C | tr ' ' '_')
  FIT_REPS=4 timeout 300 rocprofv3 --kernel-trace --pmc $C -d $OUT/fitpmc/$name -o pmc -- python tools/fit_only.py > $OUT/fitpmc_$name.txt 2> $OUT/fitpmc_$name.err
  echo "pmc $C rc=$?" >> $OUT/fitpmc_summary.txt
done
python tools/rocpd_pmc.py $OUT/fitpmc >> $OUT/fitpmc_summary.txt 2>&1
find $OUT/fitpmc -size +30M -delete
cat $OUT/fitpmc_summary.txt
