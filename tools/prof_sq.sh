# SQ counters of the k32 conv / wgrad kernels on shapes of the 16 x 512^2 step (two --pmc passes each, counters in their own runs)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_sq
rm -rf $O; mkdir -p $O
cd $R
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
P2="SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
for shape in "128 128 256" "64 64 512"; do
  tag=$(echo $shape | tr ' ' '_')
  for op in conv wgrad; do
    rocprofv3 --kernel-trace --pmc $P1 --output-format csv -d $O/${op}_${tag}_p1 -o p -- python3 tools/micro_split_one.py $shape 1 $op 1 > $O/log_${op}_${tag}_p1.txt 2>&1
    rocprofv3 --kernel-trace --pmc $P2 --output-format csv -d $O/${op}_${tag}_p2 -o p -- python3 tools/micro_split_one.py $shape 1 $op 1 > $O/log_${op}_${tag}_p2.txt 2>&1
  done
  python3 tools/sq_counters.py $O/sq_${tag}.csv "16 images, $shape (Cin Cout size)" $O/conv_${tag}_p1 $O/conv_${tag}_p2 $O/wgrad_${tag}_p1 $O/wgrad_${tag}_p2 -- conv_halo_k32_kernel wgrad_k32_kernel
done
rm -rf $O/*_p1 $O/*_p2
ls $O
