export TMPDIR=/tmp; cd /tmp
R=$GRAFT_REPO_ROOT
for spec in "9 524288 128 1152 0 9 32" "2 131072 256 2304 0 9 16" "4 524288 128 1152 0 9 32"; do
  tag=$(echo $spec | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc2/${tag}_sq -- python3 $R/tools/bench_gemm.py one $spec 3 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_VALU SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT --output-format csv -d $R/gpurun_out/pmc2/${tag}_sq2 -- python3 $R/tools/bench_gemm.py one $spec 3 > /dev/null 2>&1
done
rocprofv3 -L 2>/dev/null | grep -i "SQ_INSTS_MFMA\|SQ_INST_CYCLES_VMEM\|SQ_WAIT_INST\|SQ_VALU_MFMA\|SQ_INSTS_VALU_MFMA\|TCP_" | head -40
