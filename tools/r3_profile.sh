#!/bin/bash
# round-3 evidence pass (GPU box): kernel stats + HBM traffic of the bench command, per-shape GEMM table, SQ counters of k_conv_gn2, SD3 stats
R=$GRAFT_REPO_ROOT; TAG=${1:-r03a}
cd $R
bash tools/profile_round.sh $TAG
bash tools/trace_fwd.sh 0 1 > gpurun_out/profile_${TAG}.by_shape.txt 2>&1
bash tools/pmc_conv_gn.sh "32 512 128 128 0" > gpurun_out/profile_${TAG}.pmc_a.log 2>&1
bash tools/pmc_conv_gn.sh "32 512 256 128 0" > gpurun_out/profile_${TAG}.pmc_b.log 2>&1
bash tools/pmc_conv_gn.sh "16 512 256 256 0" > gpurun_out/profile_${TAG}.pmc_c.log 2>&1
bash tools/profile_sd3.sh ${TAG}_sd3 > gpurun_out/profile_${TAG}_sd3.log 2>&1
bash tools/profile_sd3.sh ${TAG}_sd3_fp8 --fp8 > gpurun_out/profile_${TAG}_sd3_fp8.log 2>&1
tail -3 gpurun_out/profile_${TAG}.by_shape.txt; tail -2 gpurun_out/profile_${TAG}.pmc_a.log
