#!/bin/bash
# round-3 pass b: full GPU suite (all failures listed), k_conv_gn2 packed-fp32 normalisation A/B (libnatinf.so = PK 1, libnatinf_pk0.so = PK 0)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3b; mkdir -p $O
cd $R
python3 -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -15 $O/pytest.log
for rep in 1 2; do
  for L in libnatinf.so libnatinf_pk0.so; do
    echo "== $L" | tee -a $O/ab.log
    NATINF_LIB=$R/naturaldiffusion_amd/$L python3 tools/bench_conv_gn.py 2>&1 | tee -a $O/ab.log
    NATINF_LIB=$R/naturaldiffusion_amd/$L python3 tools/ab_knob.py natinf_set_gemm_pref512 1 2>&1 | grep "ms per" | tail -2 | tee -a $O/ab.log
  done
done
