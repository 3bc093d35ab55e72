# round 5: SQ instruction counters of the lane kernel for prebuilt library variants (libsvgf_hip.so.<tag>), and the persistent-launch pricing
# usage: exp_r05_pmc_variants.sh "A C"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
L=$R/cuda-path-tracer-denoising_amd/libsvgf_hip.so
cp $L $L.orig
for v in $1; do
  cp $L.$v $L; touch $L
  rocprofv3 -i $R/tools/pmc2.txt -d /tmp/pmc_v -o p --output-format csv -- python $R/tools/probe.py --variants 4 --frames 6 > /dev/null 2>&1
  echo "== variant $v"
  python $R/tools/pmc_summary.py /tmp/pmc_v "atrous" | grep -E "SQ_INSTS_VALU|SQ_INSTS_SALU|SQ_INSTS_LDS|SQ_INSTS_VMEM_RD"
  rm -rf /tmp/pmc_v
done
cp $L.orig $L
echo "== ubench9 (kernel boundary vs grid barrier)"
timeout 120 $R/tools/ubench9 40 400
timeout 120 $R/tools/ubench9 20 400
