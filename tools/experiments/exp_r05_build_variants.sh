# round 5: build libsvgf_hip.so.<tag> for a list of "tag=flags" pairs (flags separated by ';'), here in the container
# usage: exp_r05_build_variants.sh "A=-DSVGF_LANE_SKEW=0" "B=-DSVGF_LANE_SKEW=1;-DSVGF_LANE_PRIO=1,1,2,1,1,2,2"
L=cuda-path-tracer-denoising_amd/libsvgf_hip.so
E=cuda-path-tracer-denoising_amd/libsvgf_hip_exp.so      # -D switches only exist in the experiments build (build.py)
for spec in "$@"; do
  tag=${spec%%=*}; flags=${spec#*=}; flags=${flags//;/ }
  SVGF_EXTRA_HIPCC_FLAGS="$flags" python -c "
import __graft_entry__ as g
pkg=g.load_package(); pkg.build.build_hip(force=True, experiments=True)" || exit 1
  cp $E $L.$tag; echo "built $tag: $flags"
done

python -c "
import __graft_entry__ as g
pkg=g.load_package(); pkg.build.build_hip(force=True, experiments=True)"      # leave the plain experiments build behind
