# usage: bash tools/ab.sh <rounds> <size> name1 name2 ...   (libraries ab_libs/<name>.so; alternates them on one box)
R=${1:-3}; SIZE=${2:-1920x1080}; shift; shift
L=cuda-path-tracer-denoising_amd/libsvgf_hip.so
cp $L /tmp/orig.so
mkdir -p gpurun_out/ab
for r in $(seq 1 $R); do for v in "$@"; do
  cp ab_libs/$v.so $L
  echo -n "$v " ; python tools/ab_measure.py $SIZE 2>/dev/null | grep '^AB'
done; done | tee gpurun_out/ab/ab_$(date +%H%M%S).txt
cp /tmp/orig.so $L
