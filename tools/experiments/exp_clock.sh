# shader clock / power while the a-trous levels run back to back (is the VALU-bound kernel power-throttled?)
( for i in $(seq 1 40); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.25; done ) > gpurun_out/clock_samples.txt &
SM=$!
python bench.py --no-cpu-baseline --steps 20000 --warmup 100 2>/dev/null | tail -1 | cut -c1-200
wait $SM
sort gpurun_out/clock_samples.txt | uniq -c | sort -rn | head -8
# bench under the driver's multi-process launcher, world size 1
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-250
