mkdir -p gpurun_out/r06_dyn
python tools/r06_dyn2.py > gpurun_out/r06_dyn/dyn2.txt 2>&1; cat gpurun_out/r06_dyn/dyn2.txt
python tools/r06_dyn2.py --prof > gpurun_out/r06_dyn/dyn2_prof.txt 2>&1; cat gpurun_out/r06_dyn/dyn2_prof.txt
