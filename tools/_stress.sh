export PYTHONPATH=$PWD
nproc > gpurun_out/stress_nproc.txt
rm -rf gpurun_out/stress_r04
python tools/stress_suite.py --loops 120 --parallel 4 --budget-s 3250 --out gpurun_out/stress_r04 > gpurun_out/stress_r04.log 2>&1
tail -3 gpurun_out/stress_r04.log
