export PYTHONPATH=$PWD
rm -rf gpurun_out/stress_p8
python tools/stress_suite.py --loops 8 --parallel 8 --conditions corun,hot --out gpurun_out/stress_p8 > gpurun_out/stress_p8.log 2>&1
tail -3 gpurun_out/stress_p8.log
