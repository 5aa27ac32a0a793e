export PYTHONPATH=$PWD
timeout 300 python tools/region_probe.py 2>&1 | tail -16
