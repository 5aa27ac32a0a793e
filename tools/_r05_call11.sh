export PYTHONPATH=$PWD
O=gpurun_out/r05_c11; mkdir -p $O
for i in 1 2; do timeout 200 python bench.py --no-aux --steps 20 --warmup 5 > $O/bench_noaux_$i.json 2> $O/bench_noaux.err; python -c "
import json;d=json.load(open('$O/bench_noaux_$i.json'));print(d['value'],d['ms_per_step'],d['timing']['ms_per_step_by_region'],d['step']['device_ms_per_step'])"; done
