export PYTHONPATH=$PWD
O=gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -8 > $O/gputest_r04_final.log
python __graft_entry__.py --smoke > $O/smoke_r04.log 2>&1; echo "smoke rc $?" >> $O/smoke_r04.log
bash tools/collect_profiles.sh r04 > $O/collect_r04.log 2>&1
python bench.py --steps 20 --warmup 5 > $O/bench_r04_driver.json 2> $O/bench_r04_driver.err
