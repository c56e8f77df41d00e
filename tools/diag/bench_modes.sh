# the three bench modes on one GPU (numbers quoted in DESIGN.md / README.md)
O=gpurun_out/${1:-modes}; mkdir -p $O
timeout -k 10 500 python bench.py --steps 3 --warmup 1 > $O/bench_match.json 2> $O/bench_match.err || exit 1
timeout -k 10 300 python bench.py --mode sharded --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_sharded.json 2> $O/bench_sharded.err || exit 1
timeout -k 10 300 python bench.py --mode sharded --ripple-combine --steps 1 --warmup 0 --no-cpu-baseline > $O/bench_sharded_ripple.json 2> $O/bench_sharded_ripple.err || exit 1
timeout -k 10 300 python bench.py --mode sharded --slots 128 --logical-ranks 8 --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_sharded128.json 2> $O/bench_sharded128.err || exit 1
timeout -k 10 400 python bench.py --mode identify --matches 16 --group 8 --steps 1 --warmup 0 --no-cpu-baseline > $O/bench_identify.json 2> $O/bench_identify.err || exit 1
echo MODES-DONE
