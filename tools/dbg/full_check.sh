cd $GRAFT_REPO_ROOT
O=gpurun_out/r04h
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "pytest rc=$?" | tee -a $O/gputest.log; tail -3 $O/gputest.log
NBED_BENCH_REHEARSE=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 6 --warmup 3 --no-real --no-small --no-df > $O/bench_2rank_rehearsal.json 2> $O/bench_2rank.err; echo "rehearsal rc=$?"; tail -c 400 $O/bench_2rank.err
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 300 $O/bench.err
