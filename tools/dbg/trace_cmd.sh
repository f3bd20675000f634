set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
O=$R/gpurun_out/r04e
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-real --no-n2000 --no-small --no-scaling --no-transform --no-cpu-baseline --no-tts --no-mu --jk-event-every 1000 > $O/bench_under_rocprof.json 2> $O/bench.err || true
ls $O/trace
python $R/tools/cycle_gaps.py $(ls $O/trace/*kernel_trace.csv | head -1) > $O/cycle_gaps.txt 2>&1 || true
cp $(ls $O/trace/*kernel_stats.csv | head -1) $O/bench_kernel_stats.csv
rm -rf $O/trace
tail -60 $O/cycle_gaps.txt
