set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
O=$R/gpurun_out/r04final
mkdir -p $O
# 1. the bench line, plain
python $R/bench.py --steps 20 --warmup 5 > $O/bench_final.json 2> $O/bench_final.err
echo "bench done"
# 2. the same command under rocprofv3 --kernel-trace --stats
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $R/bench.py --steps 20 --warmup 5 > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err || true
cp $(ls $O/trace/*kernel_stats.csv | head -1) $O/bench_kernel_stats.csv
python $R/tools/cycle_gaps.py $(ls $O/trace/*kernel_trace.csv | head -1) > $O/cycle_gaps.txt 2>&1 || true
rm -rf $O/trace
echo "rocprof done"
# 3. PMC traffic of the J/K kernels
for N in 148 256 384; do
  K=jk_mx_kernel; if [ $N = 148 ]; then K=jk_m8_kernel; fi
  python $R/tools/time_jk_kernel.py $N > $O/time_jk_kernel_$N.txt 2>&1
  tail -1 $O/time_jk_kernel_$N.txt
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f$N -- python $R/tools/time_jk_kernel.py $N > $O/pmc_f$N.log 2>&1 || true
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w$N -- python $R/tools/time_jk_kernel.py $N > $O/pmc_w$N.log 2>&1 || true
  B=$(python -c "n=$N; m=n*(n+1)//2; print(4*m*(m+1) if n == 148 else 8*m*m)")  # (8-fold unique integrals at N = 148, the 4-fold ones above)
  python $R/tools/pmc_traffic.py $O/f$N $O/w$N $K $B $O/jk_traffic_n$N.json "N_AO=$N whole tensor, two densities, tools/time_jk_kernel.py $N, final code of round 4" | cut -c1-200
  rm -rf $O/f$N $O/w$N $O/pmc_f$N.log $O/pmc_w$N.log
done
