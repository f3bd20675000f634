set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
O=$R/gpurun_out/r04d
mkdir -p $O
for N in 256 384; do
  python $R/tools/time_jk_kernel.py $N > $O/time_jk_kernel_$N.txt 2>&1
  cat $O/time_jk_kernel_$N.txt | tail -1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f$N -- python $R/tools/time_jk_kernel.py $N > $O/pmc_f$N.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w$N -- python $R/tools/time_jk_kernel.py $N > $O/pmc_w$N.log 2>&1
  B=$(python -c "n=$N; print(8*(n*(n+1)//2)**2)")
  python $R/tools/pmc_traffic.py $O/f$N $O/w$N jk_mx_kernel $B $O/jk_mx_traffic_n$N.json "N_AO=$N whole tensor, two densities, tools/time_jk_kernel.py $N"
  rm -rf $O/f$N $O/w$N
done
