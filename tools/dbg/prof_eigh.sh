set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
O=$R/gpurun_out/r04g
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o eigh -- python $R/tools/dbg/eigh_grid_test.py $@ > $O/prof_eigh.log 2>&1 || true
tail -5 $O/prof_eigh.log
cp $(ls $O/trace/*kernel_stats.csv | head -1) $O/eigh_kernel_stats.csv
rm -rf $O/trace
head -25 $O/eigh_kernel_stats.csv | cut -c1-200
