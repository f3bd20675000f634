set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
O=$R/gpurun_out/r04j
mkdir -p $O
python $R/tools/tts_run.py 148 > $O/tts_plain.txt 2>&1
cat $O/tts_plain.txt | tail -3
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o tts -- python $R/tools/tts_run.py 148 > $O/tts_prof.txt 2>&1 || true
python $R/tools/tts_summary.py $(ls $O/trace/*kernel_trace.csv | head -1) all > $O/tts_summary.txt 2>&1
rm -rf $O/trace
head -3 $O/tts_summary.txt; grep -n "idle" $O/tts_summary.txt; tail -28 $O/tts_summary.txt
