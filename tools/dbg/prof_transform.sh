set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
O=$R/gpurun_out/r04l
mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o tr -- python $R/tools/time_transform_rs.py 3 > $O/tr.log 2>&1 || true
tail -2 $O/tr.log
python - <<EOF
import csv
rows=[(r["Kernel_Name"],int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Grid_Size"],r["Workgroup_Size"]) for r in csv.DictReader(open("$(ls $O/trace/*kernel_trace.csv | head -1)"))]
rows.sort(key=lambda r:r[1])
# last build: the last 30 launches
tail=rows[-14:]
t0=tail[0][1]
for n,s,e,g,w in tail:
    print(f"t {(s-t0)/1e3:9.1f} dur {(e-s)/1e3:8.1f} grid {g:>10} wg {w:>5} {n[:100]}")
EOF
rm -rf $O/trace
