#!/bin/bash
# per-kernel averages (rocprofv3 --kernel-trace --stats) of the tree's library and of variants: m8_prof_ab.sh NAME...
export PYTHONPATH=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in "" $@; do
  unset NBX_LIB; if [ -n "$v" ]; then export NBX_LIB=$GRAFT_REPO_ROOT/build/variants/libnbx_$v.so; fi
  O=$GRAFT_REPO_ROOT/gpurun_out/m8pab_$v; rm -rf $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -o m8 -- python3 $GRAFT_REPO_ROOT/tools/dbg/m8_time.py 148 40 > $O.log 2>&1
  echo "== variant '$v'"
  python3 - $O <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'm8_' in r['Name'] or 'jk_m8' in r['Name']:
        if 'pack' in r['Name']: continue
        print('  ', r['Name'].split('::')[-1][:28], r['Calls'], round(float(r['AverageNs'])/1000,1), 'min', round(float(r['MinNs'])/1000,1))
PY
done
