cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/lf -o r -- $GRAFT_REPO_ROOT/tools/probes/launch_floor > /tmp/lf.log 2>&1
python3 - <<'PY'
import csv,glob,statistics as st
f=glob.glob('/tmp/lf/*kernel_trace.csv')[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
cfgs=[l.strip() for l in open('/tmp/lf.log') if l.startswith('cfg')]
for i,c in enumerate(cfgs):
    d=[int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in rows[i*20+5:(i+1)*20]]
    print(c, 'median us', st.median(d)/1e3)
PY
