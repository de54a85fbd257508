cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_c
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d /tmp/prof_c -o t --output-format csv -- python3 tools/exp_vocoder_only.py 6 > /tmp/c.log 2>&1 || { tail /tmp/c.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/prof_c/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'reschain' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows]
print(len(d), 'reschain launches (order k = 3, 7, 11 per pass):')
for i in range(0, len(d), 3): print('  ', ' '.join(f'{x:8.1f}' for x in d[i:i+3]))
PY
