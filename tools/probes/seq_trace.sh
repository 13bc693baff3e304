# per-kernel durations and inter-kernel gaps of one launch sequence: bash tools/probes/seq_trace.sh <tag> <config> <what> [particles]
T=$1; shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/$T
mkdir -p $O
rm -rf $O/prof
rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python3 tools/probes/seq_trace.py "$@" > $O/run.txt 2>&1
python3 - $O "$@" <<'PY'
import csv, glob, sys, collections, re
O = sys.argv[1]
reps = int(open(O + '/run.txt').read().split('calls')[-1].split()[0])
for f in glob.glob(O + '/prof/**/*kernel_trace.csv', recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
    per = len(rows) // (reps + reps // 3) if reps else 1
    rows = rows[-per * reps:]
    def short(n):
        n = re.sub(r'^void ', '', n); n = re.sub(r'cmcd::', '', n); n = re.sub(r'\(.*$', '', n)
        return n[:58]
    dur = collections.OrderedDict(); gaps = []
    for i, r in enumerate(rows):
        dur.setdefault(short(r['Kernel_Name']), []).append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
        if i: gaps.append(int(r['Start_Timestamp']) - int(rows[i - 1]['End_Timestamp']))
    med = lambda v: sorted(v)[len(v) // 2] / 1e3
    print(' '.join(sys.argv[2:]), ': %d kernels per call' % per)
    tot = 0.0
    for k, v in dur.items():
        c = len(v) / reps
        tot += med(v) * c
        print('  %-58s x %5.1f per call  median %8.2f us' % (k, c, med(v)))
    gaps.sort()
    print('  gaps: median %.2f us, p90 %.2f us, sum per call %.2f us' % (gaps[len(gaps) // 2] / 1e3, gaps[9 * len(gaps) // 10] / 1e3, sum(gaps) / 1e3 / reps))
    print('  sum of median durations per call %.2f us; span per call %.2f us' % (tot, (int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])) / 1e3 / reps))
PY
rm -rf $O/prof
