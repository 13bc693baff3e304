# engine clock while the wide-batch lgcp path runs (fp32 matrix instructions on every CU): bash tools/probes/sclk_under_load.sh [n]
# rocm-smi is sampled every ~0.1 s beside the run; prints the distribution of the sclk readings taken while the GPU was busy
N=${1:-15000}
cd $GRAFT_REPO_ROOT
O=gpurun_out/sclk
mkdir -p $O
rm -f $O/samples.txt
( for k in $(seq 1 400); do rocm-smi --showclocks --showuse --showpower 2>/dev/null | grep -E "sclk|GPU use|Power" | tr '\n' ' ' >> $O/samples.txt; echo >> $O/samples.txt; sleep 0.05; done ) &
SAMP=$!
python3 tools/probes/lgcp_time.py $N > $O/run.txt 2>&1
kill $SAMP 2>/dev/null
wait $SAMP 2>/dev/null
tail -2 $O/run.txt
python3 - <<'PY'
import re, collections
rows = [l for l in open('gpurun_out/sclk/samples.txt') if 'sclk' in l]
busy = collections.Counter(); idle = collections.Counter()
for l in rows:
    m = re.search(r'sclk clock level: \S+ \((\d+)Mhz\)', l)
    u = re.search(r'GPU use \(%\): (\d+)', l)
    pw = re.search(r'Power \(W\): ([\d.]+)', l)
    if not m: continue
    (busy if (u and int(u.group(1)) > 50) else idle)[int(m.group(1))] += 1
print('samples', len(rows))
print('sclk MHz while GPU use > 50 %:', sorted(busy.items()))
print('sclk MHz otherwise:', sorted(idle.items()))
print('example lines:'); print(''.join(rows[len(rows)//2:len(rows)//2+3]))
PY
