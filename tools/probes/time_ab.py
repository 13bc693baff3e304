"""Interleaved timing A/B of two builds of libcmcd_hip.so on the GPU box (forward of the named batch + the saturated
batch), each measurement in its own process:  python tools/probes/time_ab.py libA.so libB.so [rounds]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
libs = [os.path.abspath(p) for p in sys.argv[1:3]]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        env = dict(os.environ, CMCD_LIB_PATH=l)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--forward-only", "--no-legs",
                              ], env=env, capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
            res[l].append((d["roofline"]["kernel_ms"], d.get("saturated", {}).get("kernel_ms")))
        except Exception as e:
            print("failed", l, out.stderr[-400:])
for l in libs:
    print(os.path.basename(l), " ".join("%.4f/%.3f" % (a, b or 0) for a, b in res[l]))
