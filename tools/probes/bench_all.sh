# one bench line per BASELINE configuration (forward throughput; the north star also carries the training objects)
cd $GRAFT_REPO_ROOT
for c in gmm_n300_k8 funnel_n300_k64 many_gmm_n2000_k256_dds many_gmm_var_n16000_k256 lgcp_n20_k128; do
  python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline --saturated 0 2>/dev/null | tail -1
done > gpurun_out/r01_k_all_configs.jsonl
python3 - <<'PY'
import json
for l in open('gpurun_out/r01_k_all_configs.jsonl'):
    r=json.loads(l); ro=r['roofline']
    print("%-28s value %.3e  ms/step %.4f  %s %.2f %s (frac %.3f)  kernel %s" % (r['config']['workload'], r['value'], r['ms_per_step'], ro['bound'], ro['achieved'], ro['unit'], ro['frac'], ro['kernel'][:28]))
PY
